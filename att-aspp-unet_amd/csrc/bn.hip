// BatchNorm2d / ReLU / MaxPool2d(2) / Dropout forward and backward around the MFMA
// convolutions (pipeline:64, the BNs of :71-90, MaxPool2d :115-118, Dropout :79).
// All HBM-bound: every access is a 16-byte vector of 8 bf16 channels, a thread keeps a
// fixed channel group so per-channel reductions live in registers, are combined through
// LDS per block and leave the block as one fp32 atomic per channel into one of
// AAU_STAT_REPLICAS accumulator copies (spreads same-address contention).
#include <stdlib.h>
#include "common.h"

namespace aau {

// ---- channel-group thread map: tid -> (pixel lane, channel group) ----
struct CGMap {
    int CG, PL, T;
    __device__ __host__ explicit CGMap(int C) {
        CG = C >> 3;
        PL = 256 / CG;
        if (PL < 1) PL = 1;
        T = CG * PL;
    }
};

// Sum `acc[8]` over the PL pixel lanes of the block; result valid on threads with pl == 0.
__device__ __forceinline__ void block_sum8(float acc[8], float* red, const CGMap& mp, int tid) {
    __syncthreads();
    if (tid < mp.T) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    }
    __syncthreads();
    if (tid < mp.CG) {
        for (int pl = 1; pl < mp.PL; ++pl) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += red[(pl * mp.CG + tid) * 8 + j];
        }
    }
}

__device__ __forceinline__ u32x4 ld16(const unsigned short* p) { return *(const u32x4*)p; }
__device__ __forceinline__ void st16(unsigned short* p, const u32x4& v) { *(u32x4*)p = v; }
__device__ __forceinline__ void ldf8(const float* p, float f[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}

// ------------------------------------------------------------------------------------
// 256 threads = 8 channels x 32 replica lanes: a lane loads its replica's two (hi, lo) pairs with 16-B loads, the 32
// lanes are added with shuffles (integers: exact, any order), lane 0 forms mean / variance in fp64 and rounds once.
// (One thread per channel walking the 32 replicas took 10-15 us per launch, 31 launches per step.)
__device__ __forceinline__ void bn_finalize_body(const long long* stats, const float* gamma, const float* beta,
                                                  float* rmean, float* rvar, int64_t* nbt, float* scale,
                                                  float* shift, float* smean, float* sinvstd, int C, float count,
                                                  float eps, float mom) {
    static_assert(AAU_STAT_REPLICAS == 32, "one replica per lane of a 32-lane group");
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    const int r = threadIdx.x & 31;
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    const bool ok = c < C;
    i64x2 a = i64x2{0, 0}, b = i64x2{0, 0};
    if (ok) {
        a = *(const i64x2*)(stats + (((size_t)r * 2 + 0) * C + c) * 2);
        b = *(const i64x2*)(stats + (((size_t)r * 2 + 1) * C + c) * 2);
    }
    long long h1 = a[0], l1 = a[1], h2 = b[0], l2 = b[1];
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        h1 += __shfl_xor(h1, o, 32);
        l1 += __shfl_xor(l1, o, 32);
        h2 += __shfl_xor(h2, o, 32);
        l2 += __shfl_xor(l2, o, 32);
    }
    if (!ok || r != 0) return;
    const bool poisoned = stats[(size_t)AAU_STAT_REPLICAS * 2 * C * 2] != 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double s1 = poisoned ? nan : (double)h1 * (1.0 / 256.0) + (double)l1 * (1.0 / 4503599627370496.0);
    const double s2 = poisoned ? nan : (double)h2 * (1.0 / 256.0) + (double)l2 * (1.0 / 4503599627370496.0);
    const double mean_d = s1 / (double)count;
    const float mean = (float)mean_d;
    const float var = fmaxf((float)(s2 / (double)count - mean_d * mean_d), 0.f);
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    smean[c] = mean;
    sinvstd[c] = invstd;
    if (rmean) {
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        rmean[c] = (1.f - mom) * rmean[c] + mom * mean;
        rvar[c] = (1.f - mom) * rvar[c] + mom * unb;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const long long* stats, const float* gamma, const float* beta,
                                                          float* rmean, float* rvar, int64_t* nbt, float* scale,
                                                          float* shift, float* smean, float* sinvstd, int C, float count,
                                                          float eps, float mom) {
    bn_finalize_body(stats, gamma, beta, rmean, rvar, nbt, scale, shift, smean, sinvstd, C, count, eps, mom);
}
// Several BatchNorm layers of the same width in ONE launch (blockIdx.y = layer): the four spatial branches of the ASPP
// (pipeline:80-83), the two 1x1 convs of an attention gate.  Each layer's arithmetic is exactly the single launch's.
constexpr int BN_MULTI_MAX = 8;
struct BnMulti { const void* p[BN_MULTI_MAX][12]; };
__global__ __launch_bounds__(256) void bn_finalize_multi_kernel(const BnMulti a, int C, float count, float eps, float mom) {
    const void* const* q = a.p[blockIdx.y];
    bn_finalize_body((const long long*)q[0], (const float*)q[1], (const float*)q[2], (float*)q[3], (float*)q[4], (int64_t*)q[5],
                     (float*)q[6], (float*)q[7], (float*)q[8], (float*)q[9], C, count, eps, mom);
}

__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rmean,
                                    const float* rvar, float* scale, float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// y = relu?(z*scale+shift) (* dropout).  A thread owns one 8-channel group (scale / shift live in 16
// registers) and walks over pixels; consecutive threads cover consecutive 16-B vectors of a pixel row.
__device__ __forceinline__ void bn_act_body(const unsigned short* z, int zp, unsigned short* y, int yp,
                                            const float* scale, const float* shift, int64_t M, int C,
                                            int relu, int64_t bhw, float drop_p, const uint64_t* seedp,
                                            int64_t ppb) {
    const CGMap mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float sc[8], sh[8];
    ldf8(scale + c, sc);
    ldf8(shift + c, sh);
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint64_t seed = drop_p > 0.f ? *seedp : 0;   // device-resident: a captured hipGraph sees a fresh value per replay
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
        const int64_t ms = bhw > 0 ? m / bhw : m;
        float f[8];
        unpack8(ld16(z + ms * zp + c), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = f[j] * sc[j] + sh[j];
            if (relu) t = fmaxf(t, 0.f);
            if (drop_p > 0.f) t = hash_uniform(seed, (uint64_t)(m * C + c + j)) >= drop_p ? t * keep_scale : 0.f;
            f[j] = t;
        }
        st16(y + m * yp + c, pack8(f));
    }
}

__global__ __launch_bounds__(256) void bn_act_kernel(const unsigned short* z, int zp, unsigned short* y, int yp,
                                                     const float* scale, const float* shift, int64_t M, int C,
                                                     int relu, int64_t bhw, float drop_p, const uint64_t* seedp,
                                                     int64_t ppb) {
    bn_act_body(z, zp, y, yp, scale, shift, M, C, relu, bhw, drop_p, seedp, ppb);
}
__global__ __launch_bounds__(256) void bn_act_multi_kernel(const BnMulti a, int zp, int yp, int64_t M, int C, int relu, int64_t ppb) {
    const void* const* q = a.p[blockIdx.y];
    bn_act_body((const unsigned short*)q[0], zp, (unsigned short*)q[1], yp, (const float*)q[2], (const float*)q[3], M, C, relu, 0,
                0.f, nullptr, ppb);
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const unsigned short* y, int yp, unsigned short* p, int pp,
                                                       int N, int H, int W, int C) {
    const int CG = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * CG;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t mo = v / CG;
        const int c = (int)(v - mo * CG) * 8;
        int xo, yo, n;
        decode3(mo, Wo, Ho, xo, yo, n);
        const unsigned short* b = y + (((int64_t)n * H + 2 * yo) * W + 2 * xo) * yp + c;
        float a0[8], a1[8], a2[8], a3[8];
        unpack8(ld16(b), a0);
        unpack8(ld16(b + yp), a1);
        unpack8(ld16(b + (int64_t)W * yp), a2);
        unpack8(ld16(b + (int64_t)W * yp + yp), a3);
#pragma unroll
        for (int j = 0; j < 8; ++j) a0[j] = fmaxf(fmaxf(a0[j], a1[j]), fmaxf(a2[j], a3[j]));
        st16(p + mo * pp + c, pack8(a0));
    }
}

// y = relu(z*scale+shift) and p = maxpool2x2(y) in one pass: a thread owns a channel group (constants in
// registers) and walks over 2x2 windows
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const unsigned short* z, int zp, unsigned short* y, int yp,
                                                          unsigned short* p, int pp, const float* scale,
                                                          const float* shift, int N, int H, int W, int C, int64_t ipb) {
    const CGMap mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo;
    float sc[8], sh[8];
    ldf8(scale + c, sc);
    ldf8(shift + c, sh);
    const int64_t i0 = slice_begin(ipb), i1 = min(total, i0 + ipb);
    for (int64_t mo = i0 + pl; mo < i1; mo += mp.PL) {
        int xo, yo, n;
        decode3(mo, Wo, Ho, xo, yo, n);
        const int64_t p00 = ((int64_t)n * H + 2 * yo) * W + 2 * xo;
        const int64_t pix[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float best[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[8];
            unpack8(ld16(z + pix[k] * zp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
            const u32x4 pk = pack8(f);
            st16(y + pix[k] * yp + c, pk);
            unpack8(pk, f);  // pool the bf16 values that were stored, as the separate kernel does
#pragma unroll
            for (int j = 0; j < 8; ++j) best[j] = k == 0 ? f[j] : fmaxf(best[j], f[j]);
        }
        st16(p + mo * pp + c, pack8(best));
    }
}

// ---- backward pass 1: masked gradient + per-channel sums ----
// POOL = true: one thread per 2x2 window (H, W even) so the max-pool routing needs no re-reads.
template <bool POOL>
__device__ __forceinline__ void bn_bwd_reduce_body(
    const unsigned short* z, int zp, const unsigned short* dy, int dyp, const unsigned short* dpool, int dpp,
    unsigned short* dz, int dzp, const float* scale, const float* shift, const float* mean, const float* invstd,
    float* red, int N, int H, int W, int C, int relu, float drop_p, const uint64_t* seedp, int64_t items_per_block,
    float* ws) {
    const uint64_t seed = drop_p > 0.f ? *seedp : 0;   // device-resident: a captured hipGraph sees a fresh value per replay
    __shared__ float sred[256 * 8];
    const CGMap mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG;
    const int c = cg * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if (tid < mp.T) {
        float sc[8], sh[8], mu[8], is[8];
        ldf8(scale + c, sc); ldf8(shift + c, sh); ldf8(mean + c, mu); ldf8(invstd + c, is);
        const int Ho = H >> 1, Wo = W >> 1;
        const int64_t nitems = POOL ? (int64_t)N * Ho * Wo : (int64_t)N * H * W;
        const int64_t i0 = slice_begin(items_per_block);
        const int64_t i1 = min(nitems, i0 + items_per_block);
        for (int64_t it = i0 + pl; it < i1; it += mp.PL) {
            if constexpr (POOL) {
                int xo, yo, n;
                decode3(it, Wo, Ho, xo, yo, n);
                const int64_t p00 = ((int64_t)n * H + 2 * yo) * W + 2 * xo;
                const int64_t pix[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
                float zz[4][8], yy[4][8], g[4][8], dp[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unpack8(ld16(z + pix[k] * zp + c), zz[k]);
                    if (dy) unpack8(ld16(dy + pix[k] * dyp + c), g[k]);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) g[k][j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t2 = zz[k][j] * sc[j] + sh[j];
                        // the pooled tensor holds bf16(relu(bn(z))): compare what the forward compared
                        yy[k][j] = bf2f(f2bf(relu ? fmaxf(t2, 0.f) : t2));
                    }
                }
                unpack8(ld16(dpool + it * dpp + c), dp);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int best = 0;
                    float bv = yy[0][j];
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (yy[k][j] > bv) { bv = yy[k][j]; best = k; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float gv = g[k][j] + (k == best ? dp[j] : 0.f);
                        if (relu && !(yy[k][j] > 0.f)) gv = 0.f;
                        const float zh = (zz[k][j] - mu[j]) * is[j];
                        s1[j] += gv;
                        s2[j] += gv * zh;
                        g[k][j] = gv;
                    }
                }
                if (dz)   // dz == null: aau_bn_bwd_apply_pool redoes the routing from dy, z and dpool
#pragma unroll
                    for (int k = 0; k < 4; ++k) st16(dz + pix[k] * dzp + c, pack8(g[k]));
            } else {
                float zz[8], g[8];
                unpack8(ld16(z + it * zp + c), zz);
                unpack8(ld16(dy + it * dyp + c), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t2 = zz[j] * sc[j] + sh[j];
                    float gv = g[j];
                    if (drop_p > 0.f)
                        gv = hash_uniform(seed, (uint64_t)(it * C + c + j)) >= drop_p ? gv * keep_scale : 0.f;
                    if (relu && !(t2 > 0.f)) gv = 0.f;
                    const float zh = (zz[j] - mu[j]) * is[j];
                    s1[j] += gv;
                    s2[j] += gv * zh;
                    g[j] = gv;
                }
                if (dz) st16(dz + it * dzp + c, pack8(g));   // dz == null: the apply pass recomputes the mask from dy
            }
        }
    }
    // this workgroup's row [2][C] of partial sums; red_fold_launch adds the rows in a fixed order (common.h)
    float* r = red_row(ws, 2 * C, blockIdx.x);
    block_sum8(s1, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + c) = f32x4{s1[0], s1[1], s1[2], s1[3]};
        *(f32x4*)(r + c + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
    }
    block_sum8(s2, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + C + c) = f32x4{s2[0], s2[1], s2[2], s2[3]};
        *(f32x4*)(r + C + c + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
    }
}

template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const unsigned short* z, int zp, const unsigned short* dy, int dyp, const unsigned short* dpool, int dpp,
    unsigned short* dz, int dzp, const float* scale, const float* shift, const float* mean, const float* invstd,
    float* red, int N, int H, int W, int C, int relu, float drop_p, const uint64_t* seedp, int64_t items_per_block,
    float* ws) {
    bn_bwd_reduce_body<POOL>(z, zp, dy, dyp, dpool, dpp, dz, dzp, scale, shift, mean, invstd, red, N, H, W, C, relu, drop_p, seedp,
                             items_per_block, ws);
}
// [z, dy, scale, shift, mean, invstd, red, ws] per layer
__global__ __launch_bounds__(256) void bn_bwd_reduce_multi_kernel(const BnMulti a, int zp, int dyp, int N, int H, int W, int C,
                                                                  int relu, int64_t items_per_block) {
    const void* const* q = a.p[blockIdx.y];
    bn_bwd_reduce_body<false>((const unsigned short*)q[0], zp, (const unsigned short*)q[1], dyp, nullptr, 0, nullptr, 0,
                              (const float*)q[2], (const float*)q[3], (const float*)q[4], (const float*)q[5], (float*)q[6], N, H, W,
                              C, relu, 0.f, nullptr, items_per_block, (float*)q[7]);
}

// ---- backward pass 2: dz = gamma*invstd*(g - mean(g) - zhat*mean(g*zhat)) in place ----
// dy == null: the masked gradient g is read from dz (written by the reduce pass, pooled layers);
// dy != null: g = relu'(z*scale+shift) * dropout(dy) is recomputed here, saving one tensor write + read.
// A thread owns one channel group: its seven per-channel constants live in registers.
__device__ __forceinline__ void bn_bwd_apply_body(const unsigned short* z, int zp, unsigned short* dz, int dzp,
                                                  const float* gamma, const float* mean, const float* invstd,
                                                  const float* red, float* dgamma, float* dbeta, int64_t M,
                                                  int C, const unsigned short* dy, int dyp, const float* scale,
                                                  const float* shift, int relu, float drop_p, const uint64_t* seedp,
                                                  const float* dl, const float* wout, int64_t ppb) {
    const uint64_t seed = drop_p > 0.f ? *seedp : 0;
    extern __shared__ float sm[];   // [2][C] replica sums, computed once per workgroup
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        const float a = red[cc], b = red[C + cc];       // totals of the reduce pass
        sm[cc] = a;
        sm[C + cc] = b;
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[cc] += a;
            if (dgamma) dgamma[cc] += b;
        }
    }
    __syncthreads();
    const CGMap mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float k0[8], k1[8], k2[8], mu[8], is[8], sc[8], sh[8];
    ldf8(mean + c, mu);
    ldf8(invstd + c, is);
    // dl != null: the incoming gradient is rank one, dy[m][c] = bf16(dl[m] * wout[c]) (network head, never stored)
    float wv[8];
    if (dy || dl) { ldf8(scale + c, sc); ldf8(shift + c, sh); }
    if (dl) ldf8(wout + c, wv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k0[j] = gamma[c + j] * is[j];
        k1[j] = sm[c + j] / (float)M;
        k2[j] = sm[C + c + j] / (float)M;
    }
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // rev: workgroups walk the tensor from its END (the part the preceding reduce pass touched last, i.e. what the
    // 256-MiB Infinity Cache still holds when the tensor is larger than the cache)
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
        float zz[8], g[8];
        unpack8(ld16(z + m * zp + c), zz);
        if (dy || dl) {
            if (dy) {
                unpack8(ld16(dy + m * dyp + c), g);
            } else {
                const float gl = dl[m];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = bf2f(f2bf(gl * wv[j]));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gv = g[j];
                if (drop_p > 0.f) gv = hash_uniform(seed, (uint64_t)(m * C + c + j)) >= drop_p ? gv * keep_scale : 0.f;
                if (relu && !(zz[j] * sc[j] + sh[j] > 0.f)) gv = 0.f;
                g[j] = bf2f(f2bf(gv));   // same rounding as the stored form
            }
        } else {
            unpack8(ld16(dz + m * dzp + c), g);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zh = (zz[j] - mu[j]) * is[j];
            g[j] = k0[j] * (g[j] - k1[j] - zh * k2[j]);
        }
        st16(dz + m * dzp + c, pack8(g));
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* z, int zp, unsigned short* dz, int dzp,
                                                           const float* gamma, const float* mean, const float* invstd,
                                                           const float* red, float* dgamma, float* dbeta, int64_t M,
                                                           int C, const unsigned short* dy, int dyp, const float* scale,
                                                           const float* shift, int relu, float drop_p, const uint64_t* seedp,
                                                           const float* dl, const float* wout, int64_t ppb) {
    bn_bwd_apply_body(z, zp, dz, dzp, gamma, mean, invstd, red, dgamma, dbeta, M, C, dy, dyp, scale, shift, relu, drop_p, seedp, dl,
                      wout, ppb);
}
// [z, dz, gamma, mean, invstd, red, dgamma, dbeta, dy, scale, shift] per layer
__global__ __launch_bounds__(256) void bn_bwd_apply_multi_kernel(const BnMulti a, int zp, int dzp, int dyp, int64_t M, int C, int relu,
                                                                 int64_t ppb) {
    const void* const* q = a.p[blockIdx.y];
    bn_bwd_apply_body((const unsigned short*)q[0], zp, (unsigned short*)q[1], dzp, (const float*)q[2], (const float*)q[3],
                      (const float*)q[4], (const float*)q[5], (float*)q[6], (float*)q[7], M, C, (const unsigned short*)q[8], dyp,
                      (const float*)q[9], (const float*)q[10], relu, 0.f, nullptr, nullptr, nullptr, ppb);
}

// The apply pass of a POOLED layer (the encoder's second ConvBNReLU: its output feeds the skip connection and, through
// MaxPool2d, the next level): one thread per 2x2 window and channel group redoes what the reduce pass did -- the gradient
// of a window is dy (skip path) plus dpool routed to the FIRST maximum of bf16(relu(bn(z))) in window order, masked by the
// ReLU -- and applies the BatchNorm backward formula.  The reduce pass then need not store the routed gradient: one write
// and one read of the layer's largest tensor against a re-read of dpool (a quarter of its size).
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(
    const unsigned short* z, int zp, unsigned short* dz, int dzp, const float* gamma, const float* mean, const float* invstd,
    const float* red, float* dgamma, float* dbeta, int N, int H, int W, int C, const unsigned short* dy, int dyp,
    const unsigned short* dpool, int dpp, const float* scale, const float* shift, int relu, int64_t items_per_block) {
    extern __shared__ float sm[];   // [2][C] totals of the reduce pass
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        const float a = red[cc], b = red[C + cc];
        sm[cc] = a;
        sm[C + cc] = b;
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[cc] += a;
            if (dgamma) dgamma[cc] += b;
        }
    }
    __syncthreads();
    const CGMap mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float k0[8], k1[8], k2[8], mu[8], is[8], sc[8], sh[8];
    ldf8(mean + c, mu);
    ldf8(invstd + c, is);
    ldf8(scale + c, sc);
    ldf8(shift + c, sh);
    const float Mf = (float)((int64_t)N * H * W);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k0[j] = gamma[c + j] * is[j];
        k1[j] = sm[c + j] / Mf;
        k2[j] = sm[C + c + j] / Mf;
    }
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t nitems = (int64_t)N * Ho * Wo;
    const int64_t i0 = slice_begin(items_per_block);
    const int64_t i1 = min(nitems, i0 + items_per_block);
    for (int64_t it = i0 + pl; it < i1; it += mp.PL) {
        int xo, yo, n;
        decode3(it, Wo, Ho, xo, yo, n);
        const int64_t p00 = ((int64_t)n * H + 2 * yo) * W + 2 * xo;
        const int64_t pix[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float zz[4][8], yy[4][8], g[4][8], dp[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unpack8(ld16(z + pix[k] * zp + c), zz[k]);
            if (dy) unpack8(ld16(dy + pix[k] * dyp + c), g[k]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[k][j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t2 = zz[k][j] * sc[j] + sh[j];
                yy[k][j] = bf2f(f2bf(relu ? fmaxf(t2, 0.f) : t2));   // what the forward compared
            }
        }
        unpack8(ld16(dpool + it * dpp + c), dp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int best = 0;
            float bv = yy[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (yy[k][j] > bv) { bv = yy[k][j]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gv = g[k][j] + (k == best ? dp[j] : 0.f);
                if (relu && !(yy[k][j] > 0.f)) gv = 0.f;
                gv = bf2f(f2bf(gv));                                 // the rounding of the stored intermediate
                const float zh = (zz[k][j] - mu[j]) * is[j];
                g[k][j] = k0[j] * (gv - k1[j] - zh * k2[j]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st16(dz + pix[k] * dzp + c, pack8(g[k]));
    }
}

// Last step of the backward pass: BatchNorm+ReLU backward of d1[0] fused with the weight gradient of its
// Conv2d(1, C, 3, pad 1) (pipeline:113).  The first layer has no input gradient, so dz is consumed here and never
// written (-2 tensor passes of the largest activation).  dz is rounded to bf16 exactly as the stored form would be.
// ws: one row [C*9] of weight-gradient partial sums per workgroup (threads combined in a fixed order through LDS);
// red_fold_launch adds the rows in row order into dw: no float atomics.
// STAGE: the flat pixel window of the workgroup's slice (+ one image row and one pixel either side) is copied into LDS
// first and the 9 taps of a pixel are LDS reads masked at the image borders (as conv1_fwd_kernel).
template <bool STAGE>
__global__ __launch_bounds__(256) void bn_bwd_apply_conv1_kernel(const unsigned short* z, int zp, const float* gamma,
                                                                 const float* mean, const float* invstd,
                                                                 const float* red, float* dgamma, float* dbeta, int M,
                                                                 int C, const unsigned short* dy, int dyp,
                                                                 const float* scale, const float* shift,
                                                                 const float* x, int H, int W, float* ws,
                                                                 const float* wconv, int ppb) {
    extern __shared__ float sm[];   // [2][C] BN sums, [C*9] conv weights (z == null), [pixel lanes][C*9] thread partials
    float* swc = sm + 2 * C;
    float* spart = sm + 11 * C;
    float* sx = spart + (size_t)(256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1) * 9 * C;     // (STAGE) the pixel window
    if constexpr (STAGE) {
        int pp = ppb;
        const int w0 = (int)slice_begin(pp);
        const int w1 = min(M, w0 + pp);
        const int n = (w1 - w0) + 2 * W + 2;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int g = w0 - W - 1 + i;
            sx[i] = (g >= 0 && g < M) ? x[g] : 0.f;
        }
    }
    if (!z)
        for (int i = threadIdx.x; i < C * 9; i += 256) swc[i] = wconv[i];
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        const float a = red[cc], b = red[C + cc];       // totals of the reduce pass
        sm[cc] = a;
        sm[C + cc] = b;
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[cc] += a;
            if (dgamma) dgamma[cc] += b;
        }
    }
    __syncthreads();
    // A thread owns FOUR channels (not the eight of the other BN kernels): 36 accumulators + 20 constants keep
    // it under 96 VGPRs, i.e. 5 waves per SIMD -- at 8 channels (184 VGPRs, 2 waves) the loop was latency bound
    // and slower than the two unfused passes.
    const int CG4 = C >> 2, PL4 = 256 / CG4, T4 = CG4 * PL4;
    const int tid = threadIdx.x;
    const int cg = tid % CG4, pl = tid / CG4, c = cg * 4;
    if (tid < T4) {
        // dz = k0*(g - k1 - zhat*k2) = ka*g + kb*z + kc  (three folded constants per channel)
        float ka[4], kb[4], kc[4], sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float is = invstd[c + j], mu = mean[c + j];
            const float k0 = gamma[c + j] * is;
            const float k1 = sm[c + j] / (float)M, k2 = sm[C + c + j] / (float)M;
            ka[j] = k0;
            kb[j] = -k0 * k2 * is;
            kc[j] = -k0 * k1 + k0 * k2 * is * mu;
            sc[j] = scale[c + j];
            sh[j] = shift[c + j];
        }
        // accumulators as float pairs: the compiler emits v_pk_fma_f32 (two taps per instruction)
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 acc[4][5];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[j][k] = f32x2{0.f, 0.f};
        const int m0 = (int)slice_begin(ppb), m1 = min(M, m0 + ppb);
        // (row, column) of the thread's pixel, advanced incrementally (no division in the loop)
        int m = m0 + pl;
        int xx = m % W, row = m / W;            // row = n*H + y
        int yy = row % H;
        const int stepx = PL4 % W, stepr = PL4 / W;
        auto load_px = [&](u32x2& zq, u32x2& gq, f32x2 v[5]) {
            gq = *(const u32x2*)(dy + (int64_t)m * dyp + c);
            float t[10];
            if constexpr (STAGE) {
                const float* p = sx + (m - m0);               // tap (ky, kx) = p[ky * W + kx]
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = (ky != 0 || yy > 0) && (ky != 2 || yy < H - 1) && (kx != 0 || xx > 0) && (kx != 2 || xx < W - 1);
                        const float tv = p[ky * W + kx];
                        t[ky * 3 + kx] = ok ? tv : 0.f;
                    }
            } else {
                conv1_taps(x + (int64_t)(row - yy) * W, yy, xx, H, W, t);
            }
            t[9] = 0.f;
            if (z) {
                zq = *(const u32x2*)(z + (int64_t)m * zp + c);
            } else {   // z of the first layer is not stored: the same fma chain as conv1_fwd gives the same bf16 bits
                zq[0] = pack2(conv1_dot(t, swc + (c + 0) * 9), conv1_dot(t, swc + (c + 1) * 9));
                zq[1] = pack2(conv1_dot(t, swc + (c + 2) * 9), conv1_dot(t, swc + (c + 3) * 9));
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = f32x2{t[2 * k], t[2 * k + 1]};
            // advance to this thread's next pixel
            m += PL4;
            xx += stepx;
            row += stepr;
            if (xx >= W) { xx -= W; ++row; }
            yy = row % H;
        };
        auto fma_px = [&](const u32x2& zq, const u32x2& gq, const f32x2 v[5]) {
            const float zz[4] = {pair_lo(zq[0]), pair_hi(zq[0]),
                                 pair_lo(zq[1]), pair_hi(zq[1])};
            const float g[4] = {pair_lo(gq[0]), pair_hi(gq[0]),
                                pair_lo(gq[1]), pair_hi(gq[1])};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gm = (zz[j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
                const float gv = bf2f(f2bf(ka[j] * gm + kb[j] * zz[j] + kc[j]));
                const f32x2 g2 = f32x2{gv, gv};
#pragma unroll
                for (int k = 0; k < 5; ++k) acc[j][k] = __builtin_elementwise_fma(g2, v[k], acc[j][k]);
            }
        };
        // two pixels per trip: both sets of loads are in flight before the first FMA block
        while (m + PL4 < m1) {
            u32x2 za, ga, zb, gb;
            f32x2 va[5], vb[5];
            load_px(za, ga, va);
            load_px(zb, gb, vb);
            fma_px(za, ga, va);
            fma_px(zb, gb, vb);
        }
        if (m < m1) {
            u32x2 za, ga;
            f32x2 va[5];
            load_px(za, ga, va);
            fma_px(za, ga, va);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) spart[(size_t)pl * C * 9 + (c + j) * 9 + k] = acc[j][k >> 1][k & 1];
    }
    __syncthreads();
    float* row = red_row(ws, C * 9, blockIdx.x);
    for (int i = threadIdx.x; i < C * 9; i += 256) {
        float a = 0.f;
        for (int q = 0; q < PL4; ++q) a += spart[(size_t)q * C * 9 + i];     // pixel lanes in lane order
        row[i] = a;
    }
}

// rows per block so that the grid is ~8 workgroups per CU and every thread gets a few iterations
// ---- fixed-order sum of the workgroup rows (common.h: red_fold_launch) ----
// A workgroup owns 4 consecutive 16-B vectors of the row (64 B) and 64 row lanes: thread (rl, v) adds rows rl, rl+64, ..
// in that order, the 64 partial sums are then added in lane order.
__device__ __forceinline__ void red_fold_body(const float* ws, int n, int nblk, float* out, int n_out, float* acc,
                                              int n_acc, float* acc2) {
    __shared__ f32x4 sm[256];
    const int v = (int)blockIdx.x * 4 + (threadIdx.x & 3), rl = threadIdx.x >> 2;
    const int nv = n >> 2;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    if (v < nv) {
        const float* p = ws + (size_t)v * 4;
        int r = rl;
        for (; r + 192 < nblk; r += 256) {      // 4 independent loads in flight, fixed order of the additions
            const f32x4 t0 = *(const f32x4*)(p + (size_t)r * n), t1 = *(const f32x4*)(p + (size_t)(r + 64) * n);
            const f32x4 t2 = *(const f32x4*)(p + (size_t)(r + 128) * n), t3 = *(const f32x4*)(p + (size_t)(r + 192) * n);
            a += t0; a += t1; a += t2; a += t3;
        }
        for (; r < nblk; r += 64) a += *(const f32x4*)(p + (size_t)r * n);
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    if (rl != 0 || v >= nv) return;
    for (int k = 1; k < 64; ++k) a += sm[k * 4 + (threadIdx.x & 3)];
    const int i = v * 4;
    if (i < n_out) {
        *(f32x4*)(out + i) = a;
    } else if (i < n_out + n_acc) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i - n_out + k] += a[k];
    } else if (i == n_out + n_acc && acc2) {
        acc2[0] += a[0];
    }
}

__global__ __launch_bounds__(256) void red_fold_kernel(const float* ws, int n, int nblk, float* out, int n_out, float* acc,
                                                       int n_acc, float* acc2) {
    red_fold_body(ws, n, nblk, out, n_out, acc, n_acc, acc2);
}
// [ws, out] per layer (the reduce pass's entries 7 and 6)
__global__ __launch_bounds__(256) void red_fold_multi_kernel(const BnMulti a, int n, int nblk) {
    const void* const* q = a.p[blockIdx.y];
    red_fold_body((const float*)q[7], n, nblk, (float*)q[6], n, nullptr, 0, nullptr);
}

int red_fold_launch(const float* ws, int n, int nblk, float* out, int n_out, float* acc, int n_acc, float* acc2, hipStream_t s) {
    hipLaunchKernelGGL(red_fold_kernel, dim3((unsigned)((n / 4 + 3) / 4)), dim3(256), 0, s, ws, n, nblk, out, n_out, acc, n_acc, acc2);
    return check_launch("red_fold");
}

static inline void rows_split(int64_t M, int PL, int64_t* blocks, int64_t* ppb) {
    int64_t b = (M + (int64_t)PL * 4 - 1) / ((int64_t)PL * 4);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    *ppb = (M + b - 1) / b;
    *blocks = (M + *ppb - 1) / *ppb;
}

static inline int grid_for(int64_t total_threads, int cap = 256 * 8) {
    int64_t g = (total_threads + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace aau

using namespace aau;

#define CHK_C(fn, C) AAU_REQUIRE((C) > 0 && (C) % 8 == 0 && (C) <= 2048, fn ": C=%d must be a multiple of 8 in [8, 2048]", (int)(C))

extern "C" int aau_bn_finalize(const aau_stat* stats, int64_t stats_bytes, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                               float* save_mean, float* save_invstd, int C, int64_t count, float eps,
                               float momentum, void* stream) {
    AAU_REQUIRE(stats && gamma && beta && scale && shift && save_mean && save_invstd, "aau_bn_finalize: null pointer");
    AAU_REQUIRE(C > 0 && count > 0, "aau_bn_finalize: C=%d count=%lld", C, (long long)count);
    AAU_CHECK_STAT("aau_bn_finalize", stats, stats_bytes, C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const long long*)stats, gamma,
                       beta, running_mean, running_var, num_batches_tracked, scale, shift, save_mean, save_invstd, C,
                       (float)count, eps, momentum);
    return check_launch("aau_bn_finalize");
}

extern "C" int aau_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float* scale, float* shift, int C, float eps,
                                void* stream) {
    AAU_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "aau_bn_fold_eval: bad args");
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, scale, shift, C, eps);
    return check_launch("aau_bn_fold_eval");
}

extern "C" int aau_bn_act(const aau_bf16* z, int z_pitch, aau_bf16* y, int y_pitch, const float* scale,
                          const float* shift, int64_t M, int C, int relu, int64_t bcast_hw, float drop_p,
                          const uint64_t* drop_seed, void* stream) {
    AAU_REQUIRE(z && y && scale && shift && M > 0, "aau_bn_act: bad args");
    CHK_C("aau_bn_act", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && y_pitch % 8 == 0, "aau_bn_act: pitches must be multiples of 8");
    AAU_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "aau_bn_act: drop_p=%f", drop_p);
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, z_pitch, y,
                       y_pitch, scale, shift, M, C, relu, bcast_hw, drop_p, drop_seed, ppb);
    return check_launch("aau_bn_act");
}

extern "C" int aau_bn_act_pool(const aau_bf16* z, int z_pitch, aau_bf16* y, int y_pitch, aau_bf16* p, int p_pitch,
                               const float* scale, const float* shift, int N, int H, int W, int C, void* stream) {
    AAU_REQUIRE(z && y && p && scale && shift && N > 0, "aau_bn_act_pool: bad args");
    AAU_REQUIRE(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0, "aau_bn_act_pool: H=%d W=%d must be even", H, W);
    CHK_C("aau_bn_act_pool", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && y_pitch % 8 == 0 && p_pitch % 8 == 0, "aau_bn_act_pool: pitches must be multiples of 8");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_bn_act_pool: pixel count overflows int32");
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ipb;
    rows_split((int64_t)N * (H / 2) * (W / 2), CGMap(C).PL, &blocks, &ipb);
    if (next_traversal()) ipb = -ipb;
    hipLaunchKernelGGL(bn_act_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, z_pitch, y,
                       y_pitch, p, p_pitch, scale, shift, N, H, W, C, ipb);
    return check_launch("aau_bn_act_pool");
}

extern "C" int aau_maxpool2(const aau_bf16* y, int y_pitch, aau_bf16* p, int p_pitch, int N, int H, int W, int C,
                            void* stream) {
    AAU_REQUIRE(y && p && N > 0, "aau_maxpool2: bad args");
    AAU_REQUIRE(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0, "aau_maxpool2: H=%d W=%d must be even", H, W);
    CHK_C("aau_maxpool2", C);
    AAU_REQUIRE(y_pitch % 8 == 0 && p_pitch % 8 == 0, "aau_maxpool2: pitches must be multiples of 8");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_maxpool2: pixel count overflows int32");
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       (hipStream_t)stream, y, y_pitch, p, p_pitch, N, H, W, C);
    return check_launch("aau_maxpool2");
}

extern "C" int64_t aau_bn_red_ws_bytes(int C) {
    if (C <= 0) return 0;
    return (int64_t)red_ws_floats(9 * C + 8, AAU_BN_RED_MAX_BLOCKS) * (int64_t)sizeof(float);   // widest row: C*9 (first layer dw)
}

extern "C" int aau_bn_bwd_reduce(const aau_bf16* z, int z_pitch, const aau_bf16* dy, int dy_pitch,
                                 const aau_bf16* dpool, int dpool_pitch, aau_bf16* dz, int dz_pitch,
                                 const float* scale, const float* shift, const float* save_mean,
                                 const float* save_invstd, float* red, int N, int H, int W, int C, int relu,
                                 float drop_p, const uint64_t* drop_seed, float* ws, void* stream) {
    AAU_REQUIRE(z && scale && shift && save_mean && save_invstd && red && ws, "aau_bn_bwd_reduce: null pointer");
    AAU_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)red & 15) == 0, "aau_bn_bwd_reduce: red / ws must be 16-byte aligned");
    AAU_REQUIRE(dy || dpool, "aau_bn_bwd_reduce: needs at least one gradient source");
    CHK_C("aau_bn_bwd_reduce", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0 && dy_pitch % 8 == 0 && dpool_pitch % 8 == 0,
                "aau_bn_bwd_reduce: pitches must be multiples of 8");
    const CGMap mp(C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    if (dpool) {
        AAU_REQUIRE(H % 2 == 0 && W % 2 == 0, "aau_bn_bwd_reduce: pooled source needs even H, W");
        AAU_REQUIRE(drop_p == 0.f, "aau_bn_bwd_reduce: dropout and pooling do not combine");
        AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_bn_bwd_reduce: pixel count overflows int32");
        const int64_t items = (int64_t)N * (H / 2) * (W / 2);
        int64_t blocks = (items + mp.PL * 4 - 1) / (mp.PL * 4);
        if (blocks > 2048) blocks = 2048;
        int64_t ipb = (items + blocks - 1) / blocks;
        blocks = (items + ipb - 1) / ipb;
        if (next_traversal()) ipb = -ipb;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                           z_pitch, dy, dy_pitch, dpool, dpool_pitch, dz, dz_pitch, scale, shift, save_mean,
                           save_invstd, red, N, H, W, C, relu, drop_p, drop_seed, ipb, ws);
        red_fold_launch(ws, 2 * C, (int)blocks, red, 2 * C, nullptr, 0, nullptr, (hipStream_t)stream);
    } else {
        const int64_t items = (int64_t)N * H * W;
        int64_t blocks = (items + mp.PL * 8 - 1) / (mp.PL * 8);
        int64_t cap = 1024;   // each workgroup ends with two block reductions + 2C replica atomics: 1024 measured -0.04 ms/step vs 2048
        if (const char* e = getenv("AAU_RED_CAP")) cap = atoi(e);   // experiment
        if (blocks > cap) blocks = cap;
        int64_t ipb = (items + blocks - 1) / blocks;
        blocks = (items + ipb - 1) / ipb;
        if (next_traversal()) ipb = -ipb;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                           z_pitch, dy, dy_pitch, dpool, dpool_pitch, dz, dz_pitch, scale, shift, save_mean,
                           save_invstd, red, N, H, W, C, relu, drop_p, drop_seed, ipb, ws);
        red_fold_launch(ws, 2 * C, (int)blocks, red, 2 * C, nullptr, 0, nullptr, (hipStream_t)stream);
    }
    return check_launch("aau_bn_bwd_reduce");
}

extern "C" int aau_bn_bwd_apply(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch, const float* gamma,
                                const float* save_mean, const float* save_invstd, const float* red, float* dgamma,
                                float* dbeta, int64_t M, int C, const aau_bf16* dy, int dy_pitch, const float* scale,
                                const float* shift, int relu, float drop_p, const uint64_t* drop_seed, void* stream) {
    AAU_REQUIRE(z && dz && gamma && save_mean && save_invstd && red && M > 0, "aau_bn_bwd_apply: bad args");
    AAU_REQUIRE(!dy || (scale && shift && dy_pitch % 8 == 0), "aau_bn_bwd_apply: dy needs scale/shift and an aligned pitch");
    CHK_C("aau_bn_bwd_apply", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0, "aau_bn_bwd_apply: pitches must be multiples of 8");
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, z, z_pitch, dz,
                       dz_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, M, C, dy, dy_pitch, scale, shift,
                       relu, drop_p, drop_seed, (const float*)nullptr, (const float*)nullptr, ppb);
    return check_launch("aau_bn_bwd_apply");
}

extern "C" int aau_bn_bwd_apply_pool(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch, const float* gamma,
                                     const float* save_mean, const float* save_invstd, const float* red, float* dgamma,
                                     float* dbeta, int N, int H, int W, int C, const aau_bf16* dy, int dy_pitch,
                                     const aau_bf16* dpool, int dpool_pitch, const float* scale, const float* shift,
                                     int relu, void* stream) {
    AAU_REQUIRE(z && dz && gamma && save_mean && save_invstd && red && dpool && scale && shift && N > 0 && H > 0 && W > 0,
                "aau_bn_bwd_apply_pool: bad args");
    AAU_REQUIRE(H % 2 == 0 && W % 2 == 0, "aau_bn_bwd_apply_pool: H=%d W=%d must be even", H, W);
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_bn_bwd_apply_pool: pixel count overflows int32");
    CHK_C("aau_bn_bwd_apply_pool", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0 && dy_pitch % 8 == 0 && dpool_pitch % 8 == 0,
                "aau_bn_bwd_apply_pool: pitches must be multiples of 8");
    const CGMap mp(C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    const int64_t items = (int64_t)N * (H / 2) * (W / 2);
    int64_t blocks = (items + mp.PL * 2 - 1) / (mp.PL * 2);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    int64_t ipb = (items + blocks - 1) / blocks;
    blocks = (items + ipb - 1) / ipb;
    if (next_traversal()) ipb = -ipb;
    hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, z,
                       z_pitch, dz, dz_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, N, H, W, C, dy, dy_pitch,
                       dpool, dpool_pitch, scale, shift, relu, ipb);
    return check_launch("aau_bn_bwd_apply_pool");
}

extern "C" int aau_bn_bwd_apply_rank1(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch, const float* gamma,
                                      const float* save_mean, const float* save_invstd, const float* red, float* dgamma,
                                      float* dbeta, int64_t M, int C, const float* dlogits, const float* w_out,
                                      const float* scale, const float* shift, void* stream) {
    AAU_REQUIRE(z && dz && gamma && save_mean && save_invstd && red && dlogits && w_out && scale && shift && M > 0,
                "aau_bn_bwd_apply_rank1: bad args");
    CHK_C("aau_bn_bwd_apply_rank1", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0, "aau_bn_bwd_apply_rank1: pitches must be multiples of 8");
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, z, z_pitch, dz,
                       dz_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, M, C, (const unsigned short*)nullptr, 0,
                       scale, shift, 1, 0.f, (const uint64_t*)nullptr, dlogits, w_out, ppb);
    return check_launch("aau_bn_bwd_apply_rank1");
}

extern "C" int aau_bn_bwd_apply_conv1(const aau_bf16* z, int z_pitch, const float* gamma, const float* save_mean,
                                      const float* save_invstd, const float* red, float* dgamma, float* dbeta, int N,
                                      int H, int W, int C, const aau_bf16* dy, int dy_pitch, const float* scale,
                                      const float* shift, const float* x, const float* w, float* dw, float* ws,
                                      void* stream) {
    AAU_REQUIRE((z || w) && gamma && save_mean && save_invstd && red && dy && scale && shift && x && dw && ws && N > 0 &&
                    H > 0 && W > 0, "aau_bn_bwd_apply_conv1: bad args");
    AAU_REQUIRE((int64_t)H * W < 0x7fffffff, "aau_bn_bwd_apply_conv1: image too large");
    CHK_C("aau_bn_bwd_apply_conv1", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dy_pitch % 8 == 0, "aau_bn_bwd_apply_conv1: pitches must be multiples of 8");
    const int64_t M = (int64_t)N * H * W;
    AAU_REQUIRE(M < 0x7fffffff, "aau_bn_bwd_apply_conv1: pixel count overflows int32");
    ProfScope prof(2, 2.0 * M * 9.0 * C, (hipStream_t)stream);
    int64_t blocks, ppb;
    AAU_REQUIRE(C <= 1024, "aau_bn_bwd_apply_conv1: C=%d too wide for one workgroup", C);
    rows_split(M, 256 / (C >> 2), &blocks, &ppb);
    const int ppb_signed = next_traversal() ? -(int)ppb : (int)ppb;
    const int PL4 = 256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1;
    const size_t base_lds = (size_t)(2 * C + 9 * C + PL4 * 9 * C) * sizeof(float);
    const size_t win = (size_t)(ppb + 2 * (int64_t)W + 2) * sizeof(float);
    if (base_lds + win <= 60 * 1024 && !getenv("AAU_CONV1_NOSTAGE"))
        hipLaunchKernelGGL(bn_bwd_apply_conv1_kernel<true>, dim3((unsigned)blocks), dim3(256), base_lds + win,
                           (hipStream_t)stream, z, z_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, (int)M, C, dy,
                           dy_pitch, scale, shift, x, H, W, ws, w, ppb_signed);
    else
        hipLaunchKernelGGL(bn_bwd_apply_conv1_kernel<false>, dim3((unsigned)blocks), dim3(256), base_lds,
                           (hipStream_t)stream, z, z_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, (int)M, C, dy,
                           dy_pitch, scale, shift, x, H, W, ws, w, ppb_signed);
    red_fold_launch(ws, C * 9, (int)blocks, nullptr, 0, dw, C * 9, nullptr, (hipStream_t)stream);
    return check_launch("aau_bn_bwd_apply_conv1");
}

// ---- several same-width BatchNorm layers per launch (aau.h: *_multi) ----
static int bn_multi_table(const char* fn, const void* const* tab, int n, int per, BnMulti& out) {
    AAU_REQUIRE(tab && n >= 1 && n <= BN_MULTI_MAX && per <= 12, "%s: %d layers (1..%d)", fn, n, BN_MULTI_MAX);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < per; ++k) out.p[i][k] = tab[i * per + k];
    return AAU_OK;
}

extern "C" int aau_bn_finalize_multi(int n, const void* const* tab, int64_t stats_bytes, int C, int64_t count, float eps, float momentum,
                                     void* stream) {
    BnMulti a;
    if (int rc = bn_multi_table("aau_bn_finalize_multi", tab, n, 10, a)) return rc;
    AAU_REQUIRE(C > 0 && count > 0, "aau_bn_finalize_multi: C=%d count=%lld", C, (long long)count);
    for (int i = 0; i < n; ++i) {
        AAU_REQUIRE(a.p[i][0] && a.p[i][1] && a.p[i][2] && a.p[i][6] && a.p[i][7] && a.p[i][8] && a.p[i][9], "aau_bn_finalize_multi: null pointer in layer %d", i);
        AAU_CHECK_STAT("aau_bn_finalize_multi", a.p[i][0], stats_bytes, C);
    }
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3((C + 7) / 8, n), dim3(256), 0, (hipStream_t)stream, a, C, (float)count, eps, momentum);
    return check_launch("aau_bn_finalize_multi");
}

extern "C" int aau_bn_act_multi(int n, const void* const* tab, int z_pitch, int y_pitch, int64_t M, int C, int relu, void* stream) {
    BnMulti a;
    if (int rc = bn_multi_table("aau_bn_act_multi", tab, n, 4, a)) return rc;
    CHK_C("aau_bn_act_multi", C);
    AAU_REQUIRE(M > 0 && z_pitch % 8 == 0 && y_pitch % 8 == 0, "aau_bn_act_multi: bad args");
    for (int i = 0; i < n; ++i) AAU_REQUIRE(a.p[i][0] && a.p[i][1] && a.p[i][2] && a.p[i][3], "aau_bn_act_multi: null pointer in layer %d", i);
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(bn_act_multi_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, a, z_pitch, y_pitch, M, C, relu, ppb);
    return check_launch("aau_bn_act_multi");
}

extern "C" int aau_bn_bwd_reduce_multi(int n, const void* const* tab, int z_pitch, int dy_pitch, int N, int H, int W, int C, int relu,
                                       void* stream) {
    BnMulti a;
    if (int rc = bn_multi_table("aau_bn_bwd_reduce_multi", tab, n, 8, a)) return rc;
    CHK_C("aau_bn_bwd_reduce_multi", C);
    AAU_REQUIRE(N > 0 && H > 0 && W > 0 && z_pitch % 8 == 0 && dy_pitch % 8 == 0, "aau_bn_bwd_reduce_multi: bad args");
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) AAU_REQUIRE(a.p[i][k], "aau_bn_bwd_reduce_multi: null pointer in layer %d", i);
    const CGMap mp(C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    const int64_t items = (int64_t)N * H * W;
    int64_t blocks = (items + mp.PL * 8 - 1) / (mp.PL * 8);
    const int64_t cap = 1024 / n > 64 ? 1024 / n : 64;      // the same number of workgroups as ONE single-layer launch
    if (blocks > cap) blocks = cap;
    int64_t ipb = (items + blocks - 1) / blocks;
    blocks = (items + ipb - 1) / ipb;
    if (next_traversal()) ipb = -ipb;
    hipLaunchKernelGGL(bn_bwd_reduce_multi_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, a, z_pitch, dy_pitch, N, H,
                       W, C, relu, ipb);
    hipLaunchKernelGGL(red_fold_multi_kernel, dim3((unsigned)((2 * C / 4 + 3) / 4), n), dim3(256), 0, (hipStream_t)stream, a, 2 * C,
                       (int)blocks);
    return check_launch("aau_bn_bwd_reduce_multi");
}

extern "C" int aau_bn_bwd_apply_multi(int n, const void* const* tab, int z_pitch, int dz_pitch, int dy_pitch, int64_t M, int C, int relu,
                                      void* stream) {
    BnMulti a;
    if (int rc = bn_multi_table("aau_bn_bwd_apply_multi", tab, n, 11, a)) return rc;
    CHK_C("aau_bn_bwd_apply_multi", C);
    AAU_REQUIRE(M > 0 && z_pitch % 8 == 0 && dz_pitch % 8 == 0 && dy_pitch % 8 == 0, "aau_bn_bwd_apply_multi: bad args");
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 11; ++k) AAU_REQUIRE(a.p[i][k] || k == 6 || k == 7, "aau_bn_bwd_apply_multi: null pointer in layer %d", i);
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(bn_bwd_apply_multi_kernel, dim3((unsigned)blocks, n), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, a, z_pitch,
                       dz_pitch, dy_pitch, M, C, relu, ppb);
    return check_launch("aau_bn_bwd_apply_multi");
}

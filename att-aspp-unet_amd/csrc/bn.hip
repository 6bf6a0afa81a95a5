// BatchNorm2d / ReLU / MaxPool2d(2) / Dropout forward and backward around the MFMA
// convolutions (pipeline:64, the BNs of :71-90, MaxPool2d :115-118, Dropout :79).
// All HBM-bound: every access is a 16-byte vector of 8 bf16 channels, a thread keeps a
// fixed channel group so per-channel reductions live in registers, are combined through
// LDS per block and leave the block as one fp32 atomic per channel into one of
// AAU_STAT_REPLICAS accumulator copies (spreads same-address contention).
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
__global__ void bn_finalize_kernel(const float* stats, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, int64_t* nbt, float* scale, float* shift,
                                   float* smean, float* sinvstd, int C, float count, float eps, float mom) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < AAU_STAT_REPLICAS; ++r) {
        s1 += stats[(size_t)r * 2 * C + c];
        s2 += stats[(size_t)r * 2 * C + C + c];
    }
    const float mean = s1 / count;
    const float var = fmaxf(s2 / count - mean * mean, 0.f);
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
__global__ __launch_bounds__(256) void bn_act_kernel(const unsigned short* z, int zp, unsigned short* y, int yp,
                                                     const float* scale, const float* shift, int64_t M, int C,
                                                     int relu, int64_t bhw, float drop_p, uint64_t seed,
                                                     int64_t ppb) {
    const CGMap mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float sc[8], sh[8];
    ldf8(scale + c, sc);
    ldf8(shift + c, sh);
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int64_t m0 = (int64_t)blockIdx.x * ppb, m1 = min(M, m0 + ppb);
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

__global__ __launch_bounds__(256) void maxpool2_kernel(const unsigned short* y, int yp, unsigned short* p, int pp,
                                                       int N, int H, int W, int C) {
    const int CG = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * CG;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t mo = v / CG;
        const int c = (int)(v - mo * CG) * 8;
        const int xo = (int)(mo % Wo);
        const int64_t t = mo / Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
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
    const int64_t i0 = (int64_t)blockIdx.x * ipb, i1 = min(total, i0 + ipb);
    for (int64_t mo = i0 + pl; mo < i1; mo += mp.PL) {
        const int xo = (int)(mo % Wo);
        const int64_t t = mo / Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
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
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const unsigned short* z, int zp, const unsigned short* dy, int dyp, const unsigned short* dpool, int dpp,
    unsigned short* dz, int dzp, const float* scale, const float* shift, const float* mean, const float* invstd,
    float* red, int N, int H, int W, int C, int relu, float drop_p, uint64_t seed, int64_t items_per_block) {
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
        const int64_t i0 = (int64_t)blockIdx.x * items_per_block;
        const int64_t i1 = min(nitems, i0 + items_per_block);
        for (int64_t it = i0 + pl; it < i1; it += mp.PL) {
            if constexpr (POOL) {
                const int xo = (int)(it % Wo);
                const int64_t t = it / Wo;
                const int yo = (int)(t % Ho);
                const int n = (int)(t / Ho);
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
    float* r = red + (size_t)(blockIdx.x % AAU_STAT_REPLICAS) * 2 * C;
    block_sum8(s1, sred, mp, tid);
    if (tid < mp.CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(r + c + j, s1[j]);
    }
    block_sum8(s2, sred, mp, tid);
    if (tid < mp.CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(r + C + c + j, s2[j]);
    }
}

// ---- backward pass 2: dz = gamma*invstd*(g - mean(g) - zhat*mean(g*zhat)) in place ----
// dy == null: the masked gradient g is read from dz (written by the reduce pass, pooled layers);
// dy != null: g = relu'(z*scale+shift) * dropout(dy) is recomputed here, saving one tensor write + read.
// A thread owns one channel group: its seven per-channel constants live in registers.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* z, int zp, unsigned short* dz, int dzp,
                                                           const float* gamma, const float* mean, const float* invstd,
                                                           const float* red, float* dgamma, float* dbeta, int64_t M,
                                                           int C, const unsigned short* dy, int dyp, const float* scale,
                                                           const float* shift, int relu, float drop_p, uint64_t seed,
                                                           int64_t ppb) {
    extern __shared__ float sm[];   // [2][C] replica sums, computed once per workgroup
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < AAU_STAT_REPLICAS; ++r) {
            a += red[(size_t)r * 2 * C + cc];
            b += red[(size_t)r * 2 * C + C + cc];
        }
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
    if (dy) { ldf8(scale + c, sc); ldf8(shift + c, sh); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k0[j] = gamma[c + j] * is[j];
        k1[j] = sm[c + j] / (float)M;
        k2[j] = sm[C + c + j] / (float)M;
    }
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int64_t m0 = (int64_t)blockIdx.x * ppb, m1 = min(M, m0 + ppb);
    for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
        float zz[8], g[8];
        unpack8(ld16(z + m * zp + c), zz);
        if (dy) {
            unpack8(ld16(dy + m * dyp + c), g);
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

// rows per block so that the grid is ~8 workgroups per CU and every thread gets a few iterations
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

extern "C" int aau_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                               float* save_mean, float* save_invstd, int C, int64_t count, float eps,
                               float momentum, void* stream) {
    AAU_REQUIRE(stats && gamma && beta && scale && shift && save_mean && save_invstd, "aau_bn_finalize: null pointer");
    AAU_REQUIRE(C > 0 && count > 0, "aau_bn_finalize: C=%d count=%lld", C, (long long)count);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, gamma,
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
                          uint64_t drop_seed, void* stream) {
    AAU_REQUIRE(z && y && scale && shift && M > 0, "aau_bn_act: bad args");
    CHK_C("aau_bn_act", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && y_pitch % 8 == 0, "aau_bn_act: pitches must be multiples of 8");
    AAU_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "aau_bn_act: drop_p=%f", drop_p);
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
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
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ipb;
    rows_split((int64_t)N * (H / 2) * (W / 2), CGMap(C).PL, &blocks, &ipb);
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
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       (hipStream_t)stream, y, y_pitch, p, p_pitch, N, H, W, C);
    return check_launch("aau_maxpool2");
}

extern "C" int aau_bn_bwd_reduce(const aau_bf16* z, int z_pitch, const aau_bf16* dy, int dy_pitch,
                                 const aau_bf16* dpool, int dpool_pitch, aau_bf16* dz, int dz_pitch,
                                 const float* scale, const float* shift, const float* save_mean,
                                 const float* save_invstd, float* red, int N, int H, int W, int C, int relu,
                                 float drop_p, uint64_t drop_seed, void* stream) {
    AAU_REQUIRE(z && scale && shift && save_mean && save_invstd && red, "aau_bn_bwd_reduce: null pointer");
    AAU_REQUIRE(dz || !dpool, "aau_bn_bwd_reduce: the pooled form must store the masked gradient (dz != NULL)");
    AAU_REQUIRE(dy || dpool, "aau_bn_bwd_reduce: needs at least one gradient source");
    CHK_C("aau_bn_bwd_reduce", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0 && dy_pitch % 8 == 0 && dpool_pitch % 8 == 0,
                "aau_bn_bwd_reduce: pitches must be multiples of 8");
    const CGMap mp(C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    if (dpool) {
        AAU_REQUIRE(H % 2 == 0 && W % 2 == 0, "aau_bn_bwd_reduce: pooled source needs even H, W");
        AAU_REQUIRE(drop_p == 0.f, "aau_bn_bwd_reduce: dropout and pooling do not combine");
        const int64_t items = (int64_t)N * (H / 2) * (W / 2);
        int64_t blocks = (items + mp.PL * 4 - 1) / (mp.PL * 4);
        if (blocks > 2048) blocks = 2048;
        const int64_t ipb = (items + blocks - 1) / blocks;
        blocks = (items + ipb - 1) / ipb;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                           z_pitch, dy, dy_pitch, dpool, dpool_pitch, dz, dz_pitch, scale, shift, save_mean,
                           save_invstd, red, N, H, W, C, relu, drop_p, drop_seed, ipb);
    } else {
        const int64_t items = (int64_t)N * H * W;
        int64_t blocks = (items + mp.PL * 8 - 1) / (mp.PL * 8);
        if (blocks > 2048) blocks = 2048;
        const int64_t ipb = (items + blocks - 1) / blocks;
        blocks = (items + ipb - 1) / ipb;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                           z_pitch, dy, dy_pitch, dpool, dpool_pitch, dz, dz_pitch, scale, shift, save_mean,
                           save_invstd, red, N, H, W, C, relu, drop_p, drop_seed, ipb);
    }
    return check_launch("aau_bn_bwd_reduce");
}

extern "C" int aau_bn_bwd_apply(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch, const float* gamma,
                                const float* save_mean, const float* save_invstd, const float* red, float* dgamma,
                                float* dbeta, int64_t M, int C, const aau_bf16* dy, int dy_pitch, const float* scale,
                                const float* shift, int relu, float drop_p, uint64_t drop_seed, void* stream) {
    AAU_REQUIRE(z && dz && gamma && save_mean && save_invstd && red && M > 0, "aau_bn_bwd_apply: bad args");
    AAU_REQUIRE(!dy || (scale && shift && dy_pitch % 8 == 0), "aau_bn_bwd_apply: dy needs scale/shift and an aligned pitch");
    CHK_C("aau_bn_bwd_apply", C);
    AAU_REQUIRE(z_pitch % 8 == 0 && dz_pitch % 8 == 0, "aau_bn_bwd_apply: pitches must be multiples of 8");
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks, ppb;
    rows_split(M, CGMap(C).PL, &blocks, &ppb);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, z, z_pitch, dz,
                       dz_pitch, gamma, save_mean, save_invstd, red, dgamma, dbeta, M, C, dy, dy_pitch, scale, shift,
                       relu, drop_p, drop_seed, ppb);
    return check_launch("aau_bn_bwd_apply");
}

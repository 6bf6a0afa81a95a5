// Bandwidth-bound ends of the network that do not belong on MFMA:
//   * the first convolution, Conv2d(1, C, 3, pad 1) on the fp32 frame (K = 9) and its
//     weight gradient (pipeline:113, d1[0]);
//   * out_conv, Conv2d(C, 1, 1) with bias (pipeline:122), forward and backward;
//   * the ASPP image-pool branch reductions / broadcast (pipeline:75-77,82);
//   * per-channel column sums (ConvTranspose2d bias gradient, pipeline:101);
//   * layout / dtype plumbing at the module boundary and the TTA flip (pipeline:336-338).
#include "common.h"

namespace aau {

struct CGMap2 {
    int CG, PL, T;
    __device__ __host__ explicit CGMap2(int C) {
        CG = C >> 3;
        PL = 256 / CG;
        if (PL < 1) PL = 1;
        T = CG * PL;
    }
};

__device__ __forceinline__ void block_sum8b(float acc[8], float* red, const CGMap2& mp, int tid) {
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

// ---- first layer forward: z[m][c] = sum_t x[m+t] * w[c][t]; stats of z ----
// STAGE = true: the workgroup first copies the flat pixel range it needs (its slice plus one image row and one pixel on
// either side) into LDS with coalesced loads; the 9 taps of a pixel are then LDS reads (neighbouring lanes read
// neighbouring words, the C/8 lanes of one pixel broadcast) masked at the image borders.  Without it every thread
// gathered its 9 taps from global memory behind 9 predicated branches and one `vmcnt(0)` per pixel: 40 FMAs per
// exposed load latency, 99 us at bs 8 / 512^2 for a kernel whose bytes take 35.
template <bool STAGE>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* x, const float* w, unsigned short* z,
                                                        long long* stats, int N, int H, int W, int C, int64_t ppb) {
    extern __shared__ float sm[];  // [C*9] weights, [256*8] reduction scratch, then (STAGE) the pixel window
    float* sw = sm;
    float* sred = sm + C * 9;
    float* sx = sred + 256 * 8;
    for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const int64_t M = (int64_t)N * H * W;
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    if constexpr (STAGE) {
        const int64_t base = m0 - W - 1;
        const int n = (int)(m1 - m0) + 2 * W + 2;
        for (int i = tid; i < n; i += 256) {
            const int64_t g = base + i;
            sx[i] = (g >= 0 && g < M) ? x[g] : 0.f;
        }
    }
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (tid < mp.T) {
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
            int xx, yy, nimg;
            decode3(m, W, H, xx, yy, nimg);
            float v[9];
            if constexpr (STAGE) {
                const float* p = sx + (int)(m - m0);          // tap (ky, kx) = p[ky * W + kx]
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = (ky != 0 || yy > 0) && (ky != 2 || yy < H - 1) && (kx != 0 || xx > 0) && (kx != 2 || xx < W - 1);
                        const float t = p[ky * W + kx];
                        v[ky * 3 + kx] = ok ? t : 0.f;
                    }
            } else {
                const int64_t t = (int64_t)nimg * H + yy;
                conv1_taps(x + (t - yy) * W, yy, xx, H, W, v);
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float acc = conv1_dot(v, sw + (c + j) * 9);
                o[j] = acc;
                s1[j] += acc;
                s2[j] += acc * acc;
            }
            if (z) *(u32x4*)(z + m * C + c) = pack8(o);   // z == null: statistics only, consumers recompute z
        }
    }
    if (stats) {   // order-independent fixed-point adds (common.h: stat_add)
        const int rep = (int)(blockIdx.x % AAU_STAT_REPLICAS);
        block_sum8b(s1, sred, mp, tid);
        if (tid < mp.CG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) stat_add(stats, C, rep, 0, c + j, s1[j]);
        }
        block_sum8b(s2, sred, mp, tid);
        if (tid < mp.CG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) stat_add(stats, C, rep, 1, c + j, s2[j]);
        }
    }
}

// ---- first layer: y = relu(bn(conv1(x))) straight from the frame (z is recomputed, never read) ----
__global__ __launch_bounds__(256) void conv1_bn_act_kernel(const float* x, const float* w, unsigned short* y, int yp,
                                                           const float* scale, const float* shift, int N, int H, int W,
                                                           int C, int64_t ppb) {
    extern __shared__ float sw[];  // [C*9]
    for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
    __syncthreads();
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    if (tid >= mp.T) return;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; }
    const int64_t M = (int64_t)N * H * W;
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
        int xx, yy, nimg;
        decode3(m, W, H, xx, yy, nimg);
        const int64_t t = (int64_t)nimg * H + yy;
        float v[9], o[8];
        conv1_taps(x + (t - yy) * W, yy, xx, H, W, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zq = bf2f(f2bf(conv1_dot(v, sw + (c + j) * 9)));   // z as it would have been stored
            o[j] = fmaxf(zq * sc[j] + sh[j], 0.f);
        }
        *(u32x4*)(y + m * yp + c) = pack8(o);
    }
}

// ---- first layer: BatchNorm-backward reduce with z recomputed from the frame ----
__global__ __launch_bounds__(256) void conv1_bn_bwd_reduce_kernel(const float* x, const float* w, const unsigned short* dy,
                                                                  int dyp, const float* scale, const float* shift,
                                                                  const float* mean, const float* invstd, float* red,
                                                                  int N, int H, int W, int C, int64_t ppb, float* ws) {
    extern __shared__ float sm[];  // [C*9] weights, then [256*8] reduction scratch
    float* sw = sm;
    float* sred = sm + C * 9;
    for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
    __syncthreads();
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (tid < mp.T) {
        float sc[8], sh[8], mu[8], is[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
        const int64_t M = (int64_t)N * H * W;
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
            int xx, yy, nimg;
            decode3(m, W, H, xx, yy, nimg);
            const int64_t t = (int64_t)nimg * H + yy;
            float v[9], g[8];
            unpack8(*(const u32x4*)(dy + m * dyp + c), g);
            conv1_taps(x + (t - yy) * W, yy, xx, H, W, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float zq = bf2f(f2bf(conv1_dot(v, sw + (c + j) * 9)));
                const float gv = (zq * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
                s1[j] += gv;
                s2[j] += gv * ((zq - mu[j]) * is[j]);
            }
        }
    }
    // this workgroup's row; red_fold_launch adds the rows in a fixed order (common.h), totals into red [2][C]
    float* r = red_row(ws, 2 * C, blockIdx.x);
    block_sum8b(s1, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + c) = f32x4{s1[0], s1[1], s1[2], s1[3]};
        *(f32x4*)(r + c + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
    }
    block_sum8b(s2, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + C + c) = f32x4{s2[0], s2[1], s2[2], s2[3]};
        *(f32x4*)(r + C + c + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
    }
}

// ---- first layer weight gradient: dw[c][t] += sum_m dz[m][c] * x[m+t] ----
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* x, const unsigned short* dz, float* dw, int N,
                                                          int H, int W, int C, int64_t ppb) {
    extern __shared__ float sacc[];  // [C*9]
    for (int i = threadIdx.x; i < C * 9; i += 256) sacc[i] = 0.f;
    __syncthreads();
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    if (tid < mp.T) {
        float acc[8][9];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[j][k] = 0.f;
        const int64_t M = (int64_t)N * H * W;
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
            int xx, yy, nimg;
            decode3(m, W, H, xx, yy, nimg);
            const int64_t t = (int64_t)nimg * H + yy;
            const float* img = x + (t - yy) * W;
            float g[8];
            unpack8(*(const u32x4*)(dz + m * C + c), g);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y2 = yy + ky - 1, x2 = xx + kx - 1;
                    const float v = ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W)
                                        ? img[(int64_t)y2 * W + x2] : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j][ky * 3 + kx] += g[j] * v;
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(&sacc[(c + j) * 9 + k], acc[j][k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 9; i += 256) atomicAdd(dw + i, sacc[i]);
}

// ---- out_conv forward: logits[m] = b + sum_c w[c]*y[m][c], one thread per pixel ----
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const unsigned short* y, int yp, const float* w,
                                                          const float* b, float* logits, int64_t M, int C) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < C; i += 256) sw[i] = w[i];
    __syncthreads();
    const float bias = b ? b[0] : 0.f;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        float acc = bias;
        for (int c = 0; c < C; c += 8) {
            float f[8];
            unpack8(*(const u32x4*)(y + m * yp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += f[j] * sw[c + j];
        }
        logits[m] = acc;
    }
}

// ---- out_conv backward: dy = dl*w ; dw += sum dl*y ; db += sum dl ----
__global__ __launch_bounds__(256) void outconv_bwd_kernel(const unsigned short* y, int yp, const float* dl,
                                                          const float* w, unsigned short* dy, int dyp, float* ws,
                                                          int64_t M, int C, int64_t ppb) {
    __shared__ float sred[256 * 8];
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float s[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = sb[j] = 0.f;
    if (tid < mp.T) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = w[c + j];
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
#pragma unroll 2
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {      // two pixels' loads in flight per thread
            const float g = dl[m];
            float f[8], o[8];
            unpack8(*(const u32x4*)(y + m * yp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += g * f[j]; o[j] = g * wv[j]; }
            if (cg == 0) sb[0] += g;
            *(u32x4*)(dy + m * dyp + c) = pack8(o);
        }
    }
    float* r = ws + (size_t)(blockIdx.x % AAU_STAT_REPLICAS) * (C + 8);
    block_sum8b(s, sred, mp, tid);
    if (tid < mp.CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(r + c + j, s[j]);
    }
    block_sum8b(sb, sred, mp, tid);
    if (tid == 0) atomicAdd(r + C, sb[0]);
}

// ---- network head, training forward: logits = out_conv(relu(bn(z))) in one pass (pipeline:121-122,126) ----
// The activated tensor of the last ConvBNReLU is consumed by out_conv only, so it is never written: one thread per
// pixel walks the channels in order, y is rounded to bf16 exactly as bn_act would store it and accumulated in the
// order outconv_fwd uses (bitwise the same logits).
__global__ __launch_bounds__(256) void bn_act_outconv_kernel(const unsigned short* z, int zp, const float* scale,
                                                             const float* shift, const float* w, const float* b,
                                                             float* logits, int64_t M, int C, int rev) {
    extern __shared__ float sm[];   // [3][C]: scale, shift, w
    for (int i = threadIdx.x; i < C; i += 256) { sm[i] = scale[i]; sm[C + i] = shift[i]; sm[2 * C + i] = w[i]; }
    __syncthreads();
    const float bias = b ? b[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const int64_t m = rev ? M - 1 - i : i;
        float acc = bias;
        for (int c = 0; c < C; c += 8) {
            float f[8];
            unpack8(*(const u32x4*)(z + m * zp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = bf2f(f2bf(fmaxf(f[j] * sm[c + j] + sm[C + c + j], 0.f)));
                acc += y * sm[2 * C + c + j];
            }
        }
        logits[m] = acc;
    }
}

// Same, for C <= 512: a pixel's channels are spread over CG = C/8 consecutive lanes of ONE wave (16-B loads, contiguous
// across lanes) and the per-pixel dot product is finished with CG-1 wave shuffles.  The one-thread-per-pixel form
// above reads 96-byte-strided pieces and ran at 1.7 TB/s.  The summation order differs from outconv_fwd's, so the
// logits agree to fp32 rounding, not bitwise.
__global__ __launch_bounds__(256) void bn_act_outconv_wave_kernel(const unsigned short* z, int zp, const float* scale,
                                                                  const float* shift, const float* w, const float* b,
                                                                  float* logits, int64_t M, int C, int rev) {
    const int CG = C >> 3;
    const int ppw = 64 / CG;                       // pixels per wave pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane % CG, pw = lane / CG;      // lanes >= ppw*CG idle
    const bool act = pw < ppw;
    const int c = cg * 8;
    float sc[8], sh[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; wv[j] = w[c + j]; }
    const float bias = b ? b[0] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * 4 * ppw;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * ppw; base < M; base += stride) {
        const int64_t i = base + pw;
        const bool ok = act && i < M;
        const int64_t m = rev ? M - 1 - i : i;
        float acc = 0.f;
        if (ok) {
            float f[8];
            unpack8(*(const u32x4*)(z + m * zp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += bf2f(f2bf(fmaxf(f[j] * sc[j] + sh[j], 0.f))) * wv[j];
        }
        float tot = acc;
        for (int o = 1; o < CG; ++o) tot += __shfl_down(acc, o, 64);   // lanes cg == 0 end with the pixel's sum
        if (ok && cg == 0) logits[m] = tot + bias;
    }
}

// ---- network head, backward: the gradient w.r.t. the last activation is rank one (dlogits[m] * w[c]), so it is
// never materialised either.  This pass = outconv_bwd (dw += sum dl*y, db += sum dl, y recomputed from z) +
// bn_bwd_reduce of the last BatchNorm (g = [y>0] * bf16(dl*w), sums of g and g*zhat).  One row [s1 C][s2 C][dw C][db 8]
// per workgroup, summed over workgroups in a fixed order (common.h: red_fold_launch): red [2][C] = totals, dw / db +=.
__global__ __launch_bounds__(256) void bn_bwd_reduce_outconv_kernel(const unsigned short* z, int zp, const float* dl,
                                                                    const float* w, const float* scale,
                                                                    const float* shift, const float* mean,
                                                                    const float* invstd, float* red, float* ws,
                                                                    int64_t M, int C, int64_t ppb) {
    __shared__ float sred[256 * 8];
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float s1[8], s2[8], sw[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = sw[j] = sb[j] = 0.f;
    if (tid < mp.T) {
        float wv[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[j] = w[c + j]; sc[j] = scale[c + j]; sh[j] = shift[c + j]; mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
            const float g = dl[m];
            float f[8];
            unpack8(*(const u32x4*)(z + m * zp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = f[j] * sc[j] + sh[j];
                const float y = bf2f(f2bf(fmaxf(t, 0.f)));
                sw[j] += g * y;
                const float gv = t > 0.f ? bf2f(f2bf(g * wv[j])) : 0.f;   // the stored form of dy is bf16
                s1[j] += gv;
                s2[j] += gv * ((f[j] - mu[j]) * is[j]);
            }
            if (cg == 0) sb[0] += g;
        }
    }
    const int n = 3 * C + 8;
    float* r = red_row(ws, n, blockIdx.x);
    block_sum8b(s1, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + c) = f32x4{s1[0], s1[1], s1[2], s1[3]};
        *(f32x4*)(r + c + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
    }
    block_sum8b(s2, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + C + c) = f32x4{s2[0], s2[1], s2[2], s2[3]};
        *(f32x4*)(r + C + c + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
    }
    block_sum8b(sw, sred, mp, tid);
    if (tid < mp.CG) {
        *(f32x4*)(r + 2 * C + c) = f32x4{sw[0], sw[1], sw[2], sw[3]};
        *(f32x4*)(r + 2 * C + c + 4) = f32x4{sw[4], sw[5], sw[6], sw[7]};
    }
    block_sum8b(sb, sred, mp, tid);
    if (tid == 0) {
        *(f32x4*)(r + 3 * C) = f32x4{sb[0], 0.f, 0.f, 0.f};
        *(f32x4*)(r + 3 * C + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// out[i] += sum over replicas of ws[r][i] (i < n); optionally a second target for element n (bias)
__global__ void fold_replicas_kernel(const float* ws, int stride, float* out, int n, float* out2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n || (i == n && out2 == nullptr)) return;
    float a = 0.f;
    for (int r = 0; r < AAU_STAT_REPLICAS; ++r) a += ws[(size_t)r * stride + i];
    if (i < n) out[i] += a;
    else out2[0] += a;
}

// ---- out[n][c] = alpha * sum_p src[n][p][c]; one block per (image, slab of pixels) ----
__global__ __launch_bounds__(256) void spatial_sum_kernel(const unsigned short* src, int sp, float* acc32, int HW,
                                                          int C, int64_t ppb) {
    __shared__ float sred[256 * 8];
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const int n = blockIdx.y;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (tid < mp.T) {
        const int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = min((int64_t)HW, p0 + ppb);
        for (int64_t p = p0 + pl; p < p1; p += mp.PL) {
            float f[8];
            unpack8(*(const u32x4*)(src + ((int64_t)n * HW + p) * sp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
    }
    block_sum8b(s, sred, mp, tid);
    if (tid < mp.CG) {   // this pixel slab's row of image n: rows [slab][n][C], added in slab order by the next kernel
        float* r = acc32 + ((int64_t)blockIdx.x * gridDim.y + n) * C + c;
        *(f32x4*)r = f32x4{s[0], s[1], s[2], s[3]};
        *(f32x4*)(r + 4) = f32x4{s[4], s[5], s[6], s[7]};
    }
}

// dst[i] = alpha * sum over the nrow slab rows (fixed order) of src[row][i]
__global__ void scale_to_bf16_kernel(const float* src, unsigned short* dst, int64_t n, float alpha, int nrow) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a = 0.f;
        for (int r = 0; r < nrow; ++r) a += src[(int64_t)r * n + i];
        dst[i] = f2bf(a * alpha);
    }
}

// dx[n][p][c] += dpooled[n][c] * inv_hw
__global__ __launch_bounds__(256) void gap_bwd_kernel(const unsigned short* dpooled, unsigned short* dx, int dxp, int N,
                                                      int HW, int C, float inv_hw) {
    const int CG = C >> 3;
    const int64_t total = (int64_t)N * HW * CG;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t m = v / CG;
        const int c = (int)(v - m * CG) * 8;
        const int n = (int)(m / HW);
        float g[8], o[8];
        unpack8(*(const u32x4*)(dpooled + (int64_t)n * C + c), g);
        unpack8(*(const u32x4*)(dx + m * dxp + c), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += g[j] * inv_hw;
        *(u32x4*)(dx + m * dxp + c) = pack8(o);
    }
}

// out[c] += sum_m src[m][c]
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* src, int sp, float* ws, int64_t M, int C,
                                                     int64_t ppb) {
    __shared__ float sred[256 * 8];
    const CGMap2 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (tid < mp.T) {
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
        for (int64_t m = m0 + pl; m < m1; m += mp.PL) {
            float f[8];
            unpack8(*(const u32x4*)(src + m * sp + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
    }
    float* r = ws + (size_t)(blockIdx.x % AAU_STAT_REPLICAS) * (C + 8);
    block_sum8b(s, sred, mp, tid);
    if (tid < mp.CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(r + c + j, s[j]);
    }
}

// ---- plumbing ----
__global__ void f32_to_bf16_kernel(const float* s, unsigned short* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = f2bf(s[i]);
}
__global__ void bf16_to_f32_kernel(const unsigned short* s, float* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = bf2f(s[i]);
}
__global__ void nchw_to_nhwc_kernel(const float* s, unsigned short* d, int dp, int N, int C, int H, int W) {
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t m = i / C;
        const int64_t n = m / HW, p = m - n * HW;
        d[m * dp + c] = f2bf(s[(n * C + c) * HW + p]);
    }
}
__global__ void nhwc_to_nchw_kernel(const unsigned short* s, int sp, float* d, int N, int C, int H, int W) {
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW;
        const int64_t t = i / HW;
        const int c = (int)(t % C);
        const int64_t n = t / C;
        d[i] = bf2f(s[(n * HW + p) * sp + c]);
    }
}
__global__ void hflip_kernel(const float* s, float* d, int64_t rows, int W) {
    const int64_t total = rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        d[i] = s[i - x + (W - 1 - x)];
    }
}
__global__ void tta_merge_kernel(const float* l, const float* lf, float* prob, int64_t rows, int W) {
    const int64_t total = rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const float v = (l[i] + lf[i - x + (W - 1 - x)]) * 0.5f;
        prob[i] = 1.f / (1.f + expf(-v));
    }
}

// Gaussian-weighted blend of overlapping window logits into the full frame:
// out[y][x] = sum_w g(y-wy, x-wx) * l_w[y-wy][x-wx] / sum_w g(...), windows on a regular grid (stride sy, sx)
__global__ void window_blend_kernel(const float* wl, float* out, int H, int W, int win, int stride, int ny, int nx,
                                    float inv_two_sigma2) {
    const int64_t total = (int64_t)H * W;
    const float c = 0.5f * (float)(win - 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        float acc = 0.f, wsum = 0.f;
        for (int iy = 0; iy < ny; ++iy) {
            const int ly = y - iy * stride;
            if (ly < 0 || ly >= win) continue;
            for (int ix = 0; ix < nx; ++ix) {
                const int lx = x - ix * stride;
                if (lx < 0 || lx >= win) continue;
                const float dy = (float)ly - c, dx = (float)lx - c;
                const float g = expf(-(dy * dy + dx * dx) * inv_two_sigma2);
                acc += g * wl[((int64_t)(iy * nx + ix) * win + ly) * win + lx];
                wsum += g;
            }
        }
        out[i] = wsum > 0.f ? acc / wsum : 0.f;
    }
}

static inline int grid1d(int64_t n, int cap = 4096) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
static inline void split_rows(int64_t M, int PL, int min_iters, int max_blocks, int64_t* blocks, int64_t* ppb) {
    int64_t b = (M + (int64_t)PL * min_iters - 1) / ((int64_t)PL * min_iters);
    if (b > max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    *ppb = (M + b - 1) / b;
    *blocks = (M + *ppb - 1) / *ppb;
}

}  // namespace aau

using namespace aau;

#define CHK_C(fn, C) AAU_REQUIRE((C) > 0 && (C) % 8 == 0 && (C) <= 2048, fn ": C=%d must be a multiple of 8 in [8, 2048]", (int)(C))

extern "C" int aau_conv1_fwd(const float* x, const float* w, aau_bf16* z, aau_stat* stats, int64_t stats_bytes, int N, int H, int W, int C,
                             void* stream) {
    AAU_REQUIRE(x && w && (z || stats) && N > 0 && H > 0 && W > 0, "aau_conv1_fwd: bad args");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_conv1_fwd: pixel count overflows int32");
    CHK_C("aau_conv1_fwd", C);
    AAU_CHECK_STAT("aau_conv1_fwd", stats, stats_bytes, C);
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows((int64_t)N * H * W, mp.PL, 16, 4096, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 2.0 * N * H * W * 9.0 * C, (hipStream_t)stream);
    const int64_t win = (ppb < 0 ? -ppb : ppb) + 2 * (int64_t)W + 2;     // staged pixel window (floats)
    if (win <= 12288 && !getenv("AAU_CONV1_NOSTAGE"))
        hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), (C * 9 + 256 * 8 + win) * sizeof(float),
                           (hipStream_t)stream, x, w, z, (long long*)stats, N, H, W, C, ppb);
    else
        hipLaunchKernelGGL(conv1_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), (C * 9 + 256 * 8) * sizeof(float),
                           (hipStream_t)stream, x, w, z, (long long*)stats, N, H, W, C, ppb);
    return check_launch("aau_conv1_fwd");
}

extern "C" int aau_conv1_wgrad(const float* x, const aau_bf16* dz, float* dw, int N, int H, int W, int C,
                               void* stream) {
    AAU_REQUIRE(x && dz && dw && N > 0 && H > 0 && W > 0, "aau_conv1_wgrad: bad args");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_conv1_wgrad: pixel count overflows int32");
    CHK_C("aau_conv1_wgrad", C);
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows((int64_t)N * H * W, mp.PL, 32, 1024, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 2.0 * N * H * W * 9.0 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)blocks), dim3(256), C * 9 * sizeof(float),
                       (hipStream_t)stream, x, dz, dw, N, H, W, C, ppb);
    return check_launch("aau_conv1_wgrad");
}

extern "C" int aau_outconv_fwd(const aau_bf16* y, int y_pitch, const float* w, const float* b, float* logits,
                               int64_t M, int C, void* stream) {
    AAU_REQUIRE(y && w && logits && M > 0, "aau_outconv_fwd: bad args");
    CHK_C("aau_outconv_fwd", C);
    AAU_REQUIRE(y_pitch % 8 == 0, "aau_outconv_fwd: pitch");
    ProfScope prof(2, 2.0 * M * C, (hipStream_t)stream);
    hipLaunchKernelGGL(outconv_fwd_kernel, dim3(grid1d(M)), dim3(256), C * sizeof(float), (hipStream_t)stream, y,
                       y_pitch, w, b, logits, M, C);
    return check_launch("aau_outconv_fwd");
}

extern "C" int aau_outconv_bwd(const aau_bf16* y, int y_pitch, const float* dlogits, const float* w, aau_bf16* dy,
                               int dy_pitch, float* dw, float* db, float* ws, int64_t M, int C, void* stream) {
    AAU_REQUIRE(y && dlogits && w && dy && dw && ws && M > 0, "aau_outconv_bwd: bad args");
    CHK_C("aau_outconv_bwd", C);
    AAU_REQUIRE(y_pitch % 8 == 0 && dy_pitch % 8 == 0, "aau_outconv_bwd: pitch");
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows(M, mp.PL, 16, 2048, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 4.0 * M * C, (hipStream_t)stream);
    zero_f32(ws, (int64_t)AAU_STAT_REPLICAS * (C + 8), (hipStream_t)stream);
    hipLaunchKernelGGL(outconv_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, y_pitch,
                       dlogits, w, dy, dy_pitch, ws, M, C, ppb);
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((C + 256) / 256), dim3(256), 0, (hipStream_t)stream, ws, C + 8, dw, C,
                       db);
    return check_launch("aau_outconv_bwd");
}

static int spatial_reduce(const aau_bf16* src, int sp, aau_bf16* out, int N, int HW, int C, float alpha,
                          float* ws, hipStream_t s) {
    const int64_t need = (int64_t)N * C;
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows(HW, mp.PL, 8, AAU_GAP_WS_ROWS, &blocks, &ppb);
    hipLaunchKernelGGL(spatial_sum_kernel, dim3((unsigned)blocks, N), dim3(256), 0, s, src, sp, ws, HW, C, ppb);
    hipLaunchKernelGGL(scale_to_bf16_kernel, dim3(grid1d(need)), dim3(256), 0, s, ws, out, need, alpha, (int)blocks);
    return check_launch("spatial reduce");
}

extern "C" int aau_gap_fwd(const aau_bf16* x, int x_pitch, aau_bf16* pooled, float* ws, int N, int HW, int C,
                           void* stream) {
    AAU_REQUIRE(x && pooled && ws && N > 0 && HW > 0, "aau_gap_fwd: bad args");
    CHK_C("aau_gap_fwd", C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    return spatial_reduce(x, x_pitch, pooled, N, HW, C, 1.0f / (float)HW, ws, (hipStream_t)stream);
}

extern "C" int aau_spatial_sum(const aau_bf16* src, int src_pitch, aau_bf16* out, float* ws, int N, int HW, int C,
                               void* stream) {
    AAU_REQUIRE(src && out && ws && N > 0 && HW > 0, "aau_spatial_sum: bad args");
    CHK_C("aau_spatial_sum", C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    return spatial_reduce(src, src_pitch, out, N, HW, C, 1.0f, ws, (hipStream_t)stream);
}

extern "C" int aau_gap_bwd(const aau_bf16* dpooled, aau_bf16* dx, int dx_pitch, int N, int HW, int C, void* stream) {
    AAU_REQUIRE(dpooled && dx && N > 0 && HW > 0, "aau_gap_bwd: bad args");
    CHK_C("aau_gap_bwd", C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid1d((int64_t)N * HW * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       dpooled, dx, dx_pitch, N, HW, C, 1.0f / (float)HW);
    return check_launch("aau_gap_bwd");
}

extern "C" int aau_colsum(const aau_bf16* src, int src_pitch, float* out, float* ws, int64_t M, int C, void* stream) {
    AAU_REQUIRE(src && out && ws && M > 0, "aau_colsum: bad args");
    CHK_C("aau_colsum", C);
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows(M, mp.PL, 16, 1024, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 0, (hipStream_t)stream);
    zero_f32(ws, (int64_t)AAU_STAT_REPLICAS * (C + 8), (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, src_pitch, ws, M,
                       C, ppb);
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((C + 256) / 256), dim3(256), 0, (hipStream_t)stream, ws, C + 8, out, C,
                       (float*)nullptr);
    return check_launch("aau_colsum");
}

// out[i] += (sum over the replicas of statistic `which`, channel c_begin + i) of a fixed-point statistics buffer
__global__ void fold_stats_kernel(const long long* stats, int C, int which, int c_begin, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += (float)stat_total(stats, C, which, c_begin + i);
}
// out[i] += total of statistic 0 of channel cA + i of buffer A (+ of channel cB + i of buffer B): 32 lanes per channel, one
// replica each, exact integer limb sums through shuffles (the same value as stat_total, a quarter of its dependent loads)
static_assert(AAU_STAT_REPLICAS == 32, "fold_stats_pair_kernel: one lane of a 32-lane group per replica (threadIdx & 31, shuffle offsets 16..1)");
__global__ __launch_bounds__(256) void fold_stats_pair_kernel(const long long* A, int CA, int cA, const long long* Bs, int CB, int cB,
                                                               int n, float* out) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), r = threadIdx.x & 31;
    long long hi = 0, lo = 0;
    bool poison = false;
    if (i < n) {
        const long long* s = A + (((size_t)r * 2 + 0) * CA + cA + i) * 2;
        hi = s[0]; lo = s[1];
        poison = A[(size_t)AAU_STAT_REPLICAS * 2 * CA * 2] != 0;
        if (Bs) {
            const long long* t = Bs + (((size_t)r * 2 + 0) * CB + cB + i) * 2;
            hi += t[0]; lo += t[1];
            poison |= Bs[(size_t)AAU_STAT_REPLICAS * 2 * CB * 2] != 0;
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64); }
    if (i < n && r == 0) {
        const double v = poison ? __longlong_as_double(0x7ff8000000000000ll)
                                : (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 4503599627370496.0);
        out[i] += (float)v;
    }
}
__global__ void stats_to_f64_kernel(const long long* stats, int C, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * C) out[i] = stat_total(stats, C, i / C, i % C);
}

extern "C" int aau_fold_stats(const aau_stat* stats, int64_t stats_bytes, int C, int which, int c_begin, int n, float* out,
                              void* stream) {
    AAU_REQUIRE(stats && out && C > 0 && (which == 0 || which == 1) && c_begin >= 0 && n > 0 && c_begin + n <= C,
                "aau_fold_stats: bad args");
    AAU_CHECK_STAT("aau_fold_stats", stats, stats_bytes, C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(fold_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)stats, C,
                       which, c_begin, n, out);
    return check_launch("aau_fold_stats");
}
extern "C" int aau_fold_stats_pair(const aau_stat* a, int64_t a_bytes, int CA, int ca_begin, const aau_stat* b, int64_t b_bytes, int CB,
                                   int cb_begin, int n, float* out, void* stream) {
    AAU_REQUIRE(a && out && CA > 0 && ca_begin >= 0 && n > 0 && ca_begin + n <= CA, "aau_fold_stats_pair: bad args (A)");
    AAU_REQUIRE(!b || (CB > 0 && cb_begin >= 0 && cb_begin + n <= CB), "aau_fold_stats_pair: bad args (B)");
    AAU_CHECK_STAT("aau_fold_stats_pair", a, a_bytes, CA);
    AAU_CHECK_STAT("aau_fold_stats_pair", b, b_bytes, b ? CB : 1);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(fold_stats_pair_kernel, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const long long*)a, CA, ca_begin,
                       (const long long*)b, CB, cb_begin, n, out);
    return check_launch("aau_fold_stats_pair");
}
// out fp32 [2][C] = the totals of an aau_stat buffer (the `red` operand of the BatchNorm-backward apply passes)
static __global__ void stats_to_f32_kernel(const long long* stats, int C, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * C) out[i] = (float)stat_total(stats, C, i / C, i % C);
}
extern "C" int aau_stats_to_red(const aau_stat* stats, int64_t stats_bytes, int C, float* red, void* stream) {
    AAU_REQUIRE(stats && red && C > 0, "aau_stats_to_red: bad args");
    AAU_CHECK_STAT("aau_stats_to_red", stats, stats_bytes, C);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(stats_to_f32_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)stats, C, red);
    return check_launch("aau_stats_to_red");
}
extern "C" int aau_stats_to_f64(const aau_stat* stats, int64_t stats_bytes, int C, double* out, void* stream) {
    AAU_REQUIRE(stats && out && C > 0, "aau_stats_to_f64: bad args");
    AAU_CHECK_STAT("aau_stats_to_f64", stats, stats_bytes, C);
    hipLaunchKernelGGL(stats_to_f64_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)stats, C, out);
    return check_launch("aau_stats_to_f64");
}

extern "C" int aau_f32_to_bf16(const float* src, aau_bf16* dst, int64_t n, void* stream) {
    AAU_REQUIRE(src && dst && n > 0, "aau_f32_to_bf16: bad args");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return check_launch("aau_f32_to_bf16");
}
extern "C" int aau_bf16_to_f32(const aau_bf16* src, float* dst, int64_t n, void* stream) {
    AAU_REQUIRE(src && dst && n > 0, "aau_bf16_to_f32: bad args");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return check_launch("aau_bf16_to_f32");
}
extern "C" int aau_nchw_to_nhwc(const float* src, aau_bf16* dst, int dst_pitch, int N, int C, int H, int W,
                                void* stream) {
    AAU_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && dst_pitch >= C, "aau_nchw_to_nhwc: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid1d((int64_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, dst_pitch, N, C, H, W);
    return check_launch("aau_nchw_to_nhwc");
}
extern "C" int aau_nhwc_to_nchw(const aau_bf16* src, int src_pitch, float* dst, int N, int C, int H, int W,
                                void* stream) {
    AAU_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && src_pitch >= C, "aau_nhwc_to_nchw: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid1d((int64_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                       src, src_pitch, dst, N, C, H, W);
    return check_launch("aau_nhwc_to_nchw");
}
extern "C" int aau_hflip_f32(const float* src, float* dst, int N, int H, int W, void* stream) {
    AAU_REQUIRE(src && dst && src != dst && N > 0 && H > 0 && W > 0, "aau_hflip_f32: bad args");
    hipLaunchKernelGGL(hflip_kernel, dim3(grid1d((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (int64_t)N * H, W);
    return check_launch("aau_hflip_f32");
}
extern "C" int aau_tta_merge(const float* l, const float* l_flipped, float* prob, int N, int H, int W, void* stream) {
    AAU_REQUIRE(l && l_flipped && prob && N > 0 && H > 0 && W > 0, "aau_tta_merge: bad args");
    hipLaunchKernelGGL(tta_merge_kernel, dim3(grid1d((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, l,
                       l_flipped, prob, (int64_t)N * H, W);
    return check_launch("aau_tta_merge");
}

extern "C" int aau_window_blend(const float* win_logits, float* out, int H, int W, int win, int stride, int ny, int nx,
                                float sigma, void* stream) {
    AAU_REQUIRE(win_logits && out && H > 0 && W > 0 && win > 0 && stride > 0 && ny > 0 && nx > 0 && sigma > 0.f,
                "aau_window_blend: bad args");
    AAU_REQUIRE((ny - 1) * stride + win >= H && (nx - 1) * stride + win >= W, "aau_window_blend: windows do not cover the frame");
    hipLaunchKernelGGL(window_blend_kernel, dim3(grid1d((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, win_logits,
                       out, H, W, win, stride, ny, nx, 1.0f / (2.f * sigma * sigma));
    return check_launch("aau_window_blend");
}

// out[i] += sum over the AAU_STAT_REPLICAS replicas of ws[r * stride + i], i < n.  Used to read a per-channel sum out of
// the statistics a conv epilogue accumulated (e.g. the ConvTranspose2d bias gradient = column sums of the gradient
// the preceding data-gradient conv produced: no separate pass over that tensor).
extern "C" int aau_fold_replicas(const float* ws, int stride, float* out, int n, void* stream) {
    AAU_REQUIRE(ws && out && n > 0 && stride >= n, "aau_fold_replicas: bad args");
    hipLaunchKernelGGL(aau::fold_replicas_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, stride, out, n,
                       (float*)nullptr);
    return aau::check_launch("aau_fold_replicas");
}

extern "C" int aau_bn_act_outconv(const aau_bf16* z, int z_pitch, const float* scale, const float* shift, const float* w,
                                  const float* b, float* logits, int64_t M, int C, void* stream) {
    AAU_REQUIRE(z && scale && shift && w && logits && M > 0, "aau_bn_act_outconv: bad args");
    CHK_C("aau_bn_act_outconv", C);
    AAU_REQUIRE(z_pitch % 8 == 0, "aau_bn_act_outconv: pitch");
    ProfScope prof(2, 4.0 * M * C, (hipStream_t)stream);
    const int rev = next_traversal();
    if (C <= 512) {
        const int ppw = 64 / (C >> 3);
        hipLaunchKernelGGL(bn_act_outconv_wave_kernel, dim3(grid1d((M + ppw - 1) / ppw * 64, 8192)), dim3(256), 0,
                           (hipStream_t)stream, z, z_pitch, scale, shift, w, b, logits, M, C, rev);
    } else {
        hipLaunchKernelGGL(bn_act_outconv_kernel, dim3(grid1d(M)), dim3(256), 3 * C * sizeof(float), (hipStream_t)stream, z,
                           z_pitch, scale, shift, w, b, logits, M, C, rev);
    }
    return check_launch("aau_bn_act_outconv");
}

extern "C" int aau_bn_bwd_reduce_outconv(const aau_bf16* z, int z_pitch, const float* dlogits, const float* w,
                                         const float* scale, const float* shift, const float* save_mean,
                                         const float* save_invstd, float* red, float* dw, float* db, float* ws, int64_t M,
                                         int C, void* stream) {
    AAU_REQUIRE(z && dlogits && w && scale && shift && save_mean && save_invstd && red && dw && ws && M > 0,
                "aau_bn_bwd_reduce_outconv: bad args");
    AAU_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)red & 15) == 0, "aau_bn_bwd_reduce_outconv: red / ws must be 16-byte aligned");
    CHK_C("aau_bn_bwd_reduce_outconv", C);
    AAU_REQUIRE(z_pitch % 8 == 0, "aau_bn_bwd_reduce_outconv: pitch");
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows(M, mp.PL, 16, 2048, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 8.0 * M * C, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_reduce_outconv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, z_pitch,
                       dlogits, w, scale, shift, save_mean, save_invstd, red, ws, M, C, ppb);
    red_fold_launch(ws, 3 * C + 8, (int)blocks, red, 2 * C, dw, C, db, (hipStream_t)stream);
    return check_launch("aau_bn_bwd_reduce_outconv");
}

extern "C" int aau_conv1_bn_act(const float* x, const float* w, aau_bf16* y, int y_pitch, const float* scale,
                                const float* shift, int N, int H, int W, int C, void* stream) {
    AAU_REQUIRE(x && w && y && scale && shift && N > 0 && H > 0 && W > 0, "aau_conv1_bn_act: bad args");
    CHK_C("aau_conv1_bn_act", C);
    AAU_REQUIRE(y_pitch % 8 == 0, "aau_conv1_bn_act: pitch");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_conv1_bn_act: pixel count overflows int32");
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows((int64_t)N * H * W, mp.PL, 16, 4096, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 2.0 * N * H * W * 9.0 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(conv1_bn_act_kernel, dim3((unsigned)blocks), dim3(256), C * 9 * sizeof(float), (hipStream_t)stream, x, w,
                       y, y_pitch, scale, shift, N, H, W, C, ppb);
    return check_launch("aau_conv1_bn_act");
}

extern "C" int aau_conv1_bn_bwd_reduce(const float* x, const float* w, const aau_bf16* dy, int dy_pitch,
                                       const float* scale, const float* shift, const float* save_mean,
                                       const float* save_invstd, float* red, int N, int H, int W, int C, float* ws,
                                       void* stream) {
    AAU_REQUIRE(x && w && dy && scale && shift && save_mean && save_invstd && red && ws && N > 0 && H > 0 && W > 0,
                "aau_conv1_bn_bwd_reduce: bad args");
    AAU_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)red & 15) == 0, "aau_conv1_bn_bwd_reduce: red / ws must be 16-byte aligned");
    CHK_C("aau_conv1_bn_bwd_reduce", C);
    AAU_REQUIRE(dy_pitch % 8 == 0, "aau_conv1_bn_bwd_reduce: pitch");
    AAU_REQUIRE((int64_t)N * H * W < 0x7fffffff, "aau_conv1_bn_bwd_reduce: pixel count overflows int32");
    const CGMap2 mp(C);
    int64_t blocks, ppb;
    split_rows((int64_t)N * H * W, mp.PL, 32, 2048, &blocks, &ppb);
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 2.0 * N * H * W * 9.0 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(conv1_bn_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), (C * 9 + 256 * 8) * sizeof(float),
                       (hipStream_t)stream, x, w, dy, dy_pitch, scale, shift, save_mean, save_invstd, red, N, H, W, C, ppb, ws);
    red_fold_launch(ws, 2 * C, (int)blocks, red, 2 * C, nullptr, 0, nullptr, (hipStream_t)stream);
    return check_launch("aau_conv1_bn_bwd_reduce");
}

// Attention gate (pipeline:85-92): x * sigmoid(BN1(psi(relu(BN(Wg g) + BN(Wx x))))).
// The two 1x1 convolutions run on the MFMA kernel; everything between them is one
// HBM pass per BatchNorm barrier (training-mode statistics force the cuts):
//   gate_psi   : s = relu(bn(zg)+bn(zx)); psi_pre = <wpsi, s>; sum / sumsq of psi_pre
//   gate_apply : alpha = sigmoid(bn1(psi_pre)); out = x * alpha (written into the concat slot)
// and the mirrored three backward passes.
#include <stdlib.h>
#include "common.h"

namespace aau {

struct CGMap3 {
    int CG, PL, T;
    __device__ __host__ explicit CGMap3(int C) {
        CG = C >> 3;
        PL = 256 / CG;
        if (PL < 1) PL = 1;
        T = CG * PL;
    }
};
__device__ __forceinline__ void block_sum8c(float acc[8], float* red, const CGMap3& mp, int tid) {
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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide sum of one float per thread, result on thread 0
__device__ __forceinline__ float block_sum1(float v, float* s4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}
__device__ __forceinline__ void ldf8g(const float* p, float f[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}

// channel-group threads (8 channels each); the per-pixel dot product is combined through LDS
__global__ __launch_bounds__(256) void gate_psi_kernel(const unsigned short* zg, const unsigned short* zx,
                                                       const float* sg, const float* hg, const float* sx,
                                                       const float* hx, const float* wpsi, float* psi_pre,
                                                       long long* stats, int64_t M, int F, int64_t ppb) {
    __shared__ float part[256];
    __shared__ float s4[4];
    const CGMap3 mp(F);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const bool active = tid < mp.T;
    float v_sg[8], v_hg[8], v_sx[8], v_hx[8], v_w[8];
    if (active) { ldf8g(sg + c, v_sg); ldf8g(hg + c, v_hg); ldf8g(sx + c, v_sx); ldf8g(hx + c, v_hx); ldf8g(wpsi + c, v_w); }
    float t1 = 0.f, t2 = 0.f;
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    for (int64_t mb = m0; mb < m1; mb += mp.PL) {
        const int64_t m = mb + pl;
        float acc = 0.f;
        if (active && m < m1) {
            float a[8], b[8];
            unpack8(*(const u32x4*)(zg + m * F + c), a);
            unpack8(*(const u32x4*)(zx + m * F + c), b);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc += fmaxf(a[j] * v_sg[j] + v_hg[j] + b[j] * v_sx[j] + v_hx[j], 0.f) * v_w[j];
        }
        part[tid] = acc;
        __syncthreads();
        if (active && cg == 0 && m < m1) {
            float tot = 0.f;
            for (int k = 0; k < mp.CG; ++k) tot += part[tid + k];
            psi_pre[m] = tot;
            t1 += tot;
            t2 += tot * tot;
        }
        __syncthreads();
    }
    if (stats) {
        const float a = block_sum1(t1, s4);
        const float b = block_sum1(t2, s4);
        if (tid == 0) {
            stat_add(stats, 1, (int)(blockIdx.x % AAU_STAT_REPLICAS), 0, 0, a);
            stat_add(stats, 1, (int)(blockIdx.x % AAU_STAT_REPLICAS), 1, 0, b);
        }
    }
}

__global__ __launch_bounds__(256) void gate_apply_kernel(const unsigned short* x, int xp, const float* psi_pre,
                                                         const float* scale1, const float* shift1, float* alpha,
                                                         unsigned short* out, int op, int64_t M, int C) {
    const int CG = C >> 3;
    const int64_t total = M * CG;
    const float sc = scale1[0], sh = shift1[0];
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t m = v / CG;
        const int cgi = (int)(v - m * CG);
        const float a = 1.f / (1.f + expf(-(psi_pre[m] * sc + sh)));
        if (cgi == 0 && alpha) alpha[m] = a;
        float f[8];
        unpack8(*(const u32x4*)(x + m * xp + cgi * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= a;
        *(u32x4*)(out + m * op + cgi * 8) = pack8(f);
    }
}

// channel-group threads: dx = dout*alpha ; dq = <dout, x> * alpha(1-alpha) ; rows of (sum dq, sum dq*psihat) -> red1
__global__ __launch_bounds__(256) void gate_bwd1_kernel(const unsigned short* dout, int dop, const unsigned short* x,
                                                        int xp, const float* alpha, const float* psi_pre,
                                                        const float* mean1, const float* invstd1,
                                                        unsigned short* dx, int dxp, float* dq, float* ws,
                                                        int64_t M, int C, int64_t ppb) {
    __shared__ float part[256];
    __shared__ float s4[4];
    const CGMap3 mp(C);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const bool active = tid < mp.T;
    const float mu = mean1[0], is = invstd1[0];
    float t1 = 0.f, t2 = 0.f;
    const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
    for (int64_t mb = m0; mb < m1; mb += mp.PL) {
        const int64_t m = mb + pl;
        float dot = 0.f, a = 0.f;
        if (active && m < m1) {
            a = alpha[m];
            float g[8], xv[8];
            unpack8(*(const u32x4*)(dout + m * dop + c), g);
            unpack8(*(const u32x4*)(x + m * xp + c), xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dot += g[j] * xv[j]; g[j] *= a; }
            *(u32x4*)(dx + m * dxp + c) = pack8(g);
        }
        part[tid] = dot;
        __syncthreads();
        if (active && cg == 0 && m < m1) {
            float tot = 0.f;
            for (int k = 0; k < mp.CG; ++k) tot += part[tid + k];
            const float q = tot * a * (1.f - a);
            dq[m] = q;
            t1 += q;
            t2 += q * (psi_pre[m] - mu) * is;
        }
        __syncthreads();
    }
    const float a = block_sum1(t1, s4);
    const float b = block_sum1(t2, s4);
    if (tid == 0) *(f32x4*)red_row(ws, 4, blockIdx.x) = f32x4{a, b, 0.f, 0.f};   // this workgroup's row; red_fold adds the rows in order
}

// channel-group threads: ds, dwpsi, per-channel sums for the two branch BNs
__global__ __launch_bounds__(256) void gate_bwd2_kernel(
    const float* dq, const float* psi_pre, const float* red1, const float* gamma1, const float* mean1,
    const float* invstd1, const unsigned short* zg, const unsigned short* zx, const float* sg, const float* hg,
    const float* sx, const float* hx, const float* mean_g, const float* invstd_g, const float* mean_x,
    const float* invstd_x, const float* wpsi, unsigned short* ds, float* ws,
    float* dgamma1, float* dbeta1, int64_t M, int F, int64_t ppb) {
    __shared__ float sred[256 * 8];
    const CGMap3 mp(F);
    const int tid = threadIdx.x;
    const int cg = tid % mp.CG, pl = tid / mp.CG, c = cg * 8;
    const float r1 = red1[0], r2 = red1[1];      // totals of step 1
    if (blockIdx.x == 0 && tid == 0) {
        if (dbeta1) dbeta1[0] += r1;
        if (dgamma1) dgamma1[0] += r2;
    }
    const float k0 = gamma1[0] * invstd1[0], k1 = r1 / (float)M, k2 = r2 / (float)M;
    const float mu1 = mean1[0], is1 = invstd1[0];
    float a_w[8], a_s[8], a_g[8], a_x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_w[j] = a_s[j] = a_g[j] = a_x[j] = 0.f;
    if (tid < mp.T) {
        float v_sg[8], v_hg[8], v_sx[8], v_hx[8], v_w[8], v_mg[8], v_ig[8], v_mx[8], v_ix[8];
        ldf8g(sg + c, v_sg); ldf8g(hg + c, v_hg); ldf8g(sx + c, v_sx); ldf8g(hx + c, v_hx); ldf8g(wpsi + c, v_w);
        ldf8g(mean_g + c, v_mg); ldf8g(invstd_g + c, v_ig); ldf8g(mean_x + c, v_mx); ldf8g(invstd_x + c, v_ix);
        const int64_t m0 = slice_begin(ppb), m1 = min(M, m0 + ppb);
        auto body = [&](int64_t m, float pp, float dqv, const u32x4& qa, const u32x4& qb) {
            const float ph = (pp - mu1) * is1;
            const float dp = k0 * (dqv - k1 - ph * k2);
            float a[8], b[8], o[8];
            unpack8(qa, a);
            unpack8(qb, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float s = a[j] * v_sg[j] + v_hg[j] + b[j] * v_sx[j] + v_hx[j];
                const float sr = fmaxf(s, 0.f);
                const float d = s > 0.f ? dp * v_w[j] : 0.f;
                a_w[j] += dp * sr;
                a_s[j] += d;
                a_g[j] += d * (a[j] - v_mg[j]) * v_ig[j];
                a_x[j] += d * (b[j] - v_mx[j]) * v_ix[j];
                o[j] = d;
            }
            *(u32x4*)(ds + m * F + c) = pack8(o);
        };
        // the loop is latency bound: keep the loads of two pixels in flight per thread
        int64_t m = m0 + pl;
        for (; m + mp.PL < m1; m += 2 * mp.PL) {
            const int64_t mb = m + mp.PL;
            const float p0 = psi_pre[m], d0 = dq[m], p1 = psi_pre[mb], d1 = dq[mb];
            const u32x4 a0 = *(const u32x4*)(zg + m * F + c), b0 = *(const u32x4*)(zx + m * F + c);
            const u32x4 a1 = *(const u32x4*)(zg + mb * F + c), b1 = *(const u32x4*)(zx + mb * F + c);
            body(m, p0, d0, a0, b0);
            body(mb, p1, d1, a1, b1);
        }
        if (m < m1) body(m, psi_pre[m], dq[m], *(const u32x4*)(zg + m * F + c), *(const u32x4*)(zx + m * F + c));
    }
    // this workgroup's row [dpsi*s | d | d*zhat_g | d*zhat_x][F]; red_fold adds the rows in a fixed order into `tot`
    float* row = red_row(ws, 4 * F, blockIdx.x);
    auto put = [&](const float a[8], int k) {
        *(f32x4*)(row + k * F + c) = f32x4{a[0], a[1], a[2], a[3]};
        *(f32x4*)(row + k * F + c + 4) = f32x4{a[4], a[5], a[6], a[7]};
    };
    block_sum8c(a_w, sred, mp, tid);
    if (tid < mp.CG) put(a_w, 0);
    block_sum8c(a_s, sred, mp, tid);
    if (tid < mp.CG) put(a_s, 1);
    block_sum8c(a_g, sred, mp, tid);
    if (tid < mp.CG) put(a_g, 2);
    block_sum8c(a_x, sred, mp, tid);
    if (tid < mp.CG) put(a_x, 3);
}

__global__ __launch_bounds__(256) void gate_bwd3_kernel(
    const unsigned short* ds, const unsigned short* zg, const unsigned short* zx, const float* gamma_g,
    const float* mean_g, const float* invstd_g, const float* gamma_x, const float* mean_x,
    const float* invstd_x, const float* tot, unsigned short* dzg, unsigned short* dzx, float* dgamma_g,
    float* dbeta_g, float* dgamma_x, float* dbeta_x, float* dwpsi, int64_t M, int F) {
    extern __shared__ float sm[];  // [6][F]
    float* g0 = sm; float* g1 = sm + F; float* g2 = sm + 2 * F;
    float* x0 = sm + 3 * F; float* x1 = sm + 4 * F; float* x2 = sm + 5 * F;
    for (int c = threadIdx.x; c < F; c += 256) {
        // tot [4][F] = (sum dpsi*s, sum d, sum d*zhat_g, sum d*zhat_x) of step 2
        const float a = tot[F + c], b = tot[2 * F + c], a2 = a, b2 = tot[3 * F + c];
        g0[c] = gamma_g[c] * invstd_g[c]; g1[c] = a / (float)M; g2[c] = b / (float)M;
        x0[c] = gamma_x[c] * invstd_x[c]; x1[c] = a2 / (float)M; x2[c] = b2 / (float)M;
        if (blockIdx.x == 0) {
            if (dbeta_g) dbeta_g[c] += a;
            if (dgamma_g) dgamma_g[c] += b;
            if (dbeta_x) dbeta_x[c] += a2;
            if (dgamma_x) dgamma_x[c] += b2;
            if (dwpsi) dwpsi[c] += tot[c];
        }
    }
    __syncthreads();
    const int CG = F >> 3;
    const int64_t total = M * CG;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int64_t m = v / CG;
        const int c = (int)(v - m * CG) * 8;
        float d[8], a[8], b[8], og[8], ox[8];
        unpack8(*(const u32x4*)(ds + m * F + c), d);
        unpack8(*(const u32x4*)(zg + m * F + c), a);
        unpack8(*(const u32x4*)(zx + m * F + c), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zhg = (a[j] - mean_g[c + j]) * invstd_g[c + j];
            const float zhx = (b[j] - mean_x[c + j]) * invstd_x[c + j];
            og[j] = g0[c + j] * (d[j] - g1[c + j] - zhg * g2[c + j]);
            ox[j] = x0[c + j] * (d[j] - x1[c + j] - zhx * x2[c + j]);
        }
        *(u32x4*)(dzg + m * F + c) = pack8(og);
        *(u32x4*)(dzx + m * F + c) = pack8(ox);
    }
}

static inline int grid1(int64_t n, int cap = 2048) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace aau

using namespace aau;
#define CHK_F(fn, C) AAU_REQUIRE((C) > 0 && (C) % 8 == 0 && (C) <= 2048, fn ": channels=%d must be a multiple of 8 in [8, 2048]", (int)(C))

extern "C" int aau_gate_psi(const aau_bf16* zg, const aau_bf16* zx, const float* sg, const float* hg,
                            const float* sx, const float* hx, const float* wpsi, float* psi_pre, aau_stat* stats,
                            int64_t stats_bytes, int64_t M, int F, void* stream) {
    AAU_REQUIRE(zg && zx && sg && hg && sx && hx && wpsi && psi_pre && M > 0, "aau_gate_psi: bad args");
    AAU_CHECK_STAT("aau_gate_psi", stats, stats_bytes, 1);
    CHK_F("aau_gate_psi", F);
    ProfScope prof(2, 2.0 * M * F, (hipStream_t)stream);
    const CGMap3 mp(F);
    int64_t b = (M + (int64_t)mp.PL * 16 - 1) / ((int64_t)mp.PL * 16);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    int64_t ppb = ((M + b - 1) / b + mp.PL - 1) / mp.PL * mp.PL;
    b = (M + ppb - 1) / ppb;
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(gate_psi_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, zg, zx, sg, hg, sx, hx,
                       wpsi, psi_pre, (long long*)stats, M, F, ppb);
    return check_launch("aau_gate_psi");
}

extern "C" int aau_gate_apply(const aau_bf16* x, int x_pitch, const float* psi_pre, const float* scale1,
                              const float* shift1, float* alpha, aau_bf16* out, int out_pitch, int64_t M, int C,
                              void* stream) {
    AAU_REQUIRE(x && psi_pre && scale1 && shift1 && out && M > 0, "aau_gate_apply: bad args");
    CHK_F("aau_gate_apply", C);
    AAU_REQUIRE(x_pitch % 8 == 0 && out_pitch % 8 == 0, "aau_gate_apply: pitch");
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(gate_apply_kernel, dim3(grid1(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, x_pitch,
                       psi_pre, scale1, shift1, alpha, out, out_pitch, M, C);
    return check_launch("aau_gate_apply");
}

extern "C" int aau_gate_bwd1(const aau_bf16* dout, int dout_pitch, const aau_bf16* x, int x_pitch,
                             const float* alpha, const float* psi_pre, const float* mean1, const float* invstd1,
                             aau_bf16* dx, int dx_pitch, float* dq, float* red1, int64_t M, int C, float* ws,
                             void* stream) {
    AAU_REQUIRE(dout && x && alpha && psi_pre && mean1 && invstd1 && dx && dq && red1 && ws && M > 0,
                "aau_gate_bwd1: bad args");
    AAU_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)red1 & 15) == 0, "aau_gate_bwd1: red1 / ws must be 16-byte aligned");
    CHK_F("aau_gate_bwd1", C);
    AAU_REQUIRE(dout_pitch % 8 == 0 && x_pitch % 8 == 0 && dx_pitch % 8 == 0, "aau_gate_bwd1: pitch");
    ProfScope prof(2, 0, (hipStream_t)stream);
    const CGMap3 mp(C);
    int64_t b = (M + (int64_t)mp.PL * 16 - 1) / ((int64_t)mp.PL * 16);
    if (b > AAU_BN_RED_MAX_BLOCKS) b = AAU_BN_RED_MAX_BLOCKS;
    if (b < 1) b = 1;
    int64_t ppb = ((M + b - 1) / b + mp.PL - 1) / mp.PL * mp.PL;
    b = (M + ppb - 1) / ppb;
    if (next_traversal()) ppb = -ppb;
    hipLaunchKernelGGL(gate_bwd1_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, dout, dout_pitch, x,
                       x_pitch, alpha, psi_pre, mean1, invstd1, dx, dx_pitch, dq, ws, M, C, ppb);
    red_fold_launch(ws, 4, (int)b, red1, 4, nullptr, 0, nullptr, (hipStream_t)stream);
    return check_launch("aau_gate_bwd1");
}

extern "C" int aau_gate_bwd2(const float* dq, const float* psi_pre, const float* red1, const float* gamma1,
                             const float* mean1, const float* invstd1, const aau_bf16* zg, const aau_bf16* zx,
                             const float* sg, const float* hg, const float* sx, const float* hx,
                             const float* mean_g, const float* invstd_g, const float* mean_x,
                             const float* invstd_x, const float* wpsi, aau_bf16* ds, float* tot, float* dgamma1,
                             float* dbeta1, int64_t M, int F, float* ws, void* stream) {
    AAU_REQUIRE(dq && psi_pre && red1 && gamma1 && mean1 && invstd1 && zg && zx && sg && hg && sx && hx && mean_g &&
                    invstd_g && mean_x && invstd_x && wpsi && ds && tot && ws && M > 0,
                "aau_gate_bwd2: bad args");
    AAU_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)tot & 15) == 0, "aau_gate_bwd2: tot / ws must be 16-byte aligned");
    CHK_F("aau_gate_bwd2", F);
    const CGMap3 mp(F);
    // 16 pixels per thread (two in flight): every workgroup ends with 4 block reductions + replica atomics, so
    // fewer, longer workgroups win (8: +20-40 %, 4: +60-100 %)
    int64_t b = (M + (int64_t)mp.PL * 16 - 1) / ((int64_t)mp.PL * 16);
    if (const char* e = getenv("AAU_GB2_PPT")) b = (M + (int64_t)mp.PL * atoi(e) - 1) / ((int64_t)mp.PL * atoi(e));   // experiment
    if (b > AAU_BN_RED_MAX_BLOCKS) b = AAU_BN_RED_MAX_BLOCKS;
    if (b < 1) b = 1;
    int64_t ppb = (M + b - 1) / b;
    b = (M + ppb - 1) / ppb;
    if (next_traversal()) ppb = -ppb;
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(gate_bwd2_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, dq, psi_pre, red1,
                       gamma1, mean1, invstd1, zg, zx, sg, hg, sx, hx, mean_g, invstd_g, mean_x, invstd_x, wpsi, ds,
                       ws, dgamma1, dbeta1, M, F, ppb);
    red_fold_launch(ws, 4 * F, (int)b, tot, 4 * F, nullptr, 0, nullptr, (hipStream_t)stream);
    return check_launch("aau_gate_bwd2");
}

extern "C" int aau_gate_bwd3(const aau_bf16* ds, const aau_bf16* zg, const aau_bf16* zx, const float* gamma_g,
                             const float* mean_g, const float* invstd_g, const float* gamma_x,
                             const float* mean_x, const float* invstd_x, const float* tot, aau_bf16* dzg,
                             aau_bf16* dzx, float* dgamma_g, float* dbeta_g, float* dgamma_x, float* dbeta_x,
                             float* dwpsi, int64_t M, int F, void* stream) {
    AAU_REQUIRE(ds && zg && zx && gamma_g && mean_g && invstd_g && gamma_x && mean_x && invstd_x && tot &&
                    dzg && dzx && M > 0, "aau_gate_bwd3: bad args");
    CHK_F("aau_gate_bwd3", F);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(gate_bwd3_kernel, dim3(grid1(M * (F / 8))), dim3(256), 6 * F * sizeof(float),
                       (hipStream_t)stream, ds, zg, zx, gamma_g, mean_g, invstd_g, gamma_x, mean_x, invstd_x,
                       tot, dzg, dzx, dgamma_g, dbeta_g, dgamma_x, dbeta_x, dwpsi, M, F);
    return check_launch("aau_gate_bwd3");
}

// =====================================================================================================================
// Residual attention gate of the ablation variant (test_ablation.py:128-143): no BatchNorm, bias on psi,
//   a = sigmoid(psi_w . relu(Wg g + Wx x) + psi_b),   out = x * a + x
// One wave per pixel (lanes own 8-channel groups, wave reductions by shuffles): Fint is max(8, C/4), so a pixel is at
// most 48 lanes of x and 12 lanes of the gate sum.  HBM-bound; the two 1x1 GEMMs stay on the MFMA conv kernels.
// =====================================================================================================================
namespace aau {

__global__ __launch_bounds__(256) void gate2_fwd_kernel(const unsigned short* zg, const unsigned short* zx, const float* wpsi,
                                                        const float* bpsi, const unsigned short* x, int xp, float* alpha,
                                                        unsigned short* out, int op, int64_t M, int F, int C) {
    const int lane = threadIdx.x & 63;
    const int FG = F >> 3, CG = C >> 3;
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = lane < FG ? wpsi[lane * 8 + j] : 0.f;
    const float b = bpsi[0];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t m = wave0; m < M; m += nw) {
        float part = 0.f;
        if (lane < FG) {
            float g[8], h[8];
            unpack8(*(const u32x4*)(zg + m * F + lane * 8), g);
            unpack8(*(const u32x4*)(zx + m * F + lane * 8), h);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += fmaxf(g[j] + h[j], 0.f) * w[j];
        }
        const float a = 1.f / (1.f + expf(-(wave_sum(part) + b)));
        if (lane == 0) alpha[m] = a;
        for (int cg = lane; cg < CG; cg += 64) {
            float f[8];
            unpack8(*(const u32x4*)(x + m * xp + cg * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = f[j] * a + f[j];
            *(u32x4*)(out + m * op + cg * 8) = pack8(f);
        }
    }
}

// dx = dout * (1 + a) (written); dpre = <dout, x> * a (1 - a); ds[f] = dpre * w[f] * [s_f > 0] (written, bf16);
// every wave stores its row (sum dpre * s_f | sum dpre) [F + 8]; red_fold_launch adds the rows in a fixed order into
// dwpsi / dbpsi
__global__ __launch_bounds__(256) void gate2_bwd_kernel(const unsigned short* dout, int dop, const unsigned short* x, int xp,
                                                        const float* alpha, const unsigned short* zg, const unsigned short* zx,
                                                        const float* wpsi, unsigned short* dx, int dxp, unsigned short* ds,
                                                        float* rep, int64_t M, int F, int C) {
    const int lane = threadIdx.x & 63;
    const int FG = F >> 3, CG = C >> 3;
    float w[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j] = lane < FG ? wpsi[lane * 8 + j] : 0.f; acc[j] = 0.f; }
    float bsum = 0.f;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t m = wave0; m < M; m += nw) {
        const float a = alpha[m];
        float dot = 0.f;
        for (int cg = lane; cg < CG; cg += 64) {
            float g[8], xv[8];
            unpack8(*(const u32x4*)(dout + m * dop + cg * 8), g);
            unpack8(*(const u32x4*)(x + m * xp + cg * 8), xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dot += g[j] * xv[j]; g[j] = g[j] * a + g[j]; }
            *(u32x4*)(dx + m * dxp + cg * 8) = pack8(g);
        }
        const float dpre = wave_sum(dot) * a * (1.f - a);
        if (lane == 0) bsum += dpre;
        if (lane < FG) {
            float g[8], h[8], o[8];
            unpack8(*(const u32x4*)(zg + m * F + lane * 8), g);
            unpack8(*(const u32x4*)(zx + m * F + lane * 8), h);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float s = g[j] + h[j];
                o[j] = s > 0.f ? dpre * w[j] : 0.f;
                acc[j] += dpre * fmaxf(s, 0.f);
            }
            *(u32x4*)(ds + m * F + lane * 8) = pack8(o);
        }
    }
    float* r = red_row(rep, F + 8, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (lane < FG) {
        *(f32x4*)(r + lane * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *(f32x4*)(r + lane * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
    if (lane == 0) {
        *(f32x4*)(r + F) = f32x4{bsum, 0.f, 0.f, 0.f};
        *(f32x4*)(r + F + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace aau

extern "C" int aau_gate2_fwd(const aau_bf16* zg, const aau_bf16* zx, const float* wpsi, const float* bpsi, const aau_bf16* x,
                             int x_pitch, float* alpha, aau_bf16* out, int out_pitch, int64_t M, int F, int C, void* stream) {
    using namespace aau;
    AAU_REQUIRE(zg && zx && wpsi && bpsi && x && alpha && out && M > 0, "aau_gate2_fwd: bad args");
    AAU_REQUIRE(F > 0 && F % 8 == 0 && F <= 512 && C > 0 && C % 8 == 0, "aau_gate2_fwd: F=%d (<= 512) and C=%d must be multiples of 8", F, C);
    AAU_REQUIRE(x_pitch % 8 == 0 && out_pitch % 8 == 0, "aau_gate2_fwd: pitch");
    ProfScope prof(2, 0, (hipStream_t)stream);
    int64_t blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gate2_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, zg, zx, wpsi, bpsi, x, x_pitch,
                       alpha, out, out_pitch, M, F, C);
    return check_launch("aau_gate2_fwd");
}

extern "C" int aau_gate2_bwd(const aau_bf16* dout, int dout_pitch, const aau_bf16* x, int x_pitch, const float* alpha,
                             const aau_bf16* zg, const aau_bf16* zx, const float* wpsi, aau_bf16* dx, int dx_pitch, aau_bf16* ds,
                             float* rep_ws, float* dwpsi, float* dbpsi, int64_t M, int F, int C, void* stream) {
    using namespace aau;
    AAU_REQUIRE(dout && x && alpha && zg && zx && wpsi && dx && ds && rep_ws && dwpsi && dbpsi && M > 0, "aau_gate2_bwd: bad args");
    AAU_REQUIRE(F > 0 && F % 8 == 0 && F <= 512 && C > 0 && C % 8 == 0, "aau_gate2_bwd: F=%d (<= 512) and C=%d must be multiples of 8", F, C);
    AAU_REQUIRE(dout_pitch % 8 == 0 && x_pitch % 8 == 0 && dx_pitch % 8 == 0, "aau_gate2_bwd: pitch");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    AAU_REQUIRE(((uintptr_t)rep_ws & 15) == 0, "aau_gate2_bwd: ws must be 16-byte aligned");
    int64_t blocks = (M + 63) / 64;           // 16 pixels per wave
    if (blocks > AAU_BN_RED_MAX_BLOCKS / 4) blocks = AAU_BN_RED_MAX_BLOCKS / 4;     // one row per wave
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gate2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dout, dout_pitch, x, x_pitch, alpha, zg, zx,
                       wpsi, dx, dx_pitch, ds, rep_ws, M, F, C);
    red_fold_launch(rep_ws, F + 8, (int)blocks * 4, nullptr, 0, dwpsi, F, dbpsi, s);
    return check_launch("aau_gate2_bwd");
}

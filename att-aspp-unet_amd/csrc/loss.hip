// Criterion and metrics on the fp32 logits (pipeline:219-232 build_criterion with
// ComboLoss :187-189 = DiceLoss :173-178 + BCE, EdgeLoss :196-216; metrics :191-194, :240).
//
// The reference selects the positive samples with nonzero()/index (a host sync and a
// data-dependent shape).  Here the same arithmetic is stated with a per-sample mask
// m_b = [sum(t_b) > 0], P = sum m_b:
//   total = BCE_w(all)/(B*HW) + [P>0] * ( sum_b m_b*dice_b / P + sum_b m_b*BCE_b/(P*HW)
//                                         + edge_w * sum_b m_b*E_b/(P*HW) )
// so the whole criterion is two launches (per-sample sums; loss + d/dlogits) with no host
// round trip and is hipGraph-capturable.  The Sobel magnitude needs a 1-pixel halo for
// the value and a 2-pixel halo for its adjoint; tiles are staged in LDS.
#include "common.h"

namespace aau {

// sums layout per sample: 0 sum t, 1 sum p, 2 sum p*t, 3 sum bce, 4 sum |gp-gt|, 5 sum pbin, 6 sum pbin*t
constexpr int NS = 8;
constexpr int TILE = 32;                 // pixels per workgroup side: 256 threads x PPT pixels (a 16 x 16 tile per workgroup was
constexpr int PPT = TILE * TILE / 256;   // 8192 latency-bound workgroups with 15 barriers each: 33 + 26 us per step)
constexpr int NREP = AAU_STAT_REPLICAS;  // sums workspace = fp32 [NREP][B][NS] (the ABI's size)
// Inside that workspace: the first [B][NS] floats are the per-sample table the consumers read; behind it live
// NACC replicas of int64 [B][NS] fixed-point accumulators (value * 2^32; tile sums stay below 2^30) and a poison word.
// Integer adds are associative: the per-sample sums -- loss, Dice, the gradient of the criterion -- do not depend on the
// order in which the 8192 tiles arrive (the fp32-atomic form did).  A non-finite tile sum raises the poison word and the
// table becomes NaN, which keeps the reference's skip-the-step-on-inf/nan behaviour (pipeline:322-324).
constexpr int NACC = 14;
static_assert((NACC * 2 + 1) * NS + 2 <= NREP * NS, "accumulators must fit behind the table");
__device__ __forceinline__ unsigned long long* crit_acc(float* sums, int B) { return (unsigned long long*)(sums + (size_t)B * NS); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// numerically stable BCE-with-logits element: max(x,0) - x*t + log1p(exp(-|x|))
__device__ __forceinline__ float bce_elem(float x, float t) {
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

// seven sums at once: wave shuffles, one LDS exchange, thread k < 7 returns total k (the others 0)
__device__ __forceinline__ float block_sum7(float (&v)[7], float (*s74)[4]) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) s74[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    return threadIdx.x < 7 ? s74[threadIdx.x][0] + s74[threadIdx.x][1] + s74[threadIdx.x][2] + s74[threadIdx.x][3] : 0.f;
}

// grid: (tiles_x, tiles_y, B); 256 threads, PPT pixels each
__global__ __launch_bounds__(256) void crit_reduce_kernel(const float* logits, const float* targets, float* sums,
                                                          int H, int W, float thr_logit, int with_edge) {
    __shared__ float sp[TILE + 2][TILE + 2], st[TILE + 2][TILE + 2];
    __shared__ float s74[7][4];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const float* L = logits + (int64_t)b * H * W;
    const float* Tt = targets + (int64_t)b * H * W;
    const int tid = threadIdx.x;
    for (int i = tid; i < (TILE + 2) * (TILE + 2); i += 256) {
        const int ly = i / (TILE + 2), lx = i % (TILE + 2);
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        sp[ly][lx] = in ? sigmoidf_(L[(int64_t)y * W + x]) : 0.f;
        st[ly][lx] = in ? Tt[(int64_t)y * W + x] : 0.f;
    }
    __syncthreads();
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int ly = (tid + k * 256) / TILE, lx = (tid + k * 256) % TILE;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        const float l = L[(int64_t)y * W + x];
        const float p = sp[ly + 1][lx + 1], t = st[ly + 1][lx + 1];
        v[0] += t; v[1] += p; v[2] += p * t; v[3] += bce_elem(l, t);
        const float pb = l > thr_logit ? 1.f : 0.f;
        v[5] += pb; v[6] += pb * t;
        if (with_edge) {
#define SOBEL(A, gx, gy)                                                                              \
    const float gx = (A[ly][lx] - A[ly][lx + 2]) + 2.f * (A[ly + 1][lx] - A[ly + 1][lx + 2]) +        \
                     (A[ly + 2][lx] - A[ly + 2][lx + 2]);                                             \
    const float gy = (A[ly][lx] + 2.f * A[ly][lx + 1] + A[ly][lx + 2]) -                              \
                     (A[ly + 2][lx] + 2.f * A[ly + 2][lx + 1] + A[ly + 2][lx + 2]);
            SOBEL(sp, gxp, gyp)
            SOBEL(st, gxt, gyt)
            v[4] += fabsf(sqrtf(gxp * gxp + gyp * gyp + 1e-8f) - sqrtf(gxt * gxt + gyt * gyt + 1e-8f));
        }
    }
    const float s = block_sum7(v, s74);
    if (tid < 7 && s != 0.f) {
        unsigned long long* acc = crit_acc(sums, (int)gridDim.z);
        if (!(fabsf(s) < 1.0e9f)) {
            atomicOr(acc + (size_t)NACC * gridDim.z * NS, 1ull);                  // poison
        } else {
            const long long q = (long long)rint((double)s * 4294967296.0);
            atomicAdd(acc + ((size_t)((blockIdx.x + blockIdx.y) % NACC) * gridDim.z + b) * NS + tid, (unsigned long long)q);
        }
    }
}

// fold the fixed-point replicas into the fp32 table [B][NS] at the head of the workspace
__global__ void crit_fold_kernel(float* sums, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NS) return;
    const unsigned long long* acc = crit_acc(sums, B);
    long long a = 0;
    for (int r = 0; r < NACC; ++r) a += (long long)acc[(size_t)r * B * NS + i];
    const bool poisoned = acc[(size_t)NACC * B * NS] != 0;
    sums[i] = poisoned ? __uint_as_float(0x7fc00000u) : (float)((double)a * (1.0 / 4294967296.0));
}

struct CritTerms {  // per-launch scalars derived from the per-sample sums
    float inv_all;   // 1/(B*HW)
    float inv_pos;   // 1/(P*HW) or 0
    float inv_P;     // 1/P or 0
};

__device__ __forceinline__ CritTerms crit_terms(const float* sums, int B, float HW) {
    int P = 0;
    for (int b = 0; b < B; ++b) P += sums[b * NS] > 0.f ? 1 : 0;
    CritTerms c;
    c.inv_all = 1.f / ((float)B * HW);
    c.inv_pos = P > 0 ? 1.f / ((float)P * HW) : 0.f;
    c.inv_P = P > 0 ? 1.f / (float)P : 0.f;
    return c;
}

__global__ void crit_loss_kernel(const float* sums, float* loss_out, int B, float HW, int finetune, float neg_w,
                                 float edge_w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const CritTerms c = crit_terms(sums, B, HW);
    float bce_all = 0.f, dice = 0.f, bce_pos = 0.f, edge = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* s = sums + b * NS;
        const bool pos = s[0] > 0.f;
        const float w = (finetune && !pos) ? neg_w : 1.f;
        bce_all += w * s[3];
        if (pos) {
            dice += 1.f - (2.f * s[2] + 1.f) / (s[1] + s[0] + 1.f);
            bce_pos += s[3];
            edge += s[4];
        }
    }
    bce_all *= c.inv_all;
    dice *= c.inv_P;
    bce_pos *= c.inv_pos;
    edge *= c.inv_pos * edge_w;
    loss_out[0] = dice + bce_pos + bce_all + edge;
    loss_out[1] = dice;
    loss_out[2] = bce_pos + bce_all;
    loss_out[3] = edge;
}

// d(total)/d(logits); grid (tiles_x, tiles_y, B); block 16x16
__global__ __launch_bounds__(256) void crit_grad_kernel(const float* logits, const float* targets, const float* sums,
                                                        float* dlogits, int B, int H, int W, int finetune, float neg_w,
                                                        float edge_w, float loss_scale) {
    __shared__ float sp[TILE + 4][TILE + 4], st[TILE + 4][TILE + 4];
    __shared__ float su[TILE + 2][TILE + 2], sv[TILE + 2][TILE + 2];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const float* L = logits + (int64_t)b * H * W;
    const float* Tt = targets + (int64_t)b * H * W;
    const int tid = threadIdx.x;
    const float HW = (float)H * (float)W;
    const CritTerms ct = crit_terms(sums, B, HW);
    const float* s = sums + b * NS;
    const bool pos = s[0] > 0.f;
    const bool do_edge = pos && edge_w > 0.f;
    if (do_edge) {
        for (int i = tid; i < (TILE + 4) * (TILE + 4); i += 256) {
            const int ly = i / (TILE + 4), lx = i % (TILE + 4);
            const int y = y0 + ly - 2, x = x0 + lx - 2;
            const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            sp[ly][lx] = in ? sigmoidf_(L[(int64_t)y * W + x]) : 0.f;
            st[ly][lx] = in ? Tt[(int64_t)y * W + x] : 0.f;
        }
        __syncthreads();
        // u = sign(gp-gt)*gx/|grad p|, v = sign(gp-gt)*gy/|grad p| on the tile + 1 halo (zero outside the image)
        for (int i = tid; i < (TILE + 2) * (TILE + 2); i += 256) {
            const int ly = i / (TILE + 2), lx = i % (TILE + 2);
            const int y = y0 + ly - 1, x = x0 + lx - 1;
            float u = 0.f, v = 0.f;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                SOBEL(sp, gxp, gyp)
                SOBEL(st, gxt, gyt)
                const float mp = sqrtf(gxp * gxp + gyp * gyp + 1e-8f);
                const float mt = sqrtf(gxt * gxt + gyt * gyt + 1e-8f);
                const float d = mp - mt;
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                u = sg * gxp / mp;
                v = sg * gyp / mp;
            }
            su[ly][lx] = u;
            sv[ly][lx] = v;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
    const int ly = (tid + k * 256) / TILE, lx = (tid + k * 256) % TILE;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    const float l = L[(int64_t)y * W + x], t = Tt[(int64_t)y * W + x];
    const float p = sigmoidf_(l);
    const float w = (finetune && !pos) ? neg_w : 1.f;
    float g = w * ct.inv_all * (p - t);
    if (pos) {
        g += ct.inv_pos * (p - t);
        const float D = s[1] + s[0] + 1.f, Nn = 2.f * s[2] + 1.f;
        g += -ct.inv_P * (2.f * t * D - Nn) / (D * D) * p * (1.f - p);
        if (do_edge) {
            // adjoint of the two cross-correlations: dE/dp_j = sum_{a,b} kx[a][b]*u[j-(a-1,b-1)] + ky[a][b]*v[...]
            // (su/sv index of pixel j is [ly+1][lx+1]; j-(a-1,b-1) -> [ly+2-a][lx+2-b])
            const float ax = (su[ly + 2][lx + 2] - su[ly + 2][lx]) + 2.f * (su[ly + 1][lx + 2] - su[ly + 1][lx]) +
                             (su[ly][lx + 2] - su[ly][lx]);
            const float ay = (sv[ly + 2][lx + 2] + 2.f * sv[ly + 2][lx + 1] + sv[ly + 2][lx]) -
                             (sv[ly][lx + 2] + 2.f * sv[ly][lx + 1] + sv[ly][lx]);
            g += edge_w * ct.inv_pos * (ax + ay) * p * (1.f - p);
        }
    }
    dlogits[(int64_t)b * H * W + (int64_t)y * W + x] = g * loss_scale;
    }
}

// ---- the loss classes used on their own (pipeline:173-189 DiceLoss / TverskyLoss / ComboLoss, :196-216 EdgeLoss) ----
// Every sample counts (no positive-subset selection: that belongs to build_criterion):
//   total = w_ratio * mean_b (1 - N_b / D_b) + w_bce * mean(bce) + w_edge * mean |grad p - grad t|
//   N_b = nu * tp_b + s_n,   D_b = d_tp * tp_b + d_p * sum p_b + d_t * sum t_b + s_d      (tp = sum p*t)
// Dice(smooth s): nu 2, s_n s, d_tp 0, d_p 1, d_t 1, s_d s.  Tversky(a, b, s): nu 1, s_n s, d_tp 1-a-b, d_p a, d_t b, s_d s.
struct TermCoef { float w_ratio, nu, s_n, d_tp, d_p, d_t, s_d, w_bce, w_edge; };

__global__ void terms_loss_kernel(const float* sums, float* loss_out, int B, float HW, TermCoef k) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ratio = 0.f, bce = 0.f, edge = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* s = sums + b * NS;
        ratio += 1.f - (k.nu * s[2] + k.s_n) / (k.d_tp * s[2] + k.d_p * s[1] + k.d_t * s[0] + k.s_d);
        bce += s[3];
        edge += s[4];
    }
    const float inv_all = 1.f / ((float)B * HW);
    ratio = k.w_ratio * ratio / (float)B;
    bce *= k.w_bce * inv_all;
    edge *= k.w_edge * inv_all;
    loss_out[0] = ratio + bce + edge;
    loss_out[1] = ratio;
    loss_out[2] = bce;
    loss_out[3] = edge;
}

__global__ __launch_bounds__(256) void terms_grad_kernel(const float* logits, const float* targets, const float* sums,
                                                         float* dlogits, int B, int H, int W, TermCoef k) {
    __shared__ float sp[TILE + 4][TILE + 4], st[TILE + 4][TILE + 4];
    __shared__ float su[TILE + 2][TILE + 2], sv[TILE + 2][TILE + 2];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const float* L = logits + (int64_t)b * H * W;
    const float* Tt = targets + (int64_t)b * H * W;
    const int tid = threadIdx.x;
    const float inv_all = 1.f / ((float)B * (float)H * (float)W);
    const float* s = sums + b * NS;
    const bool do_edge = k.w_edge != 0.f;
    if (do_edge) {
        for (int i = tid; i < (TILE + 4) * (TILE + 4); i += 256) {
            const int ly = i / (TILE + 4), lx = i % (TILE + 4);
            const int y = y0 + ly - 2, x = x0 + lx - 2;
            const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            sp[ly][lx] = in ? sigmoidf_(L[(int64_t)y * W + x]) : 0.f;
            st[ly][lx] = in ? Tt[(int64_t)y * W + x] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < (TILE + 2) * (TILE + 2); i += 256) {
            const int ly = i / (TILE + 2), lx = i % (TILE + 2);
            const int y = y0 + ly - 1, x = x0 + lx - 1;
            float u = 0.f, v = 0.f;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                SOBEL(sp, gxp, gyp)
                SOBEL(st, gxt, gyt)
                const float mp = sqrtf(gxp * gxp + gyp * gyp + 1e-8f);
                const float mt = sqrtf(gxt * gxt + gyt * gyt + 1e-8f);
                const float d = mp - mt;
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                u = sg * gxp / mp;
                v = sg * gyp / mp;
            }
            su[ly][lx] = u;
            sv[ly][lx] = v;
        }
        __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < PPT; ++kk) {
    const int ly = (tid + kk * 256) / TILE, lx = (tid + kk * 256) % TILE;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    const float l = L[(int64_t)y * W + x], t = Tt[(int64_t)y * W + x];
    const float p = sigmoidf_(l);
    float g = k.w_bce * inv_all * (p - t);
    if (k.w_ratio != 0.f) {
        const float Nn = k.nu * s[2] + k.s_n, D = k.d_tp * s[2] + k.d_p * s[1] + k.d_t * s[0] + k.s_d;
        g += -(k.w_ratio / (float)B) * (k.nu * t * D - Nn * (k.d_tp * t + k.d_p)) / (D * D) * p * (1.f - p);
    }
    if (do_edge) {
        const float ax = (su[ly + 2][lx + 2] - su[ly + 2][lx]) + 2.f * (su[ly + 1][lx + 2] - su[ly + 1][lx]) +
                         (su[ly][lx + 2] - su[ly][lx]);
        const float ay = (sv[ly + 2][lx + 2] + 2.f * sv[ly + 2][lx + 1] + sv[ly + 2][lx]) -
                         (sv[ly][lx + 2] + 2.f * sv[ly][lx + 1] + sv[ly][lx]);
        g += k.w_edge * inv_all * (ax + ay) * p * (1.f - p);
    }
    dlogits[(int64_t)b * H * W + (int64_t)y * W + x] = g;
    }
}

__global__ void seg_metrics_kernel(const float* sums, float* out, int B) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float d = 0.f, i = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* s = sums + b * NS;
        d += (2.f * s[2] + 1.f) / (s[1] + s[0] + 1.f);
        i += s[6] / (s[5] + s[0] - s[6] + 1e-7f);
    }
    out[0] = d / (float)B;
    out[1] = i / (float)B;
}

}  // namespace aau

using namespace aau;

extern "C" int aau_criterion(const float* logits, const float* targets, float* sums, float* loss_out,
                             float* dlogits, int B, int H, int W, int finetune, float neg_bce_w, float edge_w,
                             float loss_scale, void* stream) {
    AAU_REQUIRE(logits && targets && sums && loss_out && B > 0 && B <= 4096 && H > 0 && W > 0,
                "aau_criterion: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    zero_f32(sums, (int64_t)NREP * B * NS, s);
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    hipLaunchKernelGGL(crit_reduce_kernel, grid, dim3(256), 0, s, logits, targets, sums, H, W, 0.f,
                       edge_w > 0.f ? 1 : 0);
    hipLaunchKernelGGL(crit_fold_kernel, dim3((B * NS + 255) / 256), dim3(256), 0, s, sums, B);
    hipLaunchKernelGGL(crit_loss_kernel, dim3(1), dim3(64), 0, s, sums, loss_out, B, (float)H * (float)W, finetune,
                       neg_bce_w, edge_w);
    if (dlogits)
        hipLaunchKernelGGL(crit_grad_kernel, grid, dim3(256), 0, s, logits, targets, sums, dlogits, B, H, W, finetune,
                           neg_bce_w, edge_w, loss_scale);
    return check_launch("aau_criterion");
}

extern "C" int aau_loss_terms(const float* logits, const float* targets, float* sums, float* loss_out, float* dlogits,
                              int B, int H, int W, const float* coef9, void* stream) {
    AAU_REQUIRE(logits && targets && sums && loss_out && coef9 && B > 0 && B <= 4096 && H > 0 && W > 0,
                "aau_loss_terms: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    const TermCoef k{coef9[0], coef9[1], coef9[2], coef9[3], coef9[4], coef9[5], coef9[6], coef9[7], coef9[8]};
    zero_f32(sums, (int64_t)NREP * B * NS, s);
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    hipLaunchKernelGGL(crit_reduce_kernel, grid, dim3(256), 0, s, logits, targets, sums, H, W, 0.f, k.w_edge != 0.f ? 1 : 0);
    hipLaunchKernelGGL(crit_fold_kernel, dim3((B * NS + 255) / 256), dim3(256), 0, s, sums, B);
    hipLaunchKernelGGL(terms_loss_kernel, dim3(1), dim3(64), 0, s, sums, loss_out, B, (float)H * (float)W, k);
    if (dlogits)
        hipLaunchKernelGGL(terms_grad_kernel, grid, dim3(256), 0, s, logits, targets, sums, dlogits, B, H, W, k);
    return check_launch("aau_loss_terms");
}

extern "C" int aau_seg_metrics(const float* logits, const float* targets, float* sums, float* metrics_out, int B,
                               int H, int W, float thr, void* stream) {
    AAU_REQUIRE(logits && targets && sums && metrics_out && B > 0 && B <= 4096 && H > 0 && W > 0,
                "aau_seg_metrics: bad args");
    AAU_REQUIRE(thr > 0.f && thr < 1.f, "aau_seg_metrics: thr=%f must be in (0,1)", thr);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    zero_f32(sums, (int64_t)NREP * B * NS, s);
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    const float thr_logit = logf(thr / (1.f - thr));  // sigmoid(l) > thr  <=>  l > logit(thr)
    hipLaunchKernelGGL(crit_reduce_kernel, grid, dim3(256), 0, s, logits, targets, sums, H, W, thr_logit, 0);
    hipLaunchKernelGGL(crit_fold_kernel, dim3((B * NS + 255) / 256), dim3(256), 0, s, sums, B);
    hipLaunchKernelGGL(seg_metrics_kernel, dim3(1), dim3(64), 0, s, sums, metrics_out, B);
    return check_launch("aau_seg_metrics");
}

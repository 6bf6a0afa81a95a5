// The ASPP image-pool branch between its two spatial ends (attention_aspp_unet_pipeline_stage.py:75-77):
//     pooled [B][Cin]  ->  Conv2d(Cin, Cout, 1, bias=False)  ->  BatchNorm2d over the B samples (training)  ->  ReLU
// is a B x Cin x Cout matrix product (8 x 384 x 768 at the benchmark size) with a BatchNorm whose statistics are per output
// channel over B values.  As generic launches that is an implicit GEMM with M = 8 rows, a statistics fold, two BatchNorm
// passes, a weight-gradient GEMM and an input-gradient GEMM: seven launches of 7-22 us each, all latency.  Here:
//   aau_poolbranch_fwd   z = x W^T, batch statistics, running statistics, folded scale / shift      (one launch)
//   aau_poolbranch_bwd   BatchNorm + ReLU backward over the batch, dgamma / dbeta, dW                 (one launch)
//   aau_poolbranch_dx    dx = dz W                                                                    (one launch)
// A workgroup of 256 threads owns 4 output channels (forward / backward) or 4 input channels (dx), one wave each; the 64 lanes
// split the reduction dimension (8 consecutive channels per lane: one trip at 384 channels, two at 768), shuffles add the
// slices in a fixed order: bitwise reproducible, no atomics.  (16 lanes per channel, the first form, made three dependent
// trips of 16-byte loads per lane: 17-22 us per launch.)
// Numerics as the generic path: 16-bit operands (packed weights, pooled activations, dz), fp32 accumulation, statistics
// from the fp32 accumulator, z and dz stored in the 16-bit type.
#include "common.h"

namespace aau {

constexpr int PB_MAXB = 16;      // samples (training-mode BatchNorm of this branch needs >= 2)

__device__ __forceinline__ float lanes16_sum(float v) {      // (the name is historical: all 64 lanes of the wave)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wpk: [Cout][Cpad] (the forward GEMM operand), x: [B][x_pitch]
__global__ __launch_bounds__(256) void poolbranch_fwd_kernel(const unsigned short* x, int xp, const unsigned short* wpk, int Cpad,
                                                            unsigned short* z, const float* gamma, const float* beta,
                                                            float* rmean, float* rvar, long long* nbt, float* scale, float* shift,
                                                            float* smean, float* sinvstd, int B, int Cin, int Cout, float eps,
                                                            float momentum) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    float acc[PB_MAXB];
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) acc[b] = 0.f;
    if (c < Cout) {
        for (int k = l * 8; k < Cin; k += 512) {                 // 8 consecutive channels per lane and trip
            float w8[8];
            unpack8(*(const u32x4*)(wpk + (size_t)c * Cpad + k), w8);
#pragma unroll
            for (int b = 0; b < PB_MAXB; ++b) {
                if (b < B) {
                    float x8[8];
                    unpack8(*(const u32x4*)(x + (size_t)b * xp + k), x8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[b] = __builtin_fmaf(w8[j], x8[j], acc[b]);
                }
            }
        }
    }
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) {
        acc[b] = lanes16_sum(acc[b]);
        if (b < B) { s1 += acc[b]; s2 += (double)acc[b] * acc[b]; }
    }
    if (c < Cout && l == 0) {
        // the arithmetic of bn_finalize_kernel (bn.hip) on the fp32 accumulators
        const double mean_d = s1 / B;
        const float mean = (float)mean_d;
        const float var = fmaxf((float)(s2 / B - mean_d * mean_d), 0.f);
        const float istd = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * istd;
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
        smean[c] = mean;
        sinvstd[c] = istd;
        if (rmean) {
            const float unb = B > 1 ? var * (float)B / (float)(B - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
        for (int b = 0; b < B; ++b) z[(size_t)b * Cout + c] = f2bf(acc[b]);
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

// dy: [B][dyp] gradient of the ReLU output; -> dz [B][Cout] (16-bit), dgamma / dbeta +=, dw [Cout][Cin] += (fp32)
__global__ __launch_bounds__(256) void poolbranch_bwd_kernel(const unsigned short* dy, int dyp, const unsigned short* z,
                                                            const unsigned short* x, int xp, const float* gamma,
                                                            const float* scale, const float* shift, const float* smean,
                                                            const float* sinvstd, unsigned short* dz, float* dgamma, float* dbeta,
                                                            float* dw, int B, int Cin, int Cout) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (c >= Cout) return;
    float g[PB_MAXB], zh[PB_MAXB], dzv[PB_MAXB];
    float s1 = 0.f, s2 = 0.f;
    const float sc = scale[c], sh = shift[c], mu = smean[c], is = sinvstd[c];
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) {
        g[b] = zh[b] = 0.f;
        if (b < B) {
            const float zz = bf2f(z[(size_t)b * Cout + c]);
            const float gv = (zz * sc + sh > 0.f) ? bf2f(dy[(size_t)b * dyp + c]) : 0.f;
            g[b] = gv;
            zh[b] = (zz - mu) * is;
            s1 += gv;
            s2 += gv * zh[b];
        }
    }
    const float k0 = gamma[c] * is, k1 = s1 / B, k2 = s2 / B;
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) dzv[b] = b < B ? bf2f(f2bf(k0 * (g[b] - k1 - zh[b] * k2))) : 0.f;
    if (l == 0) {
        dbeta[c] += s1;
        dgamma[c] += s2;
        for (int b = 0; b < B; ++b) dz[(size_t)b * Cout + c] = f2bf(dzv[b]);
    }
    for (int k = l * 8; k < Cin; k += 512) {
        float a8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = 0.f;
#pragma unroll
        for (int b = 0; b < PB_MAXB; ++b) {
            if (b < B) {
                float x8[8];
                unpack8(*(const u32x4*)(x + (size_t)b * xp + k), x8);
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = __builtin_fmaf(dzv[b], x8[j], a8[j]);
            }
        }
        float* o = dw + (size_t)c * Cin + k;
        *(f32x4*)o = f32x4{o[0] + a8[0], o[1] + a8[1], o[2] + a8[2], o[3] + a8[3]};
        *(f32x4*)(o + 4) = f32x4{o[4] + a8[4], o[5] + a8[5], o[6] + a8[6], o[7] + a8[7]};
    }
}

// wpd: [Cin][Cpad_d] (the data-gradient operand: row k holds W[:, k]); dx [B][dxp] = dz W
__global__ __launch_bounds__(256) void poolbranch_dx_kernel(const unsigned short* dz, const unsigned short* wpd, int Cpadd,
                                                           unsigned short* dx, int dxp, int B, int Cin, int Cout) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    float acc[PB_MAXB];
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) acc[b] = 0.f;
    if (k < Cin) {
        for (int c = l * 8; c < Cout; c += 512) {
            float w8[8];
            unpack8(*(const u32x4*)(wpd + (size_t)k * Cpadd + c), w8);
#pragma unroll
            for (int b = 0; b < PB_MAXB; ++b) {
                if (b < B) {
                    float d8[8];
                    unpack8(*(const u32x4*)(dz + (size_t)b * Cout + c), d8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[b] = __builtin_fmaf(w8[j], d8[j], acc[b]);
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < PB_MAXB; ++b) acc[b] = lanes16_sum(acc[b]);
    if (k < Cin && l == 0)
        for (int b = 0; b < B; ++b) dx[(size_t)b * dxp + k] = f2bf(acc[b]);
}

}  // namespace aau

using namespace aau;

// B >= 2: BatchNorm over the B pooled vectors in training mode -- with one sample per channel the variance is 0 and the
// reference's BatchNorm2d raises "Expected more than 1 value per channel when training" (pipeline:75-77)
#define PB_CHECK(fn, B, Cin, Cout)                                                                                         \
    AAU_REQUIRE((B) >= 2 && (B) <= PB_MAXB && (Cin) >= 8 && (Cin) % 8 == 0 && (Cout) >= 8 && (Cout) % 8 == 0,                 \
                fn ": B=%d (2..%d: training-mode BatchNorm needs more than 1 value per channel), Cin=%d, Cout=%d (multiples of 8)", \
                (int)(B), PB_MAXB, (int)(Cin), (int)(Cout))

extern "C" int aau_poolbranch_fwd(const aau_bf16* x, int x_pitch, const aau_bf16* wpk, int Cpad, aau_bf16* z, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                  float* scale, float* shift, float* save_mean, float* save_invstd, int B, int Cin, int Cout,
                                  float eps, float momentum, void* stream) {
    AAU_REQUIRE(x && wpk && z && gamma && beta && scale && shift && save_mean && save_invstd, "aau_poolbranch_fwd: null pointer");
    PB_CHECK("aau_poolbranch_fwd", B, Cin, Cout);
    AAU_REQUIRE(x_pitch % 8 == 0 && Cpad % 8 == 0 && Cpad >= Cin && (((uintptr_t)x | (uintptr_t)wpk) & 15) == 0,
                "aau_poolbranch_fwd: 16-byte rows");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 2.0 * B * Cin * Cout, s);
    hipLaunchKernelGGL(poolbranch_fwd_kernel, dim3((Cout + 3) / 4), dim3(256), 0, s, x, x_pitch, wpk, Cpad, z, gamma, beta, running_mean,
                       running_var, (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd, B, Cin, Cout, eps, momentum);
    return check_launch("aau_poolbranch_fwd");
}

extern "C" int aau_poolbranch_bwd(const aau_bf16* dy, int dy_pitch, const aau_bf16* z, const aau_bf16* x, int x_pitch,
                                  const float* gamma, const float* scale, const float* shift, const float* save_mean,
                                  const float* save_invstd, aau_bf16* dz, float* dgamma, float* dbeta, float* dw, int B, int Cin,
                                  int Cout, void* stream) {
    AAU_REQUIRE(dy && z && x && gamma && scale && shift && save_mean && save_invstd && dz && dgamma && dbeta && dw,
                "aau_poolbranch_bwd: null pointer");
    PB_CHECK("aau_poolbranch_bwd", B, Cin, Cout);
    AAU_REQUIRE(x_pitch % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dw & 15) == 0, "aau_poolbranch_bwd: 16-byte rows");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 2.0 * B * Cin * Cout, s);
    hipLaunchKernelGGL(poolbranch_bwd_kernel, dim3((Cout + 3) / 4), dim3(256), 0, s, dy, dy_pitch, z, x, x_pitch, gamma, scale, shift,
                       save_mean, save_invstd, dz, dgamma, dbeta, dw, B, Cin, Cout);
    return check_launch("aau_poolbranch_bwd");
}

extern "C" int aau_poolbranch_dx(const aau_bf16* dz, const aau_bf16* wpd, int Cpad_d, aau_bf16* dx, int dx_pitch, int B, int Cin,
                                 int Cout, void* stream) {
    AAU_REQUIRE(dz && wpd && dx, "aau_poolbranch_dx: null pointer");
    PB_CHECK("aau_poolbranch_dx", B, Cin, Cout);
    AAU_REQUIRE(Cpad_d % 8 == 0 && Cpad_d >= Cout && (((uintptr_t)dz | (uintptr_t)wpd) & 15) == 0, "aau_poolbranch_dx: 16-byte rows");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 2.0 * B * Cin * Cout, s);
    hipLaunchKernelGGL(poolbranch_dx_kernel, dim3((Cin + 3) / 4), dim3(256), 0, s, dz, wpd, Cpad_d, dx, dx_pitch, B, Cin, Cout);
    return check_launch("aau_poolbranch_dx");
}

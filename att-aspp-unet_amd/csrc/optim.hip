// Optimiser step and weight packing over FLAT fp32 buffers (all 103 parameter tensors
// live back to back in one allocation, so clip_grad_norm_ + AdamW are three launches):
//   pipeline:323  torch.nn.utils.clip_grad_norm_(params, 1.0)   -> aau_grad_sqnorm + coefficient
//   pipeline:302  torch.optim.AdamW(lr, betas (.9,.999), eps 1e-8, weight_decay 5e-4)
//   pipeline:322/324 GradScaler.unscale_ / skipped step on inf -> inv_scale + finite check
// and the per-step fp32 -> bf16 repack of the convolution weights into the GEMM operand
// layouts of igemm.hip (forward: rows = Cout; data-gradient: rows = Cin, taps flipped).
#include "common.h"

namespace aau {

// One partial per workgroup into out[1 + blockIdx.x]; sqnorm_fold_kernel adds the partials in index order into out[0]
// (no float atomics: the clip coefficient, and with it the whole update, is bitwise reproducible).
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, int64_t n, float inv_scale, float* out) {
    __shared__ float s4[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = v[k] * inv_scale; acc += t * t; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) { const float t = g[i] * inv_scale; acc += t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[1 + blockIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
}

__global__ __launch_bounds__(256) void sqnorm_fold_kernel(float* out, int nblk) {
    __shared__ float s[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += out[1 + i];
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += s[i];
        out[0] = t;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, float* m, float* v, const float* g, int64_t n,
                                                    const float* norm_ws, const int64_t* step_dev, float lr,
                                                    float b1, float b2, float eps, float wd, float max_norm,
                                                    float inv_scale) {
    const float sq = norm_ws[0];
    if (!(sq < INFINITY)) return;  // inf / nan gradients: skip the step (GradScaler semantics)
    const float total = sqrtf(sq);
    float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;
    coef = fminf(coef, 1.f) * inv_scale;
    const double t = (double)(step_dev[0] + 1);  // bias corrections in double, as the Python-side reference
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, t));
    const float step_size = lr / bc1, decay = 1.f - lr * wd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);   // lerp_, as torch's single-tensor path
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

// Same update with the hyper-parameters in DEVICE memory (a replayed hipGraph then follows the LR schedule) and one
// (lr, weight_decay) pair per parameter group: hyp = [G][2] fp32, group_of_block = group id of every 64-element block of
// the flat buffers (parameters are 64-element aligned), or null for one group.  test_ablation.py:576-586 trains the
// attention parameters at twice the backbone's rate.
__global__ __launch_bounds__(256) void adamw_groups_kernel(float* p, float* m, float* v, const float* g, int64_t n,
                                                           const float* norm_ws, const int64_t* step_dev,
                                                           const float* hyp, const unsigned char* group_of_block,
                                                           float b1, float b2, float eps, float max_norm,
                                                           float inv_scale) {
    const float sq = norm_ws[0];
    if (!(sq < INFINITY)) return;
    const float total = sqrtf(sq);
    float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;
    coef = fminf(coef, 1.f) * inv_scale;
    const double t = (double)(step_dev[0] + 1);
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, t));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int grp = group_of_block ? group_of_block[i >> 6] : 0;
        const float lr = hyp[2 * grp], wd = hyp[2 * grp + 1];
        const float step_size = lr / bc1, decay = 1.f - lr * wd;
        const float gi = g[i] * coef;
        const float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

__global__ void step_inc_kernel(const float* norm_ws, int64_t* step_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && norm_ws[0] < INFINITY) step_dev[0] += 1;
}

// packed[dst_off + (r*T + t)*Cpad + c] = bf16(flat[src_off + r1*s_r + r2*s_r2 + tt*s_t + c*s_c]) (0 for c >= C)
// One thread produces 8 consecutive packed channels (one 16-B store).  The thread -> (row, tap, channel
// group) map follows the SOURCE's contiguous axis so that the fp32 reads coalesce: channel-fastest when
// s_c == 1 (two 16-B loads per thread), row-fastest otherwise (the transposed data-gradient operands).
__global__ __launch_bounds__(256) void pack_kernel(const float* flat, unsigned short* packed,
                                                   const aau_pack_entry* table, int n_entries) {
    // entry of this workgroup = number of entries that begin at or before it, minus one: counted by the whole workgroup
    // (the serial walk over the table was a chain of up to n_entries dependent loads per workgroup: 64 us per step)
    const int64_t blk = blockIdx.x;
    int cnt = 0;
    for (int k = threadIdx.x; k < n_entries; k += 256) cnt += table[k].blk_begin <= blk ? 1 : 0;
    __shared__ int s_cnt[4];
    int wsum = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = wsum;
    __syncthreads();
    const aau_pack_entry ent = table[s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] - 1];
    const unsigned C8 = ent.Cpad >> 3;
    const unsigned total = (unsigned)ent.R * ent.T * C8;
    const unsigned i = (unsigned)(blk - ent.blk_begin) * 256u + threadIdx.x;
    if (i >= total) return;
    unsigned r, t, c8;
    if (ent.s_c == 1) {
        c8 = i % C8;
        const unsigned rt = i / C8;
        t = rt % ent.T;
        r = rt / ent.T;
    } else {
        r = i % ent.R;
        const unsigned ct = i / ent.R;
        t = ct % ent.T;
        c8 = ct / ent.T;
    }
    const unsigned tt = ent.t_flip ? ent.T - 1 - t : t;
    int64_t off = ent.src_off + (int64_t)tt * ent.s_t;
    if (ent.R2 > 0) off += (int64_t)(r / ent.R2) * ent.s_r + (int64_t)(r % ent.R2) * ent.s_r2;
    else off += (int64_t)r * ent.s_r;
    const int c0 = c8 * 8;
    float v[8];
    if (ent.s_c == 1 && c0 + 8 <= ent.C && ((off + c0) & 3) == 0) {
        const f32x4 a = *(const f32x4*)(flat + off + c0), b = *(const f32x4*)(flat + off + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < ent.C) ? flat[off + (int64_t)(c0 + j) * ent.s_c] : 0.f;
    }
    *(u32x4*)(packed + ent.dst_off + ((int64_t)r * ent.T + t) * ent.Cpad + c0) = pack8(v);
}

// Up to 8 buffers cleared in ONE launch (blockIdx.y = buffer), plus an optional 64-bit counter bumped by `inc`: the
// step's per-pass housekeeping (statistics / reduction arenas, the flat gradient, the dropout seed) was five ATen fills.
struct ZeroMulti { void* p[8]; long long n16[8]; };
__global__ __launch_bounds__(256) void zero_multi_kernel(const ZeroMulti z, unsigned long long* counter, unsigned long long inc) {
    if (counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *counter += inc;
    u32x4* q = (u32x4*)z.p[blockIdx.y];
    const long long n = z.n16[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) q[i] = u32x4{0u, 0u, 0u, 0u};
}

}  // namespace aau

using namespace aau;

extern "C" int aau_grad_sqnorm(const float* grad, int64_t n, float inv_scale, float* norm_ws, void* stream) {
    AAU_REQUIRE(grad && norm_ws && n > 0, "aau_grad_sqnorm: bad args");
    AAU_REQUIRE(((uintptr_t)grad & 15) == 0, "aau_grad_sqnorm: grad must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > AAU_SQNORM_WS - 4) blocks = AAU_SQNORM_WS - 4;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad, n, inv_scale, norm_ws);
    hipLaunchKernelGGL(sqnorm_fold_kernel, dim3(1), dim3(256), 0, s, norm_ws, (int)blocks);
    return check_launch("aau_grad_sqnorm");
}

extern "C" int aau_adamw_step(float* p, float* m, float* v, const float* g, int64_t n, const float* norm_ws,
                              int64_t* step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float max_norm, float inv_scale, void* stream) {
    AAU_REQUIRE(p && m && v && g && norm_ws && step_dev && n > 0, "aau_adamw_step: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, m, v, g, n, norm_ws, step_dev, lr,
                       beta1, beta2, eps, weight_decay, max_norm, inv_scale);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, s, norm_ws, step_dev);
    return check_launch("aau_adamw_step");
}

extern "C" int aau_adamw_step_dev(float* p, float* m, float* v, const float* g, int64_t n, const float* norm_ws,
                                  int64_t* step_dev, const float* hyp, const unsigned char* group_of_block, int n_groups,
                                  float beta1, float beta2, float eps, float max_norm, float inv_scale, void* stream) {
    AAU_REQUIRE(p && m && v && g && norm_ws && step_dev && hyp && n > 0, "aau_adamw_step_dev: bad args");
    AAU_REQUIRE(n_groups >= 1 && n_groups <= 256 && (n_groups == 1 || group_of_block),
                "aau_adamw_step_dev: %d groups need a block->group table", n_groups);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, m, v, g, n, norm_ws, step_dev, hyp,
                       n_groups > 1 ? group_of_block : nullptr, beta1, beta2, eps, max_norm, inv_scale);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, s, norm_ws, step_dev);
    return check_launch("aau_adamw_step_dev");
}

extern "C" int aau_pack_weights(const float* flat, aau_bf16* packed, const aau_pack_entry* table_dev,
                                int n_entries, int64_t total_blocks, void* stream) {
    AAU_REQUIRE(flat && packed && table_dev && n_entries > 0 && total_blocks > 0 && total_blocks < 0x7fffffff,
                "aau_pack_weights: bad args");
    ProfScope prof(3, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, flat, packed,
                       table_dev, n_entries);
    return check_launch("aau_pack_weights");
}

extern "C" int aau_zero_multi(void* const* bufs, const int64_t* bytes, int n, uint64_t* counter, uint64_t counter_inc,
                              void* stream) {
    AAU_REQUIRE(n >= 0 && n <= 8 && (n == 0 || (bufs && bytes)), "aau_zero_multi: %d buffers (at most 8)", n);
    AAU_REQUIRE(n > 0 || counter, "aau_zero_multi: nothing to do");
    ZeroMulti z;
    long long most = 1;
    for (int i = 0; i < 8; ++i) { z.p[i] = nullptr; z.n16[i] = 0; }
    for (int i = 0; i < n; ++i) {
        AAU_REQUIRE(bufs[i] && bytes[i] > 0 && bytes[i] % 16 == 0 && ((uintptr_t)bufs[i] & 15) == 0,
                    "aau_zero_multi: buffer %d must be 16-byte aligned with a size that is a multiple of 16 (%lld)", i,
                    (long long)bytes[i]);
        z.p[i] = bufs[i];
        z.n16[i] = bytes[i] / 16;
        if (z.n16[i] > most) most = z.n16[i];
    }
    long long blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(3, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks, (unsigned)(n > 0 ? n : 1)), dim3(256), 0, (hipStream_t)stream, z,
                       (unsigned long long*)counter, (unsigned long long)counter_inc);
    return check_launch("aau_zero_multi");
}

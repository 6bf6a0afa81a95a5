// Byte / integer image work around the network on gfx950 (HBM-bound; no MFMA here):
//   * aau_seg_counts: the integer counts behind eval_segmentation_batch.py:41-49 Dice / IoU.
// (The GPU-resident inference tail and input pipeline of SURVEY.md section 8 rows f1 / f2 / f4 live here too.)
#include "common.h"

namespace aau {

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void seg_counts_kernel(const TA* __restrict__ a, const TB* __restrict__ b, int64_t n,
                                                         unsigned long long* __restrict__ out) {
    unsigned na = 0, nb = 0, ni = 0;   // per-thread counts stay below 2^32 (grid-stride over < 2^40 elements)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool x = a[i] > (TA)0, y = b[i] > (TB)0;
        na += x; nb += y; ni += (x && y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        na += __shfl_xor(na, o, 64); nb += __shfl_xor(nb, o, 64); ni += __shfl_xor(ni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (na) atomicAdd(out + 0, (unsigned long long)na);
        if (nb) atomicAdd(out + 1, (unsigned long long)nb);
        if (ni) atomicAdd(out + 2, (unsigned long long)ni);
    }
}

__global__ void zero_u64_kernel(unsigned long long* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0ull;
}

}  // namespace aau

using namespace aau;

extern "C" int aau_seg_counts(const void* a, int a_is_f32, const void* b, int b_is_f32, int64_t n, uint64_t* out3,
                              void* stream) {
    AAU_REQUIRE(a && b && out3 && n > 0, "aau_seg_counts: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    unsigned long long* o = (unsigned long long*)out3;
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(64), 0, s, o, 3);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 g((unsigned)blocks), t(256);
    if (a_is_f32 && b_is_f32) hipLaunchKernelGGL((seg_counts_kernel<float, float>), g, t, 0, s, (const float*)a, (const float*)b, n, o);
    else if (a_is_f32) hipLaunchKernelGGL((seg_counts_kernel<float, unsigned char>), g, t, 0, s, (const float*)a, (const unsigned char*)b, n, o);
    else if (b_is_f32) hipLaunchKernelGGL((seg_counts_kernel<unsigned char, float>), g, t, 0, s, (const unsigned char*)a, (const float*)b, n, o);
    else hipLaunchKernelGGL((seg_counts_kernel<unsigned char, unsigned char>), g, t, 0, s, (const unsigned char*)a, (const unsigned char*)b, n, o);
    return check_launch("aau_seg_counts");
}

// Argument block and epilogue helpers shared by the 3x3 kernels (conv3x3.hip, conv3x3s.hip).  Internal.
#pragma once
#include "common.h"

namespace aau {

struct C3Args {
    aau_conv_desc d;
    const unsigned short* src;
    const unsigned short* wpk;
    unsigned short* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    float* stats;
    int rev;             // 1: walk the patches from the end (aau_traverse)
    int nchunk;          // Cpad / 32
    unsigned src_bytes, wpk_bytes;
    int tiles_x, tiles_y;
    // conv3x3s only (aau_conv_igemm_bnred): dst is the gradient dy of a BatchNorm -> ReLU layer whose raw conv output is
    // bn_z; the epilogue then accumulates that layer's BatchNorm-backward sums (sum g, sum g * zhat, g = dy * relu') into
    // `stats` instead of (sum v, sum v^2)
    const unsigned short* bn_z = nullptr;
    int bn_zp = 0;
    const float* bn_scale = nullptr;
    const float* bn_shift = nullptr;
    const float* bn_mean = nullptr;
    const float* bn_invstd = nullptr;
    // conv3x3s only (aau_conv_igemm_bnin): src is the RAW conv output z of the producing BatchNorm -> ReLU layer; the kernel
    // applies y = relu(z * in_scale + in_shift) in LDS behind the landing fill (the activation is never written to memory)
    const float* in_scale = nullptr;
    const float* in_shift = nullptr;
    int nowide;          // experiment (AAU_NO_WIDE_STORE): 8-byte epilogue stores
    int nopair;          // experiment (AAU_RESW_NOPAIR): no two-taps-per-K-block packing of a short last chunk
};

// 16-B k-group swizzle of the 64-B LDS rows ([row][32 channels]).  ds_read_b128 is serviced in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (guide: LDS), i.e. with lane = 16 * kgroup + row a group holds rows
// f, f+12 of k-group a and rows f+4, f+8 of k-group a^1 for f = 0..3, and the four rows of one residue mod 4 share a
// 16-bank window: their swizzled k-groups must differ.  kgroup ^ 2*bit2(row) does that for ANY first row (the 3x3
// taps shift the 16-row window by 0..2 + 18 per halo row); the round-1 form (a 4-entry table on bits 2-3) was
// conflict-free only for windows that start at a multiple of 8 rows: SQ_LDS_BANK_CONFLICT was 0.22-0.30 of the LDS
// cycles of every halo kernel.
__device__ __forceinline__ int swz32(int row, int lc) { return lc ^ ((row >> 1) & 2); }

// epilogue statistics of one accumulator quad (4 consecutive channels q.. of one pixel): (sum v, sum v^2)
__device__ __forceinline__ void epi_stats(const C3Args&, int64_t, int, const float v[4], float s1[4], float s2[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
}

// Wide epilogue of one pair of accumulator quads (the same 16-channel group of two pixels A, B): statistics, bias and the
// folded-BN affine in the MFMA layout (channels q .. q+3), then the cross-lane swap of common.h (swap_pair8), after which
// this lane owns channels qw .. qw+7 of ONE of the two pixels and finishes with a single 16-byte (read-modify-)write at
// `out`, its own destination for (that pixel, qw).  Every lane of the wave must call this (the swap is a wave operation).
__device__ __forceinline__ void epi_pair_wide(const C3Args& a, const aau_conv_desc& d, int q, int qw, const f32x4& accA,
                                              const f32x4& accB, bool want_stats, float s1[4], float s2[4],
                                              unsigned short* out, bool store) {
    float va[4], vb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { va[r] = accA[r]; vb[r] = accB[r]; }
    if (q < d.Cout) {
        if (want_stats) { epi_stats(a, 0, q, va, s1, s2); epi_stats(a, 0, q, vb, s1, s2); }
        if (a.bias) {
            const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] += b[r]; vb[r] += b[r]; }
        }
        if (a.scale) {
            const f32x4 sc = *(const f32x4*)(a.scale + q);
            const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] = va[r] * sc[r] + sh[r]; vb[r] = vb[r] * sc[r] + sh[r]; }
        }
    }
    float w[8];
    swap_pair8(va, vb, w);
    if (qw >= d.Cout || !store) return;
    if (d.accumulate) {
        float o[8];
        unpack8(*(const u32x4*)out, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] += o[r];
    }
    if (d.relu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = fmaxf(w[r], 0.f);
    }
    *(u32x4*)out = pack8(w);
}

}  // namespace aau

// Strip-streaming 3x3 (stride 1, pad 1) convolution with REGISTER-RESIDENT weights for the high-resolution,
// few-channel layers (48 / 96 channels in and out: d1.1, d2.*, u1.conv.*, u2.conv.1 of pipeline:113-121, forward and
// data-gradient).
//
// Those layers are streaming problems (arithmetic intensity 216-430 FLOP/B): what bounds them is how many bytes a CU
// keeps in flight, not the MFMA.  The resident-weight kernels of conv3x3.hip keep the packed weights in LDS (83-110 KB),
// which leaves room for ONE 24-KB halo tile of look-ahead per CU: 2.7 TB/s chip-wide (MFMA 0.35, wait
// 0.45, HBM a third of peak).  Here the weights live in REGISTERS instead: a wave owns one 16-channel output group and
// holds all of its 9-tap x Cin fragments (60 / 108 VGPRs) for the life of the kernel, so
//   * LDS holds nothing but pixels: a ring of R image rows of an 18-pixel-wide column STRIP; the workgroup marches down
//     the strip, 16 new rows per 16x16 patch (no vertical halo re-fetch), with 16-32 rows = 55 KB of LDS-DMA in flight;
//   * no weight fragment is ever read from LDS: 6 pixel-fragment reads per 12 MFMAs.
// Pixels are stored channel-contiguous ([row][18 px][Cin], pixel stride 96 B / 224 B = 6 / 14 sixteen-byte units: any
// stride = 2 mod 4 units is bank-conflict free for the ds_read_b128 lane groups of gfx950), so for one vertical tap the
// three horizontal taps of an output pixel are ONE contiguous K run (Cin = 48: 144 elements = 4.5 K-blocks, the half
// block multiplies zero weights) and every LDS address is a row base + an immediate.
// 12 waves (3 per SIMD, <= 168 VGPRs): wave = (16-channel group g, row-quad slot); one barrier per patch.
#include <stdlib.h>
#include "common.h"
#include "c3args.h"

namespace aau {

template <int N>
__device__ __forceinline__ void wait_vm_s() {
    static_assert(N >= 0 && N <= 10, "count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// A work unit is a vertical segment of one 16-pixel-wide strip of one image: `segh` patches (the last segment of a strip
// may be shorter).  The fill stream of a unit is blocks b = -1 .. K-1 of 16 image rows: block b holds rows
// ys + 16 b + 1 .. ys + 16 b + 16, patch b (rows ys + 16 b ...) needs the last two rows of block b-1 and all of block b.
template <int CIN, int G>
__global__ __launch_bounds__(768) void conv3x3s_kernel(const C3Args a, int nunits, int strips, int nseg, int segh) {
    constexpr int PXB = CIN == 48 ? 96 : 224;          // bytes per pixel in LDS
    constexpr int SU = PXB / 16;                       // ... in 16-byte units
    constexpr int NB = CIN == 48 ? 5 : 9;              // K-blocks per vertical tap
    constexpr int ROWB = CIN == 48 ? 2048 : 4096;      // bytes per ring row (18 px, padded to whole 1-KiB DMA pieces)
    constexpr int IPR = ROWB / 1024;                   // LDS-DMA instructions per row
    constexpr int R = CIN == 48 ? 64 : 36;             // ring rows
    constexpr int D = CIN == 48 ? 2 : 1;               // blocks in flight
    constexpr int NF = (16 * IPR + 11) / 12;           // fill instructions per wave and block
    constexpr int SLOTS = 12 / G;                      // row-quad slots
    constexpr int NQ = 4 / SLOTS;                      // row quads per wave and patch
    constexpr int NST = NQ * 2;                        // 16-byte stores per lane and patch
    constexpr int BQ = 16 * G;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(R >= 18 + 16 * D, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // [R][ROWB] ring | 1 KiB scratch

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int g = wave % G, slot = wave / G;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const int tiles_y = d.H / 16;

    // ---- weights of this wave's 16-channel group: registers, once ----
    bf16x8 wr[3][NB];
    {
        const int q = g * 16 + fr;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int tx, c0;
                if constexpr (CIN == 48) {
                    const int t = 4 * j + fk;           // 16-byte unit inside the 3-pixel K run
                    tx = t / SU; c0 = (t - tx * SU) * 8;
                } else {
                    tx = j / 3; c0 = (j - tx * 3) * 32 + fk * 8;
                }
                bf16x8 v = {};
                if (tx < 3 && q < d.Cout) v = *(const bf16x8*)(a.wpk + ((size_t)(q * 9 + ty * 3 + tx) * d.Cpad + c0));
                wr[ty][j] = v;
            }
    }

    // ---- fill roles: this lane always fetches the same (pixel, 16-byte part) of a row; the row is wave-uniform ----
    const int sub = wave % IPR;
    const int piece = sub * 64 + lane;
    const int fpx = piece / SU, fpart = piece - fpx * SU;
    const bool lane_ok = fpx < 18 && fpart * 8 < CIN;
    const int split_c = d.src_split_c > 0 ? d.src_split_c : 0x7fffffff;
    const int sadj = (fpart * 8 >= split_c) ? d.src_split_off - d.src_split_c : 0;
    const int dsplit_c = d.dst_split_c > 0 ? d.dst_split_c : 0x7fffffff;
    const int dsplit_adj = d.dst_split_off - d.dst_split_c;

    auto unit_decode = [&](int u, int& n, int& x0, int& ys, int& K) {
        if (a.rev) u = nunits - 1 - u;
        const int seg = u % nseg;
        const int t2 = u / nseg;
        const int strip = t2 % strips;
        n = t2 / strips; x0 = strip * 16; ys = seg * segh * 16;
        K = tiles_y - seg * segh; if (K > segh) K = segh;
    };

    // issue cursor
    int iu = blockIdx.x, ib = -1, in_ = 0, ix0 = 0, iys = 0, iK = 0;
    unsigned ivec = OOB;                       // per-lane byte offset of (pixel, part) inside a row, OOB outside the image
    auto issue_unit_setup = [&]() {
        if (iu < nunits) {
            unit_decode(iu, in_, ix0, iys, iK);
            const int x = ix0 - 1 + fpx;
            ivec = (lane_ok && (unsigned)x < (unsigned)d.W) ? (unsigned)((x * d.src_pitch + fpart * 8 + sadj) * 2) : OOB;
        }
    };
    issue_unit_setup();
    int irb = 0;                               // ring row of the next block to issue
    auto issue_block = [&]() {
        const bool live = iu < nunits;
        const int yb = iys + 16 * ib + 1;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int idx = i * 12 + wave;
            const int r = idx / IPR;                                   // row of the block (wave-uniform)
            const int y = yb + r;
            const bool row_ok = live && idx < 16 * IPR && (unsigned)y < (unsigned)d.H && (ib >= 0 || r >= 14);
            int rr = irb + r; if (rr >= R) rr -= R;
            unsigned char* dstp = idx < 16 * IPR ? dsm + rr * ROWB + sub * 1024 : dsm + R * ROWB;
            const unsigned soff = row_ok ? (unsigned)(((in_ * d.H + y) * d.W) * d.src_pitch * 2) : 0u;
            const unsigned v = row_ok ? ivec : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dstp), 16, (int)v, (int)soff, 0, 0);
        }
        irb += 16; if (irb >= R) irb -= R;
        if (live) {
            if (++ib == iK) { iu += gridDim.x; ib = -1; issue_unit_setup(); }
        }
    };

    const bool want_stats = a.stats != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int lane_off = fr * PXB + fk * 16;

#pragma unroll
    for (int i = 0; i < D; ++i) issue_block();

    int crb = 0;                               // ring row of block t
    bool prev_patch = false;
    for (int cu = blockIdx.x; cu < nunits; cu += gridDim.x) {
        int n, x0, ys, K;
        unit_decode(cu, n, x0, ys, K);
        for (int cb = -1; cb < K; ++cb) {
            // block t has landed: the only younger operations are the fills of the blocks behind it and the previous
            // patch's stores (vmcnt retires in issue order)
            if (prev_patch) wait_vm_s<NF * (D - 1) + NST>(); else wait_vm_s<NF * (D - 1)>();
            __builtin_amdgcn_s_barrier();
            issue_block();                      // block t + D, into rows nobody reads any more
            prev_patch = cb >= 0;
            if (cb >= 0) {
                const int y0 = ys + 16 * cb;
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const int q = slot + qi * SLOTS;
                    unsigned va[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        int rr = crb + R - 2 + 4 * q + r;
                        rr -= (rr >= 2 * R) ? 2 * R : (rr >= R ? R : 0);
                        va[r] = (unsigned)(rr * ROWB + lane_off);
                    }
                    f32x4 acc[4];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        constexpr int dummy = 0; (void)dummy;
                        const int off = CIN == 48 ? 64 * j : (j / 3) * PXB + (j % 3) * 64;
                        bf16x8 af[6];
#pragma unroll
                        for (int r = 0; r < 6; ++r) af[r] = *(const bf16x8*)(dsm + va[r] + off);
#pragma unroll
                        for (int r = 0; r < 6; ++r)
#pragma unroll
                            for (int ty = 0; ty < 3; ++ty) {
                                const int mi = r - ty;
                                if (mi >= 0 && mi < 4) acc[mi] = AAU_MFMA16(wr[ty][j], af[r], acc[mi], 0, 0, 0);
                            }
                    }
                    // epilogue of the quad: rows in pairs, one 16-byte store per lane and pair (c3args.h)
#pragma unroll
                    for (int mp = 0; mp < 4; mp += 2) {
                        const int yl = y0 + 4 * q + mp + (fk & 1);
                        const int64_t pl = ((int64_t)n * d.H + yl) * d.W + x0 + fr;
                        const int qw = g * 16 + 8 * (fk >> 1);
                        epi_pair_wide(a, d, g * 16 + 4 * fk, qw, acc[mp], acc[mp + 1], want_stats, s1, s2,
                                      a.dst + pl * d.dst_pitch + qw + (qw >= dsplit_c ? dsplit_adj : 0), true);
                    }
                }
            }
            crb += 16; if (crb >= R) crb -= R;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)dsm;                      // [12][2][BQ]
        __syncthreads();
        for (int i = tid; i < 12 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = row16_sum(s1[r]), x2 = row16_sum(s2[r]);
            if (fr == 0) {
                mine[g * 16 + 4 * fk + r] = x1;
                mine[BQ + g * 16 + 4 * fk + r] = x2;
            }
        }
        __syncthreads();
        stats_publish(sst, 12, BQ, tid, 0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// true when the strip-streaming kernel serves this descriptor (conv3x3_applicable has already passed)
bool conv3x3s_applicable(const aau_conv_desc* d, const void* src, const void* dst) {
    if (getenv("AAU_NO_C3S")) return false;
    if (!((d->Cin == 48 || d->Cin == 96) && (d->Cout == 48 || d->Cout == 96))) return false;
    if (d->Cpad != (d->Cin == 48 ? 64 : 96) || d->accumulate) return false;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || d->src_pitch % 8 || d->dst_pitch % 8) return false;
    if (d->src_split_c > 0 && (d->src_split_c % 8 || d->src_split_off % 8)) return false;
    if (d->dst_split_c > 0 && (d->dst_split_c % 8 || d->dst_split_off % 8)) return false;
    int minp = 1;
    if (const char* e = getenv("AAU_C3S_MINPATCH")) minp = atoi(e);
    return (int64_t)d->N * (d->H / 16) * (d->W / 16) >= minp;
}

int conv3x3s_launch(C3Args& a, hipStream_t s) {
    const aau_conv_desc& d = a.d;
    const int strips = d.W / 16, tiles_y = d.H / 16;
    // cut the strips into vertical segments until every CU has a unit (each segment restarts the row stream: at least
    // two patches per segment where the image allows it)
    int nseg = 1;
    while ((int64_t)d.N * strips * nseg < 256 && tiles_y / (nseg * 2) >= 2) nseg *= 2;
    const int segh = (tiles_y + nseg - 1) / nseg;
    nseg = (tiles_y + segh - 1) / segh;
    const int64_t nunits = (int64_t)d.N * strips * nseg;
    if (nunits > 0x7fffffff) { set_error("conv3x3s: too many units"); return AAU_E_INVALID; }
    const int grid = nunits < 256 ? (int)nunits : 256;
    const bool c48 = d.Cin == 48, g3 = d.Cout == 48;
    const size_t lds = (c48 ? (size_t)64 * 2048 : (size_t)36 * 4096) + 1024;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv3x3s_kernel<48, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)conv3x3s_kernel<48, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)conv3x3s_kernel<96, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)conv3x3s_kernel<96, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    prof_tag(c48 ? (g3 ? "conv3x3s<48,48>" : "conv3x3s<48,96>") : (g3 ? "conv3x3s<96,48>" : "conv3x3s<96,96>"));
    if (c48 && g3) hipLaunchKernelGGL((conv3x3s_kernel<48, 3>), dim3(grid), dim3(768), lds, s, a, (int)nunits, strips, nseg, segh);
    else if (c48) hipLaunchKernelGGL((conv3x3s_kernel<48, 6>), dim3(grid), dim3(768), lds, s, a, (int)nunits, strips, nseg, segh);
    else if (g3) hipLaunchKernelGGL((conv3x3s_kernel<96, 3>), dim3(grid), dim3(768), lds, s, a, (int)nunits, strips, nseg, segh);
    else hipLaunchKernelGGL((conv3x3s_kernel<96, 6>), dim3(grid), dim3(768), lds, s, a, (int)nunits, strips, nseg, segh);
    return check_launch("aau_conv_igemm(3x3 strips, register-resident weights)");
}

}  // namespace aau

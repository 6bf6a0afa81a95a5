// Halo-tiled 3x3 (stride 1, pad 1, dilation 1) convolution on MFMA for gfx950: forward and
// data-gradient of every ConvBNReLU of the network (pipeline:63 via :113-121) -- 80 % of its FLOPs.
//
// The generic implicit-GEMM kernel (igemm.hip) re-gathers the activation tile for each of the
// 9 taps, so its L2 -> LDS fill traffic is 9x the tile; at 128x96 tiles that fill rate, not
// the MFMA, bounds it.  Here a workgroup owns a 16x16 spatial patch (256 output pixels) x BQ
// output channels.  Per 32-channel chunk the 18x18 halo of the patch is staged ONCE
// (buffer_load ... lds, zero fill of the image border by the descriptor range check) and all
// 9 taps read their shifted 16-pixel rows out of it; only the small weight tile
// [BQ][32] is streamed per (chunk, tap) step through a 3-slot ring.
//
// Pipeline: loads run 2 steps (weights) / 9 steps (next halo) ahead and stay in flight across
// the per-step barrier: counted s_waitcnt vmcnt(N) + raw s_barrier, never vmcnt(0) in the loop.
// Every wave issues the same number of LDS-DMA instructions per tile (the last one of a weight
// tile is lane-masked) so the counts are wave-uniform constants.
// 4 waves, each 4 patch rows x 16 px x BQ channels (24 / 12 accumulator tiles of
// v_mfma_f32_16x16x32_bf16); 66 / 57 KiB LDS -> 2 workgroups per CU.
#include <stdlib.h>
#include "common.h"
#include "c3args.h"

namespace aau {

// s_waitcnt takes an immediate; the pipeline below only ever needs these four counts per tile shape
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 8, "count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// PW = patch width in 16-pixel columns.  PW = 2 (512 threads, 16 x 32 patch, one workgroup per CU) shares
// every weight tile and the halo between 8 waves: half the LDS-DMA instructions per wave and half the
// weight traffic per FLOP at the same 2 waves per SIMD.
template <int BQ, int PW>
__global__ __launch_bounds__(256 * PW) void conv3x3_kernel(const C3Args a) {
    constexpr int BK = 32;
    constexpr int NWAVE = 4 * PW;
    constexpr int HW_ = 16 * PW + 2;        // halo width
    constexpr int HROWS = 18 * HW_;         // halo pixels: 324 / 612
    constexpr int HL = (HROWS + 16 * NWAVE - 1) / (16 * NWAVE);   // halo LDS-DMA instructions per wave: 6 / 5
    constexpr int HPAD = HL * NWAVE * 16;   // rows staged (rows >= HROWS are zero fill)
    constexpr int NI = BQ / 16;             // channel tiles per wave
    constexpr int MI = 4;                   // patch rows per wave
    constexpr int WROWS = BQ / NWAVE;       // weight-tile rows staged by one wave: 24 / 12 / 6
    constexpr int WL = (WROWS + 15) / 16;   // weight-tile LDS-DMA instructions per wave
    constexpr int NS = 3;                   // weight ring slots (deeper rings measured slower: not latency bound)
    constexpr int PD = NS - 1;              // weight tiles are issued PD steps ahead
    constexpr int HALO_E = HPAD * BK;       // elements
    constexpr int WT_E = BQ * BK;
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * HALO_E + NS * WT_E];
    auto sH = [&](int b) -> unsigned short* { return smem + b * HALO_E; };
    auto sWt = [&](int slot) -> unsigned short* { return smem + 2 * HALO_E + slot * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // tile order: channel tile fastest, then patches; bijective XCD remap (see igemm.hip)
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int nwg = gridDim.x;
    int bid = a.rev ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq;
    int patch = bid / ntq;
    const int px_t = patch % a.tiles_x;
    patch /= a.tiles_x;
    const int py_t = patch % a.tiles_y;
    const int n = patch / a.tiles_y;
    const int q0 = tq * BQ, y0 = py_t * 16, x0 = px_t * 16 * PW;
    const int wr = wave & 3, wcol = wave >> 2;   // wave -> 4 patch rows x one 16-pixel column

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // ---- halo: this lane's (row, chunk slot) per instruction; byte offset of channel 0 or OOB ----
    unsigned hoff[HL];
    bool htail[HL];
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const int hr = (i * NWAVE + wave) * 16 + (lane >> 2);
        const int lc = swz32(hr, lane & 3);
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = hr < HROWS && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
        hoff[i] = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc * 8) * 2) : OOB;
        htail[i] = tail_c0 + lc * 8 < d.Cin;
    }
    const bool has_tail = d.Cpad != d.Cin;
    // ---- weights: rows of this wave's share of the [BQ][32] tile ----
    unsigned woff[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int row = wave * WROWS + j * 16 + (lane >> 2);
        const int lc = swz32(row, lane & 3);
        const bool ok = (j * 16 + (lane >> 2)) < WROWS && (q0 + row) < d.Cout;
        woff[j] = ok ? (unsigned)(((q0 + row) * 9 * d.Cpad + lc * 8) * 2) : OOB;
    }
    constexpr int WLAST = WROWS - (WL - 1) * 16;   // rows covered by the last instruction: 8 or 12
    auto issue_halo = [&](int chunk) {
        const bool last = has_tail && chunk == a.nchunk - 1;
        unsigned short* base = sH(chunk & 1);
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            const unsigned v = (last && !htail[i]) ? OOB : hoff[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base + (i * NWAVE + wave) * 16 * BK), 16, (int)v,
                                                     chunk * BK * 2, 0, 0);
        }
    };
    auto issue_w = [&](int slot, int chunk, int tap) {
        const int soff = (tap * d.Cpad + chunk * BK) * 2;
        unsigned short* base = sWt(slot) + wave * WROWS * BK;
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            if (j < WL - 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(base + j * 16 * BK), 16, (int)woff[j], soff, 0, 0);
            } else if (lane < WLAST * 4) {   // lane-masked tail: still ONE instruction per wave
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(base + j * 16 * BK), 16, (int)woff[j], soff, 0, 0);
            }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    // halo (B operand) fragments of a step can be fetched one step early: the halo of the current chunk
    // landed long ago, only the weight tile of a step needs that step's barrier.
    bf16x8 af[MI];
    auto load_halo_frags = [&](int hb, int tap) {
        const int ty = tap / 3, tx = tap - ty * 3;
        const unsigned short* hbase = sH(hb);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int hr = (wr * MI + mi + ty) * HW_ + wcol * 16 + fr + tx;
            af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
        }
    };
    auto compute = [&](int slot, int next_hb, int next_tap, bool prefetch) {
        bf16x8 wf[NI];
        const unsigned short* wbase = sWt(slot);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = ni * 16 + fr;
            wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
        }
        bf16x8 cur[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) cur[mi] = af[mi];
#ifdef C3_PREFETCH
        if (prefetch) load_halo_frags(next_hb, next_tap);
#endif
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = AAU_MFMA16(wf[ni], cur[mi], acc[ni][mi], 0, 0, 0);
    };

    // ---- software pipeline over S = nchunk*9 (chunk, tap) steps ----
    // issue order inside a step (after its barrier): [halo of the next chunk, at tap 0], then w(s+PD).
    const int S = a.nchunk * 9;
    issue_halo(0);
    {
        int c = 0, t = 0;
        for (int k = 0; k < PD && k < S; ++k) {
            issue_w(k, c, t);
            if (++t == 9) { t = 0; ++c; }
        }
    }
    int chunk = 0, tap = 0, slot = 0;
    int ctap = PD % 9, cchunk = PD / 9, cslot = PD % NS;   // (chunk, tap, slot) of step s + PD
    for (int s = 0; s < S; ++s) {
        // outstanding loads younger than w(s): w(s+1 .. s+PD-1) that exist, plus the halo of the next
        // chunk if it was issued during one of the last PD-1 steps (i.e. at this chunk's tap 0)
        int nw = S - 1 - s;
        if (nw > PD - 1) nw = PD - 1;
        const bool halo_recent = tap >= 1 && tap <= PD - 1 && chunk + 1 < a.nchunk;
        static_assert(PD == 2, "the literal wait counts below assume a 2-step weight prefetch");
        if (halo_recent) { if (nw) wait_vm<HL + WL>(); else wait_vm<HL>(); }
        else             { if (nw) wait_vm<WL>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
#ifdef ABL_STAMP
        if (s == 0) tst[1] = __builtin_amdgcn_s_memtime();
#endif
        if (tap == 0 && chunk + 1 < a.nchunk) issue_halo(chunk + 1);
        if (s + PD < S) issue_w(cslot, cchunk, ctap);
#ifndef C3_PREFETCH
        load_halo_frags(chunk & 1, tap);
#else
        if (s == 0) load_halo_frags(0, 0);
#endif
        {
            // the next step's halo fragments: same chunk (taps 1..8) or, at tap 8, the next chunk's buffer,
            // whose LDS-DMA was waited for before THIS step's barrier only if it was issued >= 2 steps ago:
            // it was issued at tap 0 of this chunk, 8 steps back -> landed and visible.
            const bool pf = s + 1 < S;
            const int ntap = tap == 8 ? 0 : tap + 1;
            const int nhb = tap == 8 ? ((chunk + 1) & 1) : (chunk & 1);
            compute(slot, nhb, ntap, pf);
        }
        if (++tap == 9) { tap = 0; ++chunk; }
        if (++slot == NS) slot = 0;
        if (++ctap == 9) { ctap = 0; ++cchunk; }
        if (++cslot == NS) cslot = 0;
    }

#ifdef ABL_STAMP
    tst[2] = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue (same contract as igemm.hip) ----
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int y = y0 + wr * MI + mi, x = x0 + wcol * 16 + fr;
        const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + ni * 16 + 4 * fk;
            if (q >= d.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
            if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
            if (a.bias) {
                const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b[r];
            }
            if (a.scale) {
                const f32x4 sc = *(const f32x4*)(a.scale + q);
                const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
            }
            unsigned short* out = a.dst + pixel * d.dst_pitch + q;
            if (d.accumulate) {
                const u32x2 old = *(const u32x2*)out;
                v[0] += pair_lo(old[0]);
                v[1] += pair_hi(old[0]);
                v[2] += pair_lo(old[1]);
                v[3] += pair_hi(old[1]);
            }
            if (d.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2 pk;
            pk[0] = pack2(v[0], v[1]);
            pk[1] = pack2(v[2], v[3]);
#ifdef ABL_NOSTORE
            if (pk[0] == 0x12345678u && pk[1] == 0x9abcdef0u)
#endif
            *(u32x2*)out = pk;
        }
    }
    if (want_stats) {
        // per-wave row sums (DPP), combined across the 4 waves in LDS (the tiles are dead now),
        // then ONE global atomic per channel and workgroup
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)smem;                      // [NWAVE][2][BQ]
        __syncthreads();                                // every wave is done reading the LDS tiles
        for (int i = tid; i < NWAVE * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, NWAVE, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
#ifdef ABL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tst[3] = __builtin_amdgcn_s_memtime();
    if (tid == 0 && a.shift != nullptr && a.scale == nullptr) {
        unsigned long long* dbg = (unsigned long long*)a.shift + (size_t)blockIdx.x * 4;
        dbg[0] = tst[0]; dbg[1] = tst[1]; dbg[2] = tst[2]; dbg[3] = tst[3];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Grouped-tap variant of the halo kernel: a pipeline step covers 2 taps (groups {0,1} {2,3} {4,5} {6,7} {8})
// instead of one, so a 32-channel chunk costs 5 barriers / waits / fragment-latency exposures instead of 9,
// with 48 MFMAs per wave between them.  Weight ring: 2 slots of 2 tap tiles, halo rows trimmed to 336:
// 67 KB, two workgroups per CU as before.
template <int BQ>
__global__ __launch_bounds__(256) void conv3x3g_kernel(const C3Args a) {
    constexpr int BK = 32, HW_ = 18, HROWS = 324, HPAD = 336, NI = BQ / 16, MI = 4;
    constexpr int WROWS = BQ / 4, WL = (WROWS + 15) / 16, WLAST = WROWS - (WL - 1) * 16;
    constexpr int HALO_E = HPAD * BK, WT_E = BQ * BK, SLOT_E = 2 * WT_E;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * HALO_E + 2 * SLOT_E];
    auto sH = [&](int b) -> unsigned short* { return smem + b * HALO_E; };
    auto sWt = [&](int slot, int k) -> unsigned short* { return smem + 2 * HALO_E + slot * SLOT_E + k * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int nwg = gridDim.x;
    int bid = a.rev ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq;
    int patch = bid / ntq;
    const int px_t = patch % a.tiles_x;
    patch /= a.tiles_x;
    const int py_t = patch % a.tiles_y;
    const int n = patch / a.tiles_y;
    const int q0 = tq * BQ, y0 = py_t * 16, x0 = px_t * 16;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // halo roles: 21 wave-instructions of 16 rows; wave 0 issues 6, the others 5
    const int hl = (wave == 0) ? 6 : 5;
    unsigned hoff[6];
    bool htail[6];
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hr = (i * 4 + wave) * 16 + (lane >> 2);
        const int lc = swz32(hr, lane & 3);
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = hr < HROWS && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
        hoff[i] = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc * 8) * 2) : OOB;
        htail[i] = tail_c0 + lc * 8 < d.Cin;
    }
    const bool has_tail = d.Cpad != d.Cin;
    unsigned woff[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int row = wave * WROWS + j * 16 + (lane >> 2);
        const int lc = swz32(row, lane & 3);
        const bool ok = (j * 16 + (lane >> 2)) < WROWS && (q0 + row) < d.Cout;
        woff[j] = ok ? (unsigned)(((q0 + row) * 9 * d.Cpad + lc * 8) * 2) : OOB;
    }
    auto issue_halo = [&](int chunk) {
        const bool last = has_tail && chunk == a.nchunk - 1;
        unsigned short* base = sH(chunk & 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < hl) {   // wave-uniform
                const unsigned v = (last && !htail[i]) ? OOB : hoff[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base + (i * 4 + wave) * 16 * BK), 16, (int)v,
                                                         chunk * BK * 2, 0, 0);
            }
        }
    };
    // group g of a chunk = taps {2g, 2g+1} for g < 4, {8} for g == 4
    auto issue_w = [&](int slot, int chunk, int g) {
        const int t0 = 2 * g, nt = (g == 4) ? 1 : 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < nt) {
                const int soff = ((t0 + k) * d.Cpad + chunk * BK) * 2;
                unsigned short* base = sWt(slot, k) + wave * WROWS * BK;
#pragma unroll
                for (int j = 0; j < WL; ++j) {
                    if (j < WL - 1) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(base + j * 16 * BK), 16, (int)woff[j], soff, 0, 0);
                    } else if (lane < WLAST * 4) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(base + j * 16 * BK), 16, (int)woff[j], soff, 0, 0);
                    }
                }
            }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    auto compute_tap = [&](const unsigned short* hbase, const unsigned short* wbase, int tap) {
        const int ty = tap / 3, tx = tap - ty * 3;
        bf16x8 wf[NI], af[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = ni * 16 + fr;
            wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int hr = (wave * MI + mi + ty) * HW_ + fr + tx;
            af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
        }
        // raised priority keeps the MFMA cluster together (A/B: +3-7 % here; the same pair costs
        // wgrad3x3_kernel 6-9 %, so it is not used there)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- pipeline over S = nchunk * 5 steps; weights one step ahead, next halo issued AFTER them at group 0 ----
    const int S = a.nchunk * 5;
    issue_halo(0);
    issue_w(0, 0, 0);
    int chunk = 0, g = 0, slot = 0;
    bool halo_prev = false;     // the previous step issued a halo tile (after its weight tiles)
    for (int s = 0; s < S; ++s) {
        if (halo_prev) {        // only that halo tile may stay in flight
            if (wave == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        halo_prev = false;
        if (s + 1 < S) {
            int g2 = g + 1, c2 = chunk;
            if (g2 == 5) { g2 = 0; ++c2; }
            issue_w(slot ^ 1, c2, g2);
        }
        if (g == 0 && chunk + 1 < a.nchunk) { issue_halo(chunk + 1); halo_prev = true; }
        const unsigned short* hbase = sH(chunk & 1);
        compute_tap(hbase, sWt(slot, 0), 2 * g);
        if (g < 4) compute_tap(hbase, sWt(slot, 1), 2 * g + 1);
        if (++g == 5) { g = 0; ++chunk; }
        slot ^= 1;
    }

    // ---- epilogue (as conv3x3_kernel) ----
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int y = y0 + wave * MI + mi, x = x0 + fr;
        const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + ni * 16 + 4 * fk;
            if (q >= d.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
            if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
            if (a.bias) {
                const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b[r];
            }
            if (a.scale) {
                const f32x4 sc = *(const f32x4*)(a.scale + q);
                const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
            }
            unsigned short* out = a.dst + pixel * d.dst_pitch + q;
            if (d.accumulate) {
                const u32x2 old = *(const u32x2*)out;
                v[0] += pair_lo(old[0]);
                v[1] += pair_hi(old[0]);
                v[2] += pair_lo(old[1]);
                v[3] += pair_hi(old[1]);
            }
            if (d.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2 pk;
            pk[0] = pack2(v[0], v[1]);
            pk[1] = pack2(v[2], v[3]);
            *(u32x2*)out = pk;
        }
    }
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)smem;                      // [4][2][BQ]
        __syncthreads();
        for (int i = tid; i < 4 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, 4, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// ------------------------------------------------------------------------------------------------
// Column-step variant of the halo kernel (conv3x3h): the LDS READ traffic is what the grouped-tap kernel runs into.
// There every wave owns 4 patch rows x BQ channels and reads, per tap, 4 pixel fragments + BQ/16 weight fragments for
// 24 MFMAs: with two workgroups per CU that is 10 KB x 8 waves = 640 LDS cycles per tap against 768 MFMA cycles per
// SIMD, plus the LDS-DMA writes -- LDS and matrix pipes co-limited.  Here
//   * the four waves are a 2 x 2 grid: 8 patch rows x BQ/2 channels each (the same 24 accumulator tiles), and
//   * a pipeline step is one tap COLUMN tx (taps tx, 3+tx, 6+tx): the 10 halo-row fragments a wave needs for the
//     three vertical taps are read once and reused, only the 3 x BQ/32 weight fragments change,
// so a step reads 10 + 9 fragments for 72 MFMAs (30 before): -37 % LDS reads, 3 barriers per chunk instead of 5.
// The loop body is unrolled over 2 chunks x 3 columns, which makes every LDS address a loop-invariant register plus an
// immediate (the swizzled halo offsets of the 30 (column, row) fragments are computed once; the first version spent
// ~150 VALU instructions per step recomputing them and only then started its MFMAs), and the fragment reads and the
// next step's LDS-DMA are interleaved INTO the MFMA stream (sched_group_barrier) instead of in front of it.
// Every wave issues the same number of LDS-DMA instructions (the spare ones are all-out-of-range pieces that land in
// a 1-KB scratch area), so the loop has no divergent code and the vmcnt counts are constants.
// LDS: two halo buffers (42 KB) + 2 slots of 3 tap tiles (36 KB) + scratch = 79 KB -> two workgroups per CU.
// Where the time goes (a fixed-shape micro-benchmark + the AAU_C3_ABL / AAU_C3_NOSTORE timing ablations, 8 x 256 x 256,
// 384 -> 96 channels): the bare read + MFMA + barrier loop runs at 1741 TFLOP/s; the weight LDS-DMA costs 9 %, the
// halo LDS-DMA 12 %, the output stores 5 %, the statistics epilogue 3-5 %: 1300 as shipped.  ~5 us per workgroup
// (address tables, first round trip, epilogue) do not shrink with Cin: conv3x3p_kernel below is the persistent form
// that hides them under the previous tile's epilogue (the default; this kernel stays as its A/B partner).
template <int BQ>
__global__ __launch_bounds__(256, 2) void conv3x3h_kernel(const C3Args a) {
    static_assert(BQ % 32 == 0, "two channel halves of whole 16-channel tiles");
    constexpr int BK = 32, HW_ = 18, HROWS = 324, NI = BQ / 32, MI = 8, QH = BQ / 2;
    constexpr int HPIECES = 21;                    // 16-row pieces of a halo tile (336 rows staged, 324 used)
    constexpr int WPIECES = 3 * BQ / 16;           // 16-row pieces of a column's three tap tiles
    constexpr int WLW = (WPIECES + 3) / 4;         // ... per wave
    constexpr int HALO_E = HPIECES * 16 * BK, WT_E = BQ * BK, SLOT_E = 3 * WT_E;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* smem = smem_h;
    unsigned short* const scratch = smem + 2 * HALO_E + 2 * SLOT_E;     // 512 elements

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int rh = wave >> 1, qh = wave & 1;
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int nwg = gridDim.x;
    int bid = a.rev ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq;
    int patch = bid / ntq;
    const int px_t = patch % a.tiles_x;
    patch /= a.tiles_x;
    const int py_t = patch % a.tiles_y;
    const int n = patch / a.tiles_y;
    const int q0 = tq * BQ, y0 = py_t * 16, x0 = px_t * 16;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // halo pieces p = i * 4 + wave (i < 6): 21 real ones, the three spare ones are out of range and land in `scratch`
    unsigned hoff[6];
    unsigned tailmask = 0;      // bit i: this lane's 8 channels of piece i exist in the LAST chunk
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hr = (i * 4 + wave) * 16 + (lane >> 2);
        const int lc = swz32(hr, lane & 3);
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = hr < HROWS && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
        hoff[i] = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc * 8) * 2) : OOB;
        if (tail_c0 + lc * 8 < d.Cin) tailmask |= 1u << i;
    }
    const bool has_tail = d.Cpad != d.Cin;
    // weight pieces p = j * 4 + wave (j < WLW): tap row k = p / (BQ/16), 16 output channels each
    unsigned woff[WLW];
#pragma unroll
    for (int j = 0; j < WLW; ++j) {
        const int p = j * 4 + wave;
        const int k = p / (BQ / 16), row = (p - k * (BQ / 16)) * 16 + (lane >> 2);
        const int lc = swz32(row, lane & 3);
        const bool ok = p < WPIECES && (q0 + row) < d.Cout;
        woff[j] = ok ? (unsigned)((((q0 + row) * 9 + 3 * k) * d.Cpad + lc * 8) * 2) : OOB;
    }
    auto issue_halo = [&](int chunk, int buf) {
        const bool dead = chunk >= a.nchunk;
        const bool last = has_tail && chunk == a.nchunk - 1;
        unsigned short* base = smem + buf * HALO_E;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = i * 4 + wave;
            const unsigned v = (dead || (last && !((tailmask >> i) & 1))) ? OOB : hoff[i];
            unsigned short* dst = (p < HPIECES) ? base + p * 16 * BK : scratch;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst), 16, (int)v, chunk * BK * 2, 0, 0);
        }
    };
    // column tx of a chunk = taps {tx, 3 + tx, 6 + tx}
    auto issue_w = [&](int slot, int chunk, int tx) {
        const bool dead = chunk >= a.nchunk;
        const int soff = (tx * d.Cpad + chunk * BK) * 2;
        unsigned short* base = smem + 2 * HALO_E + slot * SLOT_E;
#pragma unroll
        for (int j = 0; j < WLW; ++j) {
            const int p = j * 4 + wave;
            unsigned short* dst = (p < WPIECES) ? base + p * 16 * BK : scratch;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst), 16, (int)(dead ? OOB : woff[j]), soff, 0, 0);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    // loop-invariant LDS element offsets: pixel fragments (column tx, halo row rh*8 + r) inside a halo buffer ...
    int aoff[3][MI + 2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < MI + 2; ++r) {
            const int hr = (rh * MI + r) * HW_ + fr + t;
            aoff[t][r] = hr * BK + swz32(hr, fk) * 8;
        }
    // ... and this lane's weight fragment inside a tap tile ((row >> 2) & 3 == (fr >> 2) & 3 for every channel tile)
    const int boff = 2 * HALO_E + (qh * QH + fr) * BK + swz32(fr, fk) * 8;

    // one pipeline step, all template-like arguments compile-time after unrolling
    auto step = [&](const int buf, const int tx, const int slot, int chunk, const bool prev_halo) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);      // the previous step's MFMAs stay in front of this wait
        if (prev_halo) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // only that halo tile may stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short* hb = smem + buf * HALO_E;
        const unsigned short* wb = smem + boff + slot * SLOT_E;
        bf16x8 af[MI + 2], wf[3][NI];
        // program order = order of need
        wf[0][0] = *(const bf16x8*)(wb);
#pragma unroll
        for (int r = 0; r < MI; ++r) af[r] = *(const bf16x8*)(hb + aoff[tx][r]);
#pragma unroll
        for (int ni = 1; ni < NI; ++ni) wf[0][ni] = *(const bf16x8*)(wb + ni * 16 * BK);
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            wf[k][0] = *(const bf16x8*)(wb + k * WT_E);
            af[MI - 1 + k] = *(const bf16x8*)(hb + aoff[tx][MI - 1 + k]);
#pragma unroll
            for (int ni = 1; ni < NI; ++ni) wf[k][ni] = *(const bf16x8*)(wb + k * WT_E + ni * 16 * BK);
        }
        // next column's weights (the slot read in the previous step), then -- at column 0 -- the next chunk's halo
        {
            int t2 = tx + 1, c2 = chunk;
            if (t2 == 3) { t2 = 0; ++c2; }
            if (!(a.rev & 4)) issue_w(slot ^ 1, c2, t2);      // (rev & 4 / 8: timing-only ablations, AAU_C3_ABL)
        }
        if (tx == 0 && !(a.rev & 8)) issue_halo(chunk + 1, buf ^ 1);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = AAU_MFMA16(wf[k][ni], af[mi + k], acc[ni][mi], 0, 0, 0);
        // schedule: 4 reads, then one read per MFMA until the 10 + 3*NI fragments are in, then one LDS-DMA per 4 MFMAs
        constexpr int NRD = MI + 2 + 3 * NI, NMF = 3 * NI * MI;
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < NRD - 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (tx == 0) {
#pragma unroll
            for (int i = 0; i < WLW + 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - (NRD - 4) - 4 * (WLW + 6), 0);
        } else {
#pragma unroll
            for (int i = 0; i < WLW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - (NRD - 4) - 4 * WLW, 0);
        }
    };

    issue_halo(0, 0);
    issue_w(0, 0, 0);
    for (int c0 = 0; c0 < a.nchunk; c0 += 2) {
        step(0, 0, 0, c0, false);
        step(0, 1, 1, c0, true);
        step(0, 2, 0, c0, false);
        if (c0 + 1 >= a.nchunk) break;
        step(1, 0, 1, c0 + 1, false);
        step(1, 1, 0, c0 + 1, true);
        step(1, 2, 1, c0 + 1, false);
    }
    // the out-of-range pieces issued by the last steps still write zeros into LDS: drain before LDS is reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue ----
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int y = y0 + rh * MI + mi, x = x0 + fr;
        const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + qh * QH + ni * 16 + 4 * fk;
            if (q >= d.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
            if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
            if (a.bias) {
                const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b[r];
            }
            if (a.scale) {
                const f32x4 sc = *(const f32x4*)(a.scale + q);
                const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
            }
            unsigned short* out = a.dst + pixel * d.dst_pitch + q;
            if (d.accumulate) {
                const u32x2 old = *(const u32x2*)out;
                v[0] += pair_lo(old[0]);
                v[1] += pair_hi(old[0]);
                v[2] += pair_lo(old[1]);
                v[3] += pair_hi(old[1]);
            }
            if (d.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2 pk;
            pk[0] = pack2(v[0], v[1]);
            pk[1] = pack2(v[2], v[3]);
            if (!(a.rev & 2)) *(u32x2*)out = pk;   // (rev & 2: timing-only ablation, AAU_C3_NOSTORE)
        }
    }
    if (want_stats) {
        // every wave stores the row sums of ITS channel half in its own block (zero elsewhere); fixed wave order
        float* sst = (float*)smem;                      // [4][2][BQ]
        __syncthreads();
        for (int i = tid; i < 4 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ + qh * QH;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, 4, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the column-step kernel (conv3x3p; AAU_C3_NOPERSIST=1 switches back): G resident workgroups walk over tiles
// t, t + G, ... and the chunk / column pipeline runs THROUGH the tile boundary -- the last chunk of a tile issues the
// first halo and the first weight column of the workgroup's next tile, so that round trip and the address set-up hide
// under the epilogue.  Statistics are published per tile through the halo buffer the finished tile no longer needs.
template <int BQ>
__global__ __launch_bounds__(256, 2) void conv3x3p_kernel(const C3Args a, int ntiles) {
    static_assert(BQ % 32 == 0, "two channel halves of whole 16-channel tiles");
    constexpr int BK = 32, HW_ = 18, HROWS = 324, NI = BQ / 32, MI = 8, QH = BQ / 2;
    constexpr int HPIECES = 21, WPIECES = 3 * BQ / 16, WLW = (WPIECES + 3) / 4;
    constexpr int HALO_E = HPIECES * 16 * BK, WT_E = BQ * BK, SLOT_E = 3 * WT_E;
    constexpr int NST = NI * MI;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_p[];
    unsigned short* smem = smem_p;
    unsigned short* const scratch = smem + 2 * HALO_E + 2 * SLOT_E;     // 512 elements

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int rh = wave >> 1, qh = wave & 1;
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int G = gridDim.x;                          // multiple of ntq (host)
    int bid = (int)blockIdx.x;
    {
        const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq, q0 = tq * BQ;           // the same for every tile of this workgroup
    const bool full_q = q0 + BQ <= d.Cout;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    unsigned tailmask = 0;
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hr = (i * 4 + wave) * 16 + (lane >> 2);
        if (tail_c0 + swz32(hr, lane & 3) * 8 < d.Cin) tailmask |= 1u << i;
    }
    const bool has_tail = d.Cpad != d.Cin;
    unsigned hoff[6];
    auto tile_coords = [&](int tile, int& n, int& y0, int& x0) {
        if (a.rev) tile = ntiles - 1 - tile;
        int patch = tile / ntq;
        const int px_t = patch % a.tiles_x;
        patch /= a.tiles_x;
        const int py_t = patch % a.tiles_y;
        n = patch / a.tiles_y; y0 = py_t * 16; x0 = px_t * 16;
    };
    auto set_hoff = [&](int tile) {        // tile >= ntiles: nothing to fetch (all out of range)
        int n, y0, x0;
        tile_coords(tile < ntiles ? tile : 0, n, y0, x0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int hr = (i * 4 + wave) * 16 + (lane >> 2);
            const int lc = swz32(hr, lane & 3);
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = tile < ntiles && hr < HROWS && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            hoff[i] = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc * 8) * 2) : OOB;
        }
    };
    unsigned woff[WLW];
#pragma unroll
    for (int j = 0; j < WLW; ++j) {
        const int p = j * 4 + wave;
        const int k = p / (BQ / 16), row = (p - k * (BQ / 16)) * 16 + (lane >> 2);
        const int lc = swz32(row, lane & 3);
        const bool ok = p < WPIECES && (q0 + row) < d.Cout;
        woff[j] = ok ? (unsigned)((((q0 + row) * 9 + 3 * k) * d.Cpad + lc * 8) * 2) : OOB;
    }
    auto issue_halo = [&](int chunk, int buf) {
        const bool last = has_tail && chunk == a.nchunk - 1;
        unsigned short* base = smem + buf * HALO_E;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = i * 4 + wave;
            const unsigned v = (last && !((tailmask >> i) & 1)) ? OOB : hoff[i];
            unsigned short* dst = (p < HPIECES) ? base + p * 16 * BK : scratch;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst), 16, (int)v, chunk * BK * 2, 0, 0);
        }
    };
    auto issue_w = [&](int slot, int chunk, int tx, bool dead) {
        const int soff = (tx * d.Cpad + chunk * BK) * 2;
        unsigned short* base = smem + 2 * HALO_E + slot * SLOT_E;
#pragma unroll
        for (int j = 0; j < WLW; ++j) {
            const int p = j * 4 + wave;
            unsigned short* dst = (p < WPIECES) ? base + p * 16 * BK : scratch;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst), 16, (int)(dead ? OOB : woff[j]), soff, 0, 0);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    int aoff[3][MI + 2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < MI + 2; ++r) {
            const int hr = (rh * MI + r) * HW_ + fr + t;
            aoff[t][r] = hr * BK + swz32(hr, fk) * 8;
        }
    const int boff = 2 * HALO_E + (qh * QH + fr) * BK + swz32(fr, fk) * 8;
    const bool want_stats = a.stats != nullptr;
    // (16-byte epilogue stores through epi_pair_wide were tried here: -3 % ... +4 % per layer, no net gain -- this kernel
    // is at its register limit and the second epilogue form spills; its store tail is ~5 % of a launch)

    auto step = [&](const int buf, const int tx, const int slot, int chunk, int nc, const bool prev_halo, bool after_store,
                    bool wdead) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (prev_halo) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (after_store) {            // the finished tile's stores may stay in flight; what they followed is back
            if constexpr (NST == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short* hb = smem + buf * HALO_E;
        const unsigned short* wb = smem + boff + slot * SLOT_E;
        bf16x8 af[MI + 2], wf[3][NI];
        wf[0][0] = *(const bf16x8*)(wb);
#pragma unroll
        for (int r = 0; r < MI; ++r) af[r] = *(const bf16x8*)(hb + aoff[tx][r]);
#pragma unroll
        for (int ni = 1; ni < NI; ++ni) wf[0][ni] = *(const bf16x8*)(wb + ni * 16 * BK);
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            wf[k][0] = *(const bf16x8*)(wb + k * WT_E);
            af[MI - 1 + k] = *(const bf16x8*)(hb + aoff[tx][MI - 1 + k]);
#pragma unroll
            for (int ni = 1; ni < NI; ++ni) wf[k][ni] = *(const bf16x8*)(wb + k * WT_E + ni * 16 * BK);
        }
        if (tx == 2) issue_w(slot ^ 1, nc, 0, wdead);
        else issue_w(slot ^ 1, chunk, tx + 1, false);
        if (tx == 0) issue_halo(nc, buf ^ 1);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = AAU_MFMA16(wf[k][ni], af[mi + k], acc[ni][mi], 0, 0, 0);
        constexpr int NRD = MI + 2 + 3 * NI, NMF = 3 * NI * MI;
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < NRD - 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (tx == 0) {
#pragma unroll
            for (int i = 0; i < WLW + 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - (NRD - 4) - 4 * (WLW + 6), 0);
        } else {
#pragma unroll
            for (int i = 0; i < WLW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - (NRD - 4) - 4 * WLW, 0);
        }
    };

    // epilogue of one tile; `fbuf` = the halo buffer this tile's last chunk used (free now: scratch for the statistics)
    auto epilogue = [&](int tile, int fbuf) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        float s1[NI][4], s2[NI][4];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int y = y0 + rh * MI + mi, x = x0 + fr;
            const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int q = q0 + qh * QH + ni * 16 + 4 * fk;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
                acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (q >= d.Cout) continue;
                if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += b[r];
                }
                if (a.scale) {
                    const f32x4 sc = *(const f32x4*)(a.scale + q);
                    const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
                }
                unsigned short* out = a.dst + pixel * d.dst_pitch + q;
                if (d.accumulate) {
                    const u32x2 old = *(const u32x2*)out;
                    v[0] += pair_lo(old[0]);
                    v[1] += pair_hi(old[0]);
                    v[2] += pair_lo(old[1]);
                    v[3] += pair_hi(old[1]);
                }
                if (d.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 pk;
                pk[0] = pack2(v[0], v[1]);
                pk[1] = pack2(v[2], v[3]);
                *(u32x2*)out = pk;
            }
        }
        if (want_stats) {
            float* sst = (float*)(smem + fbuf * HALO_E);      // [4][2][BQ]
            __syncthreads();                                  // every wave has finished reading that halo buffer
            for (int i = tid; i < 4 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
            __syncthreads();
            float* mine = sst + wave * 2 * BQ + qh * QH;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                    if (fr == 0) {
                        mine[ni * 16 + 4 * fk + r] = x1;
                        mine[BQ + ni * 16 + 4 * fk + r] = x2;
                    }
                }
            }
            __syncthreads();
            stats_publish(sst, 4, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
        }
    };

    // ---- the chunk stream of this workgroup's tiles ----
    int tile = bid, c = 0;
    bool after_store = false;
    const bool count_stores = full_q && !d.accumulate && !want_stats;   // then exactly NST stores follow the last loads
    set_hoff(tile);
    issue_halo(0, 0);
    issue_w(0, 0, 0, false);
    while (true) {
        {   // even chunk of the stream: halo buffer 0, weight slots 0 1 0
            const bool last = c == a.nchunk - 1;
            const int nc = last ? 0 : c + 1;
            if (last) set_hoff(tile + G);                  // the halo issued in this chunk belongs to the next tile
            step(0, 0, 0, c, nc, false, after_store, false);
            after_store = false;
            step(0, 1, 1, c, nc, true, false, false);
            step(0, 2, 0, c, nc, false, false, last && tile + G >= ntiles);
            ++c;
            if (last) {
                epilogue(tile, 0);
                after_store = count_stores;
                tile += G; c = 0;
                if (tile >= ntiles) break;
                if (a.nchunk > 1) set_hoff(tile);          // chunk 1.. of the new tile
            }
        }
        {   // odd chunk of the stream: halo buffer 1, weight slots 1 0 1
            const bool last = c == a.nchunk - 1;
            const int nc = last ? 0 : c + 1;
            if (last) set_hoff(tile + G);
            step(1, 0, 1, c, nc, false, after_store, false);
            after_store = false;
            step(1, 1, 0, c, nc, true, false, false);
            step(1, 2, 1, c, nc, false, false, last && tile + G >= ntiles);
            ++c;
            if (last) {
                epilogue(tile, 1);
                after_store = count_stores;
                tile += G; c = 0;
                if (tile >= ntiles) break;
                if (a.nchunk > 1) set_hoff(tile);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // out-of-range pieces of the last steps still write zeros
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (opt-in, AAU_C3_LOADER=1; measured 25-30 % slower than conv3x3g, see conv3x3_launch):
// the grouped-tap kernel with a DEDICATED LOADER WAVE (wave 4 of 5).  In conv3x3g every wave issues its share of the
// LDS-DMA (~5 buffer_load ... lds per step); inside a wave that also runs 48 MFMAs and 20 ds_read_b128 per step each
// of them costs 100-185 cycles of that wave's in-order stream, i.e. ~40 % on top of the 768 MFMA cycles -- which is
// the measured 44 % MFMA utilisation.  A wave that does nothing else issues a 1-KiB piece every ~20 cycles (guide:
// ldsdma-fill), so one loader wave per workgroup carries all 12 weight + 21 halo pieces of a step, waits for them
// with its OWN vmcnt, and releases the four MFMA waves through the step barrier; the MFMA waves issue no VMEM at all
// in the loop.  154 VGPRs -> 3 waves per SIMD -> two 5-wave workgroups per CU as before (67 KB LDS each).
template <int BQ>
__global__ __launch_bounds__(320) void conv3x3l_kernel(const C3Args a) {
    constexpr int BK = 32, HW_ = 18, HROWS = 324, HPAD = 336, NI = BQ / 16, MI = 4;
    constexpr int HALO_E = HPAD * BK, WT_E = BQ * BK, SLOT_E = 2 * WT_E;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short c3l_smem[];   // 2 halo buffers + 3 weight slots: 78 KB
    unsigned short* smem = c3l_smem;
    auto sH = [&](int b) -> unsigned short* { return smem + b * HALO_E; };
    auto sWt = [&](int slot, int k) -> unsigned short* { return smem + 2 * HALO_E + slot * SLOT_E + k * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int nwg = gridDim.x;
    int bid = a.rev ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq;
    int patch = bid / ntq;
    const int px_t = patch % a.tiles_x;
    patch /= a.tiles_x;
    const int py_t = patch % a.tiles_y;
    const int n = patch / a.tiles_y;
    const int q0 = tq * BQ, y0 = py_t * 16, x0 = px_t * 16;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    const bool loader = wave == 4;
    f32x4 acc[NI][MI];     // zeroed on the MFMA waves' path only: the loader wave must not carry 96 dead registers
    const int fr = lane & 15, fk = lane >> 4;
    auto compute_tap = [&](const unsigned short* hbase, const unsigned short* wbase, int tap) {
        const int ty = tap / 3, tx = tap - ty * 3;
        bf16x8 wf[NI], af[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = ni * 16 + fr;
            wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int hr = (wave * MI + mi + ty) * HW_ + fr + tx;
            af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
        }
        // raised priority keeps the MFMA cluster together (A/B: +3-7 % here; the same pair costs
        // wgrad3x3_kernel 6-9 %, so it is not used there)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- pipeline over S = nchunk * 5 steps; weights one step ahead, next halo issued AFTER them at group 0 ----
    // loader: wait (own vmcnt) until step s's tiles landed -> barrier s -> issue step s+1 (its slot was read in step
    // s-1, which every MFMA wave finished before arriving at barrier s).  MFMA waves: barrier s -> compute step s.
    const int S = a.nchunk * 5;
    int chunk = 0, g = 0, slot = 0;
    if (loader) {
        // halo: 21 wave-instructions of 16 rows, all issued by the loader wave
        constexpr int HI = 21;
        unsigned hoff[HI];
        bool htail[HI];
        const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            const int hr = i * 16 + (lane >> 2);
            const int lc = swz32(hr, lane & 3);
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = hr < HROWS && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            hoff[i] = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc * 8) * 2) : OOB;
            htail[i] = tail_c0 + lc * 8 < d.Cin;
        }
        const bool has_tail = d.Cpad != d.Cin;
        constexpr int WI = BQ / 16;       // weight-tile pieces of 16 rows (BQ = 48 / 96: whole pieces)
        static_assert(BQ % 16 == 0, "weight tile in whole 16-row pieces");
        unsigned woff[WI];
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int row = j * 16 + (lane >> 2);
            const int lc = swz32(row, lane & 3);
            woff[j] = (q0 + row) < d.Cout ? (unsigned)(((q0 + row) * 9 * d.Cpad + lc * 8) * 2) : OOB;
        }
        auto issue_halo = [&](int chunk) {
            const bool last = has_tail && chunk == a.nchunk - 1;
            unsigned short* base = sH(chunk & 1);
#pragma unroll
            for (int i = 0; i < HI; ++i) {
                const unsigned v = (last && !htail[i]) ? OOB : hoff[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base + i * 16 * BK), 16, (int)v, chunk * BK * 2, 0, 0);
            }
        };
        // group g of a chunk = taps {2g, 2g+1} for g < 4, {8} for g == 4.  ALWAYS 2 * WI pieces (the missing tap of group
        // 4 is loaded as zeros through out-of-range offsets) so that the loader's vmcnt counts are constants.
        auto issue_w = [&](int slot, int chunk, int g) {
            const int t0 = 2 * g, nt = (g == 4) ? 1 : 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int soff = k < nt ? ((t0 + k) * d.Cpad + chunk * BK) * 2 : 0;
                unsigned short* base = sWt(slot, k);
#pragma unroll
                for (int j = 0; j < WI; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(base + j * 16 * BK), 16,
                                                             (int)(k < nt ? woff[j] : OOB), soff, 0, 0);
            }
        };
        // Three weight slots: the group of step s+2 is issued right after barrier s (its slot was read in step s-1), so
        // it has a whole MFMA step to land and its issue time overlaps the MFMA waves' step s.  Issue order:
        // halo(0) w(0) w(1) | iteration s: w(s+2), then halo(chunk+1) when s is the first step of a chunk.
        // Loads younger than w(s) at the top of iteration s: w(s+1) and at most one halo tile (issued right after w(s)
        // or right after w(s+1)): 12 or 33 pieces may stay in flight (vmcnt retires in order).
        static_assert(2 * WI == 12 || 2 * WI == 6, "wait counts below");
        issue_halo(0);
        issue_w(0, 0, 0);
        if (S > 1) issue_w(1, 0, 1);
        int g2 = 2, c2 = 0, islot = 2;          // (chunk, group, slot) of step s + 2
        bool halo_m1 = false, halo_m2 = false;   // iterations s-1 / s-2 issued a halo tile
        for (int s = 0; s < S; ++s) {
            if (s + 1 >= S) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (halo_m1 || halo_m2) {
                if constexpr (WI == 6) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            } else {
                if constexpr (WI == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            halo_m2 = halo_m1;
            halo_m1 = false;
            if (s + 2 < S) {
                issue_w(islot, c2, g2);
                if (++g2 == 5) { g2 = 0; ++c2; }
                islot = islot == 2 ? 0 : islot + 1;
            }
            if (g == 0 && chunk + 1 < a.nchunk) { issue_halo(chunk + 1); halo_m1 = true; }
            if (++g == 5) { g = 0; ++chunk; }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            __builtin_amdgcn_s_barrier();
            const unsigned short* hbase = sH(chunk & 1);
            compute_tap(hbase, sWt(slot, 0), 2 * g);
            if (g < 4) compute_tap(hbase, sWt(slot, 1), 2 * g + 1);
            if (++g == 5) { g = 0; ++chunk; }
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    // ---- epilogue (as conv3x3_kernel) ----
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (loader) break;          // wave-uniform
        const int y = y0 + wave * MI + mi, x = x0 + fr;
        const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + ni * 16 + 4 * fk;
            if (q >= d.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
            if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
            if (a.bias) {
                const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b[r];
            }
            if (a.scale) {
                const f32x4 sc = *(const f32x4*)(a.scale + q);
                const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
            }
            unsigned short* out = a.dst + pixel * d.dst_pitch + q;
            if (d.accumulate) {
                const u32x2 old = *(const u32x2*)out;
                v[0] += pair_lo(old[0]);
                v[1] += pair_hi(old[0]);
                v[2] += pair_lo(old[1]);
                v[3] += pair_hi(old[1]);
            }
            if (d.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2 pk;
            pk[0] = pack2(v[0], v[1]);
            pk[1] = pack2(v[2], v[3]);
            *(u32x2*)out = pk;
        }
    }
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)smem;                      // [5][2][BQ]
        __syncthreads();
        for (int i = tid; i < 5 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0 && !loader) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, 5, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// ------------------------------------------------------------------------------------------------
// Resident-weight variant for the high-resolution, few-channel layers (d1.1, d2.0, u1.conv.*):
// there the whole packed weight matrix of a channel tile (<= 110 KB) fits in LDS next to two halo
// buffers, K is short (18-27 steps) and the per-workgroup prologue / epilogue of the kernel above
// costs as much as its main loop.  One PERSISTENT workgroup per CU loads the weights once, then
// walks over patches: per (patch, chunk) it waits for ONE halo tile (the next one is already in
// flight), runs all 9 taps with no barrier in between, and keeps the BatchNorm statistics in
// registers across patches (one set of atomics per workgroup at the very end).
// NW = 4: one wave per SIMD, 4 patch rows per wave; NW = 8: two waves per SIMD, 2 patch rows per wave (same LDS
// image, 60 % more LDS reads per MFMA, but the serial phases of one wave overlap with its SIMD partner's).
template <int BQ, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_resw_kernel(const C3Args a, int npatch) {
    constexpr int BK = 32, HW_ = 18, HROWS = HW_ * HW_, HPAD = 384, NI = BQ / 16, MI = 16 / NW, HL = 24 / NW;
    constexpr int HALO_E = HPAD * BK, WT_E = BQ * BK;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];   // [2 halo][nchunk*9 weight tiles]
    auto sH = [&](int b) -> unsigned short* { return dsm + b * HALO_E; };
    auto sWt = [&](int chunk, int tap) -> unsigned short* { return dsm + 2 * HALO_E + (chunk * 9 + tap) * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int q0 = blockIdx.y * BQ;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);
    // A last chunk with at most 16 real channels (Cin = 48: k-groups 2, 3 would be zero padding) is staged PAIRED: the
    // upper half of halo row r holds the 16 channels of pixel r+1, and the upper half of the tap (ty, 0) weight tile the
    // 16 channels of tap (ty, 1).  One MFMA then multiplies taps (ty, 0) and (ty, 1) together and the tap loop skips
    // tx = 1: 9 -> 6 K-blocks for that chunk (18 -> 15 per patch at Cin = 48) with unchanged LDS read addresses -- the
    // pairing lives entirely in the per-lane SOURCE addresses of the fills.  (Pairing arbitrary taps in the read
    // addresses instead saves one more block but costs ~35 per-lane address registers: 256 VGPRs + scratch, slower.)
    const bool pair_last = (d.Cin & 31) != 0 && (d.Cin & 31) <= 16 && !a.nopair;

    // ---- weights: all (chunk, tap) tiles of this channel tile, once ----
    {
        const int ntile = a.nchunk * 9;
        constexpr int PIECES = BQ * 4;                        // 16-B pieces per tile
        for (int base = 0; base < ntile * PIECES; base += 64 * NW) {   // uniform trip count; wave-linear 1 KiB pieces
            const int p = base + tid;
            const int tile = p / PIECES, r = p - tile * PIECES;
            const int row = r >> 2, lc = swz32(row, r & 3);
            const int chunk = tile / 9, tap = tile - chunk * 9;
            const bool ok = tile < ntile && q0 + row < d.Cout;
            // paired last chunk (see pair_last): k-groups 2, 3 of the tap (ty, 0) tile hold tap (ty, 1)'s 16 channels
            const bool pr = pair_last && chunk == a.nchunk - 1 && tap % 3 == 0 && lc >= 2;
            const unsigned v = ok ? (unsigned)((((q0 + row) * 9 + tap + (pr ? 1 : 0)) * d.Cpad + chunk * BK + (pr ? lc - 2 : lc) * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dsm + 2 * HALO_E + (base + wave * 64) * 8), 16, (int)v, 0, 0, 0);
        }
    }
    // ---- halo roles (patch independent part) ----
    int hy_[HL], hx_[HL], lc_[HL];
    bool htail[HL], htail_p[HL];
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const int hr = (i * NW + wave) * 16 + (lane >> 2);
        lc_[i] = swz32(hr, lane & 3);
        hy_[i] = hr < HROWS ? hr / HW_ : -100000;
        hx_[i] = hr % HW_;
        htail[i] = tail_c0 + lc_[i] * 8 < d.Cin;
        htail_p[i] = tail_c0 + (lc_[i] - 2) * 8 < d.Cin;      // paired last chunk, upper k-groups
    }
    const bool has_tail = d.Cpad != d.Cin;
    const int split_c = d.src_split_c > 0 ? d.src_split_c : 0x7fffffff;
    const int split_adj = d.src_split_off - d.src_split_c;           // elements
    const int dsplit_c = d.dst_split_c > 0 ? d.dst_split_c : 0x7fffffff;
    const int dsplit_adj = d.dst_split_off - d.dst_split_c;
    auto patch_origin = [&](int patch, int& n, int& y0, int& x0) {
        if (a.rev) patch = npatch - 1 - patch;
        const int px_t = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int py_t = t2 % a.tiles_y;
        n = t2 / a.tiles_y; y0 = py_t * 16; x0 = px_t * 16;
    };
    auto issue_halo = [&](int buf, int patch, int chunk) {
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        const bool last = has_tail && chunk == a.nchunk - 1;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            // paired last chunk: the upper k-groups fetch the NEXT pixel's first 16 channels of the chunk
            const bool pr = pair_last && chunk == a.nchunk - 1 && lc_[i] >= 2;
            const int lcs = pr ? lc_[i] - 2 : lc_[i];
            const int y = y0 - 1 + hy_[i], x = x0 - 1 + hx_[i] + (pr ? 1 : 0);
            const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W && !(last && !(pr ? htail_p[i] : htail[i])) &&
                            !(pr && hx_[i] + 1 >= HW_);
            // two-plane source (aau.h): this lane's 8 channels of the chunk may live in the second plane
            const int sadj = (chunk * BK + lcs * 8 >= split_c) ? split_adj : 0;
            const unsigned v = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lcs * 8 + sadj) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sH(buf) + (i * NW + wave) * 16 * BK), 16, (int)v,
                                                     chunk * BK * 2, 0, 0);
        }
    };

    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;

    const int first = blockIdx.x, stride = gridDim.x;
    const bool full_tiles = q0 + BQ <= d.Cout;   // every lane issues all NI*MI stores of a patch
    // 16-byte epilogue stores (epi_pair_wide) when the destination allows them
    const bool wide = ((uintptr_t)a.dst & 15) == 0 && d.dst_pitch % 8 == 0 && !a.nowide &&
                      (d.dst_split_c <= 0 || (d.dst_split_c % 8 == 0 && d.dst_split_off % 8 == 0));
    int t = 0;
    if (first < npatch) issue_halo(0, first, 0);
    for (int patch = first; patch < npatch; patch += stride) {
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < a.nchunk; ++chunk, ++t) {
            // halo(t) (and, the first time, the weights) must have landed.  At the first chunk of a later
            // patch the only younger operations are the previous patch's NI*MI output stores per lane
            // (vmcnt retires in issue order), which may stay in flight.
            if (t > 0 && chunk == 0 && full_tiles && wide) {          // half as many (16-byte) stores per lane
                if constexpr (NI * MI == 6) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else if (t > 0 && chunk == 0 && full_tiles) {
                if constexpr (NI * MI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            {   // prefetch the next (patch, chunk) tile into the other buffer
                int np = patch, nc = chunk + 1;
                if (nc == a.nchunk) { nc = 0; np += stride; }
                if (np < npatch) issue_halo((t + 1) & 1, np, nc);
            }
            const unsigned short* hbase = sH(t & 1);
            const bool paired = pair_last && chunk == a.nchunk - 1;     // wave-uniform
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3, tx = tap % 3;
                if (tx == 1 && paired) continue;                        // multiplied together with tap (ty, 0)
                const unsigned short* wbase = sWt(chunk, tap);
                bf16x8 wf[NI], af[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = ni * 16 + fr;
                    wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int hr = (wave * MI + mi + ty) * HW_ + fr + tx;
                    af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
                }
#ifdef AAU_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#ifdef AAU_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
        }
        // ---- per-patch epilogue ----
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        if (wide) {
            static_assert(MI % 2 == 0, "pixel rows are stored in pairs");
#pragma unroll
            for (int mp = 0; mp < MI; mp += 2) {
                const int yl = y0 + wave * MI + mp + (fk & 1);        // the row this lane stores after the swap
                const int64_t pl = ((int64_t)n * d.H + yl) * d.W + x0 + fr;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int qw = q0 + ni * 16 + 8 * (fk >> 1);
                    epi_pair_wide(a, d, q0 + ni * 16 + 4 * fk, qw, acc[ni][mp], acc[ni][mp + 1], want_stats, s1[ni], s2[ni],
                                  a.dst + pl * d.dst_pitch + qw + (qw >= dsplit_c ? dsplit_adj : 0), true);
                }
            }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int y = y0 + wave * MI + mi, x = x0 + fr;
            const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int q = q0 + ni * 16 + 4 * fk;
                if (q >= d.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
                if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += b[r];
                }
                if (a.scale) {
                    const f32x4 sc = *(const f32x4*)(a.scale + q);
                    const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
                }
                if (d.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 pk;
                pk[0] = pack2(v[0], v[1]);
                pk[1] = pack2(v[2], v[3]);
                *(u32x2*)(a.dst + pixel * d.dst_pitch + q + (q >= dsplit_c ? dsplit_adj : 0)) = pk;
            }
        }
    }
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)dsm;                      // [NW][2][BQ]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < NW * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, NW, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// Two patch streams per workgroup (8 waves = 2 groups of 4) SHARING one resident weight image: 2 waves per SIMD for the
// layers whose weights leave room for four halo buffers (Cin <= 64, 48 output channels: 54 KiB + 96 KiB).  At one
// wave per SIMD the patch time of the kernel above is a serial sum (MFMA issue, epilogue VALU, LDS reads, stores:
// each 13-22 % by ablation); the second group fills those gaps.  Both groups run the same number of steps (host
// guarantees npatch % (2 * gridDim.x) == 0), so the workgroup-wide barriers line up.
__global__ __launch_bounds__(512) void conv3x3_resw2_kernel(const C3Args a, int npatch) {
    constexpr int BQ = 48;
    constexpr int BK = 32, HW_ = 18, HROWS = HW_ * HW_, HPAD = 384, NI = BQ / 16, MI = 4, HL = 6;
    constexpr int HALO_E = HPAD * BK, WT_E = BQ * BK;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];   // [2 groups][2 halo][nchunk*9 weight tiles]

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;   // group, wave within the group
    auto sH = [&](int b) -> unsigned short* { return dsm + (grp * 2 + b) * HALO_E; };
    auto sWt = [&](int chunk, int tap) -> unsigned short* { return dsm + 4 * HALO_E + (chunk * 9 + tap) * WT_E; };
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int q0 = blockIdx.y * BQ;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);
    // A last chunk with at most 16 real channels (Cin = 48: k-groups 2, 3 would be zero padding) is staged PAIRED: the
    // upper half of halo row r holds the 16 channels of pixel r+1, and the upper half of the tap (ty, 0) weight tile the
    // 16 channels of tap (ty, 1).  One MFMA then multiplies taps (ty, 0) and (ty, 1) together and the tap loop skips
    // tx = 1: 9 -> 6 K-blocks for that chunk (18 -> 15 per patch at Cin = 48) with unchanged LDS read addresses -- the
    // pairing lives entirely in the per-lane SOURCE addresses of the fills.  (Pairing arbitrary taps in the read
    // addresses instead saves one more block but costs ~35 per-lane address registers: 256 VGPRs + scratch, slower.)
    const bool pair_last = (d.Cin & 31) != 0 && (d.Cin & 31) <= 16 && !a.nopair;

    // ---- weights: all (chunk, tap) tiles of this channel tile, once ----
    {
        const int ntile = a.nchunk * 9;
        constexpr int PIECES = BQ * 4;                        // 16-B pieces per tile
        for (int base = 0; base < ntile * PIECES; base += 512) {   // uniform trip count; wave-linear 1 KiB pieces
            const int p = base + tid;
            const int tile = p / PIECES, r = p - tile * PIECES;
            const int row = r >> 2, lc = swz32(row, r & 3);
            const int chunk = tile / 9, tap = tile - chunk * 9;
            const bool ok = tile < ntile && q0 + row < d.Cout;
            // paired last chunk (see pair_last): k-groups 2, 3 of the tap (ty, 0) tile hold tap (ty, 1)'s 16 channels
            const bool pr = pair_last && chunk == a.nchunk - 1 && tap % 3 == 0 && lc >= 2;
            const unsigned v = ok ? (unsigned)((((q0 + row) * 9 + tap + (pr ? 1 : 0)) * d.Cpad + chunk * BK + (pr ? lc - 2 : lc) * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dsm + 4 * HALO_E + (base + wave8 * 64) * 8), 16, (int)v, 0, 0, 0);
        }
    }
    // ---- halo roles (patch independent part) ----
    int hy_[HL], hx_[HL], lc_[HL];
    bool htail[HL], htail_p[HL];
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const int hr = (i * 4 + wave) * 16 + (lane >> 2);
        lc_[i] = swz32(hr, lane & 3);
        hy_[i] = hr < HROWS ? hr / HW_ : -100000;
        hx_[i] = hr % HW_;
        htail[i] = tail_c0 + lc_[i] * 8 < d.Cin;
        htail_p[i] = tail_c0 + (lc_[i] - 2) * 8 < d.Cin;      // paired last chunk, upper k-groups
    }
    const bool has_tail = d.Cpad != d.Cin;
    auto patch_origin = [&](int patch, int& n, int& y0, int& x0) {
        if (a.rev) patch = npatch - 1 - patch;
        const int px_t = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int py_t = t2 % a.tiles_y;
        n = t2 / a.tiles_y; y0 = py_t * 16; x0 = px_t * 16;
    };
    auto issue_halo = [&](int buf, int patch, int chunk) {
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        const bool last = has_tail && chunk == a.nchunk - 1;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            // paired last chunk: the upper k-groups fetch the NEXT pixel's first 16 channels of the chunk
            const bool pr = pair_last && chunk == a.nchunk - 1 && lc_[i] >= 2;
            const int lcs = pr ? lc_[i] - 2 : lc_[i];
            const int y = y0 - 1 + hy_[i], x = x0 - 1 + hx_[i] + (pr ? 1 : 0);
            const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W && !(last && !(pr ? htail_p[i] : htail[i])) &&
                            !(pr && hx_[i] + 1 >= HW_);
            const unsigned v = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lcs * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sH(buf) + (i * 4 + wave) * 16 * BK), 16, (int)v,
                                                     chunk * BK * 2, 0, 0);
        }
    };

    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;

    const int first = blockIdx.x * 2 + grp, stride = gridDim.x * 2;
    const bool full_tiles = q0 + BQ <= d.Cout;   // every lane issues all NI*MI stores of a patch
    // 16-byte epilogue stores (epi_pair_wide) when the destination allows them
    const bool wide = ((uintptr_t)a.dst & 15) == 0 && d.dst_pitch % 8 == 0 && !a.nowide;
    int t = 0;
    if (first < npatch) issue_halo(0, first, 0);
    for (int patch = first; patch < npatch; patch += stride) {
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < a.nchunk; ++chunk, ++t) {
            // halo(t) (and, the first time, the weights) must have landed.  At the first chunk of a later
            // patch the only younger operations are the previous patch's NI*MI output stores per lane
            // (vmcnt retires in issue order), which may stay in flight.
            if (t > 0 && chunk == 0 && full_tiles && wide) {          // half as many (16-byte) stores per lane
                if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else if (t > 0 && chunk == 0 && full_tiles) {
                if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            {   // prefetch the next (patch, chunk) tile into the other buffer
                int np = patch, nc = chunk + 1;
                if (nc == a.nchunk) { nc = 0; np += stride; }
                if (np < npatch && !(a.rev & 8)) issue_halo((t + 1) & 1, np, nc);
            }
            const unsigned short* hbase = sH(t & 1);
            const bool paired = pair_last && chunk == a.nchunk - 1;     // wave-uniform
#ifdef AAU_RESW2_TAPLOOP    // ablation build (scripts/gpu_resw2_ab.sh): one tap at a time, 63 reads per chunk
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3, tx = tap % 3;
                if (tx == 1 && paired) continue;
                const unsigned short* wbase = sWt(chunk, tap);
                bf16x8 wf[NI], af[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = ni * 16 + fr;
                    wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int hr = (wave * MI + mi + ty) * HW_ + fr + tx;
                    af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        if (!(a.rev & 4)) acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
        }
#else
            // Column step with vertical reuse: for one horizontal tap offset tx the wave's MI output rows and the three
            // vertical taps touch MI + 2 halo rows; read those once and let each feed up to three MFMA rows.  45 wave-wide
            // ds_read_b128 per chunk (3 x (6 activation + 9 weight)) instead of 63 (9 taps x (4 + 3)).
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                if (tx == 1 && paired) continue;                        // multiplied together with the tx = 0 column
                bf16x8 ar[MI + 2];
#pragma unroll
                for (int r = 0; r < MI + 2; ++r) {
                    const int hr = (wave * MI + r) * HW_ + fr + tx;
                    ar[r] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
                }
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const unsigned short* wbase = sWt(chunk, ty * 3 + tx);
                    bf16x8 wf[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int row = ni * 16 + fr;
                        wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            if (!(a.rev & 4)) acc[ni][mi] = AAU_MFMA16(wf[ni], ar[mi + ty], acc[ni][mi], 0, 0, 0);   // (rev & 2/4/8: timing-only ablations, AAU_RESW_ABL)
                }
            }
        }
#endif
        // ---- per-patch epilogue ----
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        if (wide) {
#pragma unroll
            for (int mp = 0; mp < MI; mp += 2) {
                const int yl = y0 + wave * MI + mp + (fk & 1);        // the row this lane stores after the swap
                const int64_t pl = ((int64_t)n * d.H + yl) * d.W + x0 + fr;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int qw = q0 + ni * 16 + 8 * (fk >> 1);
                    epi_pair_wide(a, d, q0 + ni * 16 + 4 * fk, qw, acc[ni][mp], acc[ni][mp + 1], want_stats, s1[ni], s2[ni],
                                  a.dst + pl * d.dst_pitch + qw, !(a.rev & 2));
                }
            }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int y = y0 + wave * MI + mi, x = x0 + fr;
            const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int q = q0 + ni * 16 + 4 * fk;
                if (q >= d.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
                if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(a.bias + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += b[r];
                }
                if (a.scale) {
                    const f32x4 sc = *(const f32x4*)(a.scale + q);
                    const f32x4 sh = *(const f32x4*)(a.shift + q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
                }
                if (d.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 pk;
                pk[0] = pack2(v[0], v[1]);
                pk[1] = pack2(v[2], v[3]);
                if (!(a.rev & 2)) *(u32x2*)(a.dst + pixel * d.dst_pitch + q) = pk;
            }
        }
    }
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)dsm;                      // [8][2][BQ]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < 8 * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave8 * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, 8, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// ------------------------------------------------------------------------------------------------
// The same persistent resident-weight structure for the 1x1 convolutions of the decoder at high resolution
// (ConvTranspose2d(2,2) forward as a GEMM with the pixel-shuffle store, the attention-gate Wg / Wx convs and their
// accumulating input gradients): K is 1-3 steps there, and the generic implicit-GEMM kernel spends its time in
// per-tile prologues, three exposed load waits and barriers (a no-store ablation of it still took 73 % of the
// time).  Here the weights ([BQ][Cpad], <= 36 KB) stay in LDS, the activation tile of the next 256 pixels is in
// flight while the current one is multiplied, and there is one barrier per 32-channel chunk.
// NBUF pixel tiles of one 32-channel chunk each form a ring; the fill of tile t + NBUF - 1 is issued while tile t is
// multiplied.  Round 2 shipped NBUF = 2, i.e. ONE 16-KB tile of look-ahead per CU: at ~1.1 us from issue to landing that
// caps a CU at ~15 GB/s (3.7 TB/s chip-wide; measured 2.2-3.4 TB/s on these byte-bound layers).  The weights of these
// layers are small (<= 48 KB), so the ring can hold 7 tiles = 96 KB in flight per CU.
// (An LDS-staged, fully coalesced epilogue -- whole 16-pixel / 32-output-pixel rows per wave-wide store -- measured 0-15 %
// SLOWER than the per-lane 16-byte stores on every layer and was removed: profiles/NOTES.md.)
template <int BQ, int NW, int NBUF, int ABL = 0>
__global__ __launch_bounds__(64 * NW) void conv1x1_resw_kernel(const C3Args a, int npatch) {
    constexpr int BK = 32, HW_ = 16, HROWS = HW_ * HW_, HPAD = 256, NI = BQ / 16, MI = 16 / NW;
    constexpr int NWF = NW;                            // waves that fill
    constexpr int HL = 16 / NWF;
    constexpr int HALO_E = HPAD * BK, WT_E = BQ * BK;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];   // [NBUF pixel tiles][nchunk weight tiles]
    constexpr int PD = NBUF - 1;                     // tiles in flight
    auto sH = [&](int b) -> unsigned short* { return dsm + b * HALO_E; };
    auto sWt = [&](int chunk) -> unsigned short* { return dsm + NBUF * HALO_E + chunk * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int q0 = blockIdx.y * BQ;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // ---- weights: all (chunk, tap) tiles of this channel tile, once ----
    {
        const int ntile = a.nchunk;
        constexpr int PIECES = BQ * 4;                        // 16-B pieces per tile
        for (int base = 0; base < ntile * PIECES; base += 64 * NW) {   // uniform trip count; wave-linear 1 KiB pieces
            const int p = base + tid;
            const int tile = p / PIECES, r = p - tile * PIECES;
            const int row = r >> 2, lc = swz32(row, r & 3);
            const int chunk = tile;
            const bool ok = tile < ntile && q0 + row < d.Cout;
            const unsigned v = ok ? (unsigned)(((q0 + row) * d.Cpad + chunk * BK + lc * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dsm + NBUF * HALO_E + (base + wave * 64) * 8), 16, (int)v, 0, 0, 0);
        }
    }
    // ---- halo roles (patch independent part) ----
    int hy_[HL], hx_[HL], lc_[HL];
    bool htail[HL];
    const int tail_c0 = (a.nchunk - 1) * BK;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const int hr = (i * NWF + wave) * 16 + (lane >> 2);
        lc_[i] = swz32(hr, lane & 3);
        hy_[i] = hr < HROWS ? hr / HW_ : -100000;
        hx_[i] = hr % HW_;
        htail[i] = tail_c0 + lc_[i] * 8 < d.Cin;
    }
    const bool has_tail = d.Cpad != d.Cin;
    auto patch_origin = [&](int patch, int& n, int& y0, int& x0) {
        if (a.rev) patch = npatch - 1 - patch;
        const int px_t = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int py_t = t2 % a.tiles_y;
        n = t2 / a.tiles_y; y0 = py_t * 16; x0 = px_t * 16;
    };
    auto issue_halo = [&](int buf, int patch, int chunk) {
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        const bool last = has_tail && chunk == a.nchunk - 1;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            const int y = y0 + hy_[i], x = x0 + hx_[i];
            const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W && !(last && !htail[i]);
            const unsigned v = (ok && !(ABL & 2)) ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + lc_[i] * 8) * 2) : OOB;   // ABL: timing ablations (-DAAU_C3S_ABLATE builds)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sH(buf) + (i * NWF + wave) * 16 * BK), 16, (int)v,
                                                     chunk * BK * 2, 0, 0);
        }
    };

    // per-channel epilogue vectors (bias, folded-BN scale / shift) staged in LDS behind the weight tiles: a global load in
    // the epilogue would wait for every fill in flight (vmcnt retires in order)
    // (behind the ROUNDED weight area: the staging loop above writes whole 8-KiB rounds, zeros past the last tile)
    float* par = (float*)((unsigned char*)dsm + NBUF * HALO_E * 2 + ((size_t)a.nchunk * BQ * 64 + 8191) / 8192 * 8192);   // [3][BQ]
    if (a.bias || a.scale) {
        const int Co_ = d.shuffle2x2 ? d.Cout >> 2 : d.Cout;
        for (int i = tid; i < BQ; i += 64 * NW) {
            const int q = q0 + i;
            const int qv = q < d.Cout ? (d.shuffle2x2 ? q % Co_ : q) : 0;
            par[i] = a.bias ? a.bias[qv] : 0.f;
            par[BQ + i] = a.scale ? a.scale[qv] : 1.f;
            par[2 * BQ + i] = a.scale ? a.shift[qv] : 0.f;
        }
        __syncthreads();
    }
    constexpr bool STATS = true;
    constexpr int NS = STATS ? NI : 1;
    const bool want_stats = STATS && a.stats != nullptr;
    float s1[NS][4], s2[NS][4];
#pragma unroll
    for (int ni = 0; ni < NS; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;

    const int first = blockIdx.x, stride = gridDim.x;
    const bool full_tiles = q0 + BQ <= d.Cout;   // every lane issues all NI*MI stores of a patch
    // 16-byte epilogue stores (common.h: swap_pair8) when the destination allows them
    const bool wide = ((uintptr_t)a.dst & 15) == 0 && d.dst_pitch % 8 == 0 && (!d.shuffle2x2 || (d.Cout >> 2) % 8 == 0) && !a.nowide;
    int t = 0;
#ifdef AAU_PW_STAMP
    // diagnostic build only (scripts/probes/pw_stamp.py): s_memtime stamps of a tile step; a.shift is the debug buffer
#define PW_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory")
    unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_epi = 0, st_steps = 0, st_begin, st_a, st_b;
    PW_STAMP(st_begin);
#endif
    // fill cursor: (patch, chunk) of the next tile to issue; the ring runs PD tiles ahead of the multiply
    int fp = first, fc = 0, fb = 0;
    const bool filler = wave < NWF;
    auto issue_next = [&]() {
        if (filler) {
            if (fp < npatch) issue_halo(fb, fp, fc);
            else {                               // past the end: keep every filling wave's instruction count the same
#pragma unroll
                for (int i = 0; i < HL; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sH(fb) + (i * NWF + wave) * 16 * BK), 16, (int)OOB, 0, 0, 0);
            }
        }
        if (++fc == a.nchunk) { fc = 0; fp += stride; }
        if (++fb == NBUF) fb = 0;
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue_next();
    int cb = 0;                                  // ring slot of tile t
    for (int patch = first; patch < npatch; patch += stride) {
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < a.nchunk; ++chunk, ++t) {
            // halo(t) (and, the first time, the weights) must have landed.  At the first chunk of a later
            // patch the only younger operations are the previous patch's NI*MI output stores per lane
            // (vmcnt retires in issue order), which may stay in flight.
            // tile t must have landed: the younger operations are the fills of tiles t+1 .. t+PD-1 (HL instructions per
            // wave each) and output stores issued between them; waiting down to the fills alone is exact when no store
            // is younger than tile t and at worst also retires a few fills that were issued PD-2 steps ago
#ifdef AAU_PW_STAMP
            PW_STAMP(st_a);
#endif
            if constexpr (PD == 1) {
                if (t > 0 && chunk == 0 && full_tiles && wide) {          // half as many (16-byte) stores per lane
                    if constexpr (NI * MI == 6) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                } else if (t > 0 && chunk == 0 && full_tiles) {
                    if constexpr (NI * MI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if constexpr (NI * MI == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                static_assert(PD == 1 || (PD - 1) * HL == 10 || (PD - 1) * HL == 8, "immediates below");
                if (!filler) { if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // its share of the weight tiles
                else if constexpr ((PD - 1) * HL == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
#ifdef AAU_PW_STAMP
            PW_STAMP(st_b); st_wait += st_b - st_a;
#endif
            __builtin_amdgcn_s_barrier();
#ifdef AAU_PW_STAMP
            PW_STAMP(st_a); st_bar += st_a - st_b;
#endif
            issue_next();                        // tile t + PD, into the slot tile t - 1 has just left
#ifdef AAU_PW_STAMP
            PW_STAMP(st_b); st_issue += st_b - st_a; ++st_steps;
#endif
            const unsigned short* hbase = sH(cb);
            if (++cb == NBUF) cb = 0;
            {
                constexpr int ty = 0, tx = 0;
                const unsigned short* wbase = sWt(chunk);
                bf16x8 wf[NI], af[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = ni * 16 + fr;
                    wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int hr = (wave * MI + mi + ty) * HW_ + fr + tx;
                    af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
                }
#ifdef AAU_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#ifdef AAU_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#ifdef AAU_PW_STAMP
            asm volatile("" :: "v"(acc[0][0]));
            PW_STAMP(st_a); st_mfma += st_a - st_b;
#endif
        }
        // ---- per-patch epilogue ----
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
#ifdef AAU_PW_STAMP
        PW_STAMP(st_a);
#endif
        if (wide) {
            static_assert(MI % 2 == 0, "pixel rows are stored in pairs");
            const int Co = d.Cout >> 2;
#pragma unroll
            for (int mp = 0; mp < MI; mp += 2) {
                const int yA = y0 + wave * MI + mp, x = x0 + fr;
                const int yl = yA + (fk & 1);                         // the row this lane stores after the swap
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int q = q0 + ni * 16 + 4 * fk;
                    float va[4], vb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { va[r] = acc[ni][mp][r]; vb[r] = acc[ni][mp + 1][r]; }
                    if (q < d.Cout) {
                        if constexpr (STATS) {
                            if (want_stats) { epi_stats(a, 0, q, va, s1[ni], s2[ni]); epi_stats(a, 0, q, vb, s1[ni], s2[ni]); }
                        }
                        const int ql = ni * 16 + 4 * fk;             // channel inside this workgroup's tile
                        if (a.bias) {
                            const f32x4 b = *(const f32x4*)(par + ql);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { va[r] += b[r]; vb[r] += b[r]; }
                        }
                        if (a.scale) {
                            const f32x4 sc = *(const f32x4*)(par + BQ + ql);
                            const f32x4 sh = *(const f32x4*)(par + 2 * BQ + ql);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { va[r] = va[r] * sc[r] + sh[r]; vb[r] = vb[r] * sc[r] + sh[r]; }
                        }
                    }
                    float w[8];
                    swap_pair8(va, vb, w);                            // every lane takes part
                    const int qw = q0 + ni * 16 + 8 * (fk >> 1);
                    if (qw >= d.Cout) continue;
                    unsigned short* out;
                    if (d.shuffle2x2) {
                        const int pos = qw / Co;
                        const int64_t op = ((int64_t)n * (2 * d.H) + (2 * yl + (pos >> 1))) * (2 * d.W) + (2 * x + (pos & 1));
                        out = a.dst + op * d.dst_pitch + (qw - pos * Co);
                    } else {
                        out = a.dst + (((int64_t)n * d.H + yl) * d.W + x) * d.dst_pitch + qw;
                    }
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
                    if (!(ABL & 4)) *(u32x4*)out = pack8(w);
                    else if (w[0] == 1.2345f) *(u32x4*)out = pack8(w);
                }
            }
#ifdef AAU_PW_STAMP
            PW_STAMP(st_b); st_epi += st_b - st_a;
#endif
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int y = y0 + wave * MI + mi, x = x0 + fr;
            const int64_t pixel = ((int64_t)n * d.H + y) * d.W + x;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int q = q0 + ni * 16 + 4 * fk;
                if (q >= d.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
                if constexpr (STATS) {
                    if (want_stats) epi_stats(a, pixel, q, v, s1[ni], s2[ni]);
                }
                // per-channel vectors are indexed by the real output channel (co for the pixel-shuffle store)
                const int Co = d.Cout >> 2;
                const int qv = d.shuffle2x2 ? q % Co : q;
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(a.bias + qv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += b[r];
                }
                if (a.scale) {
                    const f32x4 sc = *(const f32x4*)(a.scale + qv);
                    const f32x4 sh = *(const f32x4*)(a.shift + qv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
                }
                unsigned short* out;
                if (d.shuffle2x2) {   // ConvTranspose2d(2,2): row q = (pos, co) goes to sub-pixel pos of the 2x grid
                    const int pos = q / Co;
                    const int64_t op = ((int64_t)n * (2 * d.H) + (2 * y + (pos >> 1))) * (2 * d.W) + (2 * x + (pos & 1));
                    out = a.dst + op * d.dst_pitch + qv;
                } else {
                    out = a.dst + pixel * d.dst_pitch + q;
                }
                if (d.accumulate) {
                    const u32x2 old = *(const u32x2*)out;
                    v[0] += pair_lo(old[0]);
                    v[1] += pair_hi(old[0]);
                    v[2] += pair_lo(old[1]);
                    v[3] += pair_hi(old[1]);
                }
                if (d.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 pk;
                pk[0] = pack2(v[0], v[1]);
                pk[1] = pack2(v[2], v[3]);
                *(u32x2*)out = pk;
            }
        }
    }
#ifdef AAU_PW_STAMP
    {
        unsigned long long st_end, rt;
        PW_STAMP(st_end);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) : : "memory");
        if (lane == 0 && a.shift) {
            unsigned long long* o = (unsigned long long*)a.shift + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
            o[0] = st_wait; o[1] = st_bar; o[2] = st_issue; o[3] = st_mfma; o[4] = st_epi; o[5] = st_steps; o[6] = st_end - st_begin; o[7] = rt;
        }
    }
#endif
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)dsm;                      // [NW][2][BQ]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < NW * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NS; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, NW, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}



// ------------------------------------------------------------------------------------------------
// Round 4: the same kernel with its address arithmetic taken out of the loop.  s_memtime stamps of conv1x1_resw_kernel
// (profiles/r04_pw_stamp_before.txt; u1.up forward, 92 us) showed that the tile it waits for has always landed (3 % of
// the loop) and that the time goes into ISSUING things: 21 % in the two LDS-DMA instructions of a step (patch decode with
// divisions, image-border predicates and 64-bit offsets per piece: 370 cycles per instruction), 37 % in the epilogue
// (a 64-bit destination address with a division per 16-byte store), 14 % at the barrier where the first four waves wait
// for the last four; for the accumulating gate gradients 62 % in the epilogue, whose read-modify-write loads queue behind
// every fill in flight (vmcnt retires in order: each patch drained the ring).  Here
//   * a piece's per-lane offset is a kernel constant (H, W are multiples of 16: a patch never meets the image border)
//     and the patch / chunk part travels in the SCALAR offset of the LDS-DMA: a step issues its fills with no vector
//     arithmetic;
//   * the stores are buffer stores: per-lane offset constant per (row pair, channel group), patch part scalar;
//     out-of-range channel groups carry an out-of-range offset, so every lane issues the same number of stores and the
//     counted waits hold;
//   * an accumulating destination is loaded at the START of the patch (asm loads hipcc does not count: beside LDS-DMA it
//     would wait vmcnt(0) for them), lands behind the patch's own MFMA steps and is waited for with a counted vmcnt;
//   * the step wait counts the loads and the previous patch's stores as young operations instead of waiting them out
//     (the old count retired three tiles of the ring at every patch start).
// Wide (16-byte) stores only, NBUF = 7; everything else stays with conv1x1_resw_kernel.
template <int BQ, int NW, bool ACC>
__global__ __launch_bounds__(64 * NW) void conv1x1_rs_kernel(const C3Args a, int npatch) {
    constexpr int NBUF = 7;
    constexpr int BK = 32, HW_ = 16, NI = BQ / 16, MI = 16 / NW;
    constexpr int HL = 16 / NW;                      // LDS-DMA instructions per wave and tile
    constexpr int HALO_E = 256 * BK, WT_E = BQ * BK;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PD = NBUF - 1;                     // tiles in flight
    constexpr int L = NI * MI / 2;                   // 16-byte stores (and read-modify-write loads) per lane and patch
    static_assert(MI == 2 && HL == 2, "eight waves: one row pair per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];   // [NBUF pixel tiles][nchunk weight tiles] | vectors
    auto sH = [&](int b) -> unsigned short* { return dsm + b * HALO_E; };
    auto sWt = [&](int chunk) -> unsigned short* { return dsm + NBUF * HALO_E + chunk * WT_E; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int q0 = blockIdx.y * BQ;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, 0x7fffffff, 0x00020000);

    // ---- weights: all chunk tiles of this channel tile, once ----
    {
        const int ntile = a.nchunk;
        constexpr int PIECES = BQ * 4;
        for (int base = 0; base < ntile * PIECES; base += 64 * NW) {
            const int p = base + tid;
            const int tile = p / PIECES, r = p - tile * PIECES;
            const int row = r >> 2, lc = swz32(row, r & 3);
            const bool ok = tile < ntile && q0 + row < d.Cout;
            const unsigned v = ok ? (unsigned)(((q0 + row) * d.Cpad + tile * BK + lc * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dsm + NBUF * HALO_E + (base + wave * 64) * 8), 16, (int)v, 0, 0, 0);
        }
    }
    // ---- fill roles: per-lane byte offset of (patch row, column, 16-byte part of the chunk) relative to the patch origin ----
    unsigned vrel[HL], vrel_tail[HL];
    const int tail_c0 = (a.nchunk - 1) * BK;
    const bool has_tail = d.Cpad != d.Cin;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const int hr = (i * NW + wave) * 16 + (lane >> 2);
        const int lc = swz32(hr, lane & 3);
        vrel[i] = (unsigned)((((hr / HW_) * d.W + (hr % HW_)) * d.src_pitch + lc * 8) * 2);
        vrel_tail[i] = (tail_c0 + lc * 8 < d.Cin) ? vrel[i] : OOB;
    }
    auto patch_origin = [&](int patch, int& n, int& y0, int& x0) {
        if (a.rev) patch = npatch - 1 - patch;
        const int px_t = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int py_t = t2 % a.tiles_y;
        n = t2 / a.tiles_y; y0 = py_t * 16; x0 = px_t * 16;
    };
    auto src_base = [&](int patch) -> unsigned {
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        return (unsigned)((((n * d.H + y0) * d.W + x0) * d.src_pitch) * 2);
    };

    // per-channel epilogue vectors (bias, folded-BN scale / shift) in LDS behind the ROUNDED weight area
    float* par = (float*)((unsigned char*)dsm + NBUF * HALO_E * 2 + ((size_t)a.nchunk * BQ * 64 + 8191) / 8192 * 8192);   // [3][BQ]
    if (a.bias || a.scale) {
        const int Co_ = d.shuffle2x2 ? d.Cout >> 2 : d.Cout;
        for (int i = tid; i < BQ; i += 64 * NW) {
            const int q = q0 + i;
            const int qv = q < d.Cout ? (d.shuffle2x2 ? q % Co_ : q) : 0;
            par[i] = a.bias ? a.bias[qv] : 0.f;
            par[BQ + i] = a.scale ? a.scale[qv] : 1.f;
            par[2 * BQ + i] = a.scale ? a.shift[qv] : 0.f;
        }
        __syncthreads();
    }
    // BNIN (aau_conv_igemm_bnin on a 1x1 / ConvTranspose forward): the source is the raw conv output z of the producing
    // ConvBNReLU; relu(z * scale + shift) is applied to each landed pixel tile in LDS by the lanes that fetched its pieces
    // (behind their vmcnt wait, in front of the step's barrier).  Table [2][192] floats, zero past Cin (the channel tail of
    // the last chunk stays relu(0) = 0).
    const bool bnin = a.in_scale != nullptr;
    float* inp = par + 3 * 96;
    if (bnin) {
        for (int i = tid; i < 192; i += 64 * NW) {
            inp[i] = i < d.Cin ? a.in_scale[i] : 0.f;
            inp[192 + i] = i < d.Cin ? a.in_shift[i] : 0.f;
        }
        __syncthreads();
    }
    int xlc[HL];                                      // 16-byte part (of the 32-channel chunk) this lane's piece i holds
#pragma unroll
    for (int i = 0; i < HL; ++i) xlc[i] = swz32((i * NW + wave) * 16 + (lane >> 2), lane & 3);
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;

    // ---- store roles: after the cross-lane swap this lane owns channels qw .. qw + 7 of the pixel (row + (fk & 1), fr) ----
    unsigned evo[NI];
    const int Co = d.Cout >> 2;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int qw = q0 + ni * 16 + 8 * (fk >> 1);
        const int row = wave * MI + (fk & 1);
        if (qw >= d.Cout) evo[ni] = OOB;
        else if (d.shuffle2x2) {
            const int pos = qw / Co;
            evo[ni] = (unsigned)((((2 * row + (pos >> 1)) * (2 * d.W) + 2 * fr + (pos & 1)) * d.dst_pitch + (qw - pos * Co)) * 2);
        } else {
            evo[ni] = (unsigned)(((row * d.W + fr) * d.dst_pitch + qw) * 2);
        }
    }
    auto dst_base = [&](int patch) -> unsigned {
        int n, y0, x0;
        patch_origin(patch, n, y0, x0);
        if (d.shuffle2x2) return (unsigned)((((n * 2 * d.H + 2 * y0) * (2 * d.W) + 2 * x0) * d.dst_pitch) * 2);
        return (unsigned)((((n * d.H + y0) * d.W + x0) * d.dst_pitch) * 2);
    };

    const int first = blockIdx.x, stride = gridDim.x;
#ifdef AAU_PW_STAMP
    // diagnostic build only (scripts/probes/pw_stamp.py): s_memtime stamps of a tile step; a.shift is the debug buffer
#define PW_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory")
    unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_epi = 0, st_steps = 0, st_begin, st_a, st_b;
    PW_STAMP(st_begin);
#endif
    // fill cursor: (patch, chunk) of the next tile to issue; the ring runs PD tiles ahead of the multiply
    int fp = first, fc = 0, fb = 0;
    unsigned fbase = fp < npatch ? src_base(fp) : 0u;
    auto issue_next = [&]() {
        const bool live = fp < npatch;                // past the end: out-of-range pieces keep the instruction count the same
        const bool last = has_tail && fc == a.nchunk - 1;
        const unsigned soff = live ? fbase + (unsigned)(fc * BK * 2) : 0u;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            const unsigned v = live ? (last ? vrel_tail[i] : vrel[i]) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sH(fb) + (i * NW + wave) * 16 * BK), 16, (int)v, (int)soff, 0, 0);
        }
        if (++fc == a.nchunk) {
            fc = 0; fp += stride;
            if (fp < npatch) fbase = src_base(fp);
        }
        if (++fb == NBUF) fb = 0;
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue_next();
    int cb = 0;                                  // ring slot of tile t
    for (int patch = first; patch < npatch; patch += stride) {
        const unsigned ebase = dst_base(patch);
        // an accumulating destination: fetched now, behind the stores of the previous patch and in front of this patch's
        // fills -- it lands while the chunks are multiplied
        u32x4 old[NI];
        if constexpr (ACC) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) old[ni] = load_b128_soff(rsD, evo[ni], ebase);
        }
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < a.nchunk; ++chunk) {
            // tile t has landed when all but the operations YOUNGER than its fills are done: the fills of tiles t+1 .. t+PD-1
            // ((PD - 1) * HL), this patch's L loads (issued less than PD steps ago: nchunk <= PD) and the previous patch's L
            // stores (issued one step before them)
#ifdef AAU_PW_STAMP
            PW_STAMP(st_a);
#endif
            if constexpr (ACC) {
                if (patch != first) asm volatile("s_waitcnt vmcnt(%0)" : : "i"((PD - 1) * HL + 2 * L) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "i"((PD - 1) * HL + L) : "memory");
            } else {
                if (patch != first) asm volatile("s_waitcnt vmcnt(%0)" : : "i"((PD - 1) * HL + L) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "i"((PD - 1) * HL) : "memory");
            }
            if (bnin) {
#pragma unroll
                for (int i = 0; i < HL; ++i) {
                    u32x4* q = (u32x4*)(sH(cb) + (i * NW + wave) * 16 * BK + lane * 8);
                    const float* t = inp + chunk * BK + xlc[i] * 8;
                    float f[8], sc[8], sh[8];
                    *(f32x4*)(sc) = *(const f32x4*)(t); *(f32x4*)(sc + 4) = *(const f32x4*)(t + 4);
                    *(f32x4*)(sh) = *(const f32x4*)(t + 192); *(f32x4*)(sh + 4) = *(const f32x4*)(t + 196);
                    unpack8(*q, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
                    *q = pack8(f);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#ifdef AAU_PW_STAMP
            PW_STAMP(st_b); st_wait += st_b - st_a;
#endif
            __builtin_amdgcn_s_barrier();
#ifdef AAU_PW_STAMP
            PW_STAMP(st_a); st_bar += st_a - st_b;
#endif
            issue_next();                        // tile t + PD, into the slot tile t - 1 has just left
#ifdef AAU_PW_STAMP
            PW_STAMP(st_b); st_issue += st_b - st_a; ++st_steps;
#endif
            const unsigned short* hbase = sH(cb);
            if (++cb == NBUF) cb = 0;
            const unsigned short* wbase = sWt(chunk);
            bf16x8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = ni * 16 + fr;
                wf[ni] = *(const bf16x8*)(wbase + row * BK + swz32(row, fk) * 8);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int hr = (wave * MI + mi) * HW_ + fr;
                af[mi] = *(const bf16x8*)(hbase + hr * BK + swz32(hr, fk) * 8);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#ifdef AAU_PW_STAMP
            asm volatile("" :: "v"(acc[0][0]));
            PW_STAMP(st_a); st_mfma += st_a - st_b;
#endif
        }
        // ---- per-patch epilogue ----
#ifdef AAU_PW_STAMP
        PW_STAMP(st_a);
#endif
        if constexpr (ACC) {
            // the loads are older than the nchunk * HL fills issued since
            // (the counted wait names no registers: a switch over asm statements with "+v" operands makes hipcc copy the
            //  loaded registers in front of the wait.  The empty statement behind it pins the first read of old[] below.)
            switch (a.nchunk) {
                case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            }
            static_assert(NI == 3 || NI == 6, "operand list below");
            if constexpr (NI == 6) asm volatile("" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3 % NI]), "+v"(old[4 % NI]), "+v"(old[5 % NI]) : : "memory");
            else asm volatile("" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]) : : "memory");
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + ni * 16 + 4 * fk;
            float va[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] = acc[ni][0][r]; vb[r] = acc[ni][1][r]; }
            if (q < d.Cout) {
                if (want_stats) { epi_stats(a, 0, q, va, s1[ni], s2[ni]); epi_stats(a, 0, q, vb, s1[ni], s2[ni]); }
                const int ql = ni * 16 + 4 * fk;
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(par + ql);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { va[r] += b[r]; vb[r] += b[r]; }
                }
                if (a.scale) {
                    const f32x4 sc = *(const f32x4*)(par + BQ + ql);
                    const f32x4 sh = *(const f32x4*)(par + 2 * BQ + ql);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { va[r] = va[r] * sc[r] + sh[r]; vb[r] = vb[r] * sc[r] + sh[r]; }
                }
            }
            float w[8];
            swap_pair8(va, vb, w);                            // every lane takes part
            if constexpr (ACC) {
                float o[8];
                unpack8(old[ni], o);
#pragma unroll
                for (int r = 0; r < 8; ++r) w[r] += o[r];
            }
            if (d.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) w[r] = fmaxf(w[r], 0.f);
            }
            store_b128_soff(pack8(w), rsD, evo[ni], ebase);
        }
#ifdef AAU_PW_STAMP
        PW_STAMP(st_b); st_epi += st_b - st_a;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AAU_PW_STAMP
    {
        unsigned long long st_end, rt;
        PW_STAMP(st_end);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) : : "memory");
        if (lane == 0 && a.shift) {
            unsigned long long* o = (unsigned long long*)a.shift + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
            o[0] = st_wait; o[1] = st_bar; o[2] = st_issue; o[3] = st_mfma; o[4] = st_epi; o[5] = st_steps; o[6] = st_end - st_begin; o[7] = rt;
        }
    }
#undef PW_STAMP
#endif
    if (want_stats) {
        float* sst = (float*)dsm;                      // [NW][2][BQ]
        __syncthreads();
        for (int i = tid; i < NW * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, NW, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// true when the resident-weight 1x1 kernel applies (high-resolution 1x1 convs with a small weight matrix)
bool conv1x1_resw_applicable(const aau_conv_desc* d, bool want_stats) {
    (void)want_stats;
    if (!(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo &&
          d->H % 16 == 0 && d->W % 16 == 0 && d->Cpad % 32 == 0))
        return false;
    if (getenv("AAU_NO_PW_RESW")) return false;
    const int BQ = d->Cout <= 48 ? 48 : 96;
    const int ntq = (d->Cout + BQ - 1) / BQ;
    const int64_t npatch = (int64_t)d->N * (d->H / 16) * (d->W / 16);
    // one patch per workgroup gains nothing from residency: 1024 patches, or 512 when there are several channel
    // tiles (the ConvTranspose GEMMs: -20 % on u2.up; the 96-channel gate convs at 512 patches were 5-20 % slower)
    int minp = ntq >= 2 ? 512 : 1024;
    if (const char* e = getenv("AAU_PW_MINPATCH")) minp = atoi(e);   // experiment
    return npatch >= minp && ntq <= 4 && (size_t)(d->Cpad / 32) * BQ * 64 <= 48 * 1024;
}

int conv1x1_resw_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                        const float* scale, const float* shift, float* stats, unsigned src_bytes, unsigned wpk_bytes,
                        hipStream_t s, const float* in_scale, const float* in_shift) {
    C3Args a;
    a.d = *d;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = bias; a.scale = scale; a.shift = shift; a.stats = stats;
    a.rev = next_traversal();
    a.nchunk = d->Cpad / 32;
    a.src_bytes = src_bytes;
    a.wpk_bytes = wpk_bytes;
    a.tiles_x = d->W / 16;
    a.nowide = getenv("AAU_NO_WIDE_STORE") != nullptr;
    a.tiles_y = d->H / 16;
    a.nopair = getenv("AAU_RESW_NOPAIR") != nullptr;
    const int BQ = d->Cout <= 48 ? 48 : 96;      // a 192-channel tile (activations read once) measured no faster
    const int ntq = (d->Cout + BQ - 1) / BQ;
    const int npatch = a.tiles_x * a.tiles_y * d->N;
    const size_t wbytes = ((size_t)a.nchunk * BQ * 64 + 8191) / 8192 * 8192;   // whole staging rounds of 512 threads
    // LDS: ring of pixel tiles (16 KB each) | weights | per-channel vectors
    int nbuf = 7;
    if ((size_t)nbuf * 256 * 64 + wbytes + 3 * 96 * 4 + 2 * 192 * 4 > 160 * 1024) { nbuf = 2; }
    if (const char* e = getenv("AAU_PW_NBUF")) { if (atoi(e) == 2) nbuf = 2; }   // A/B: round 2's one tile of look-ahead
    const size_t lds = (size_t)nbuf * 256 * 64 + wbytes + 3 * 96 * 4 + 2 * 192 * 4;      // ... | BNIN table
    int per_cu = lds <= 80 * 1024 ? 2 : 1;
    if (const char* e = getenv("AAU_PW_PERCU")) per_cu = atoi(e);   // experiment
    int gx = 256 * per_cu / ntq;
    if (gx > npatch) gx = npatch;
    if (gx < 1) gx = 1;
    auto go = [&](auto kern) {
        static bool attr = false;
        if (!attr) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, dim3(gx, ntq), dim3(512), lds, s, a, npatch);
    };
    // round 4's form (scalar-offset fills and stores, prefetched read-modify-write): 16-byte stores, the 7-tile ring, every
    // tensor below 2 GiB, at most 6 chunks (the loads of a patch must stay younger than the tile a step waits for)
    const int64_t Mo = (int64_t)d->N * d->H * d->W * (d->shuffle2x2 ? 4 : 1);
    const bool wide_ok = ((uintptr_t)dst & 15) == 0 && d->dst_pitch % 8 == 0 && (!d->shuffle2x2 || (d->Cout >> 2) % 8 == 0) && !a.nowide;
    const bool rs = nbuf == 7 && wide_ok && a.nchunk <= 6 && Mo * d->dst_pitch * 2 < 0x7fffffff && !getenv("AAU_PW_OLD");
    if (in_scale && !rs) { set_error("aau_conv_igemm_bnin: this 1x1 problem is not served by the scalar-offset kernel (aau_conv_bnin_ok)"); return AAU_E_INVALID; }
    prof_tag(rs ? (BQ == 48 ? "conv1x1_rs<48>" : "conv1x1_rs<96>") : (BQ == 48 ? "conv1x1_resw<48>" : "conv1x1_resw<96>"));
    if (BQ == 48) {
        if (rs) { if (d->accumulate) go(conv1x1_rs_kernel<48, 8, true>); else go(conv1x1_rs_kernel<48, 8, false>); }
        else if (nbuf == 7) go(conv1x1_resw_kernel<48, 8, 7>);
        else go(conv1x1_resw_kernel<48, 8, 2>);
    } else {
        if (rs) { if (d->accumulate) go(conv1x1_rs_kernel<96, 8, true>); else go(conv1x1_rs_kernel<96, 8, false>); }
        else if (nbuf == 7) go(conv1x1_resw_kernel<96, 8, 7>);
        else go(conv1x1_resw_kernel<96, 8, 2>);
    }
#ifdef AAU_C3S_ABLATE
    // (the launch above already ran; an ablation run times BOTH and the difference is what the switch removes)
    if (const char* e = getenv("AAU_PW_ABL")) {
        const int abl = atoi(e);
        if (BQ == 96 && abl == 2) go(conv1x1_resw_kernel<96, 8, 7, 2>);
        if (BQ == 96 && abl == 4) go(conv1x1_resw_kernel<96, 8, 7, 4>);
        if (BQ == 96 && abl == 6) go(conv1x1_resw_kernel<96, 8, 7, 6>);
        if (BQ == 48 && abl == 2) go(conv1x1_resw_kernel<48, 8, 7, 2>);
        if (BQ == 48 && abl == 4) go(conv1x1_resw_kernel<48, 8, 7, 4>);
        if (BQ == 48 && abl == 6) go(conv1x1_resw_kernel<48, 8, 7, 6>);
    }
#endif
    return check_launch("aau_conv_igemm(1x1 resident weights)");
}

// conv3x3s.hip
bool conv3x3s_applicable(const aau_conv_desc* d, const void* src, const void* dst);
int conv3x3s_launch(C3Args& a, hipStream_t s);

// true when conv3x3_launch would take the (single patch stream) resident-weight kernel, the one that serves two-plane
// operands
bool conv3x3_split_ok(const aau_conv_desc* d) {
    const bool narrow = d->Cout <= 48;
    const int BQ = narrow ? 48 : 96;
    const int nchunk = d->Cpad / 32;
    const int npatch = (d->W / 16) * (d->H / 16) * d->N;
    const size_t wbytes = ((size_t)nchunk * 9 * BQ * 64 + 8191) / 8192 * 8192;
    const size_t lds = (size_t)2 * 384 * 64 + wbytes;
    const int ntq = (d->Cout + BQ - 1) / BQ;
    return !d->accumulate && lds <= 160 * 1024 && npatch >= 1024 && ntq <= 2;
}

// true when the halo kernel applies to this descriptor
bool conv3x3_applicable(const aau_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && !d->shuffle2x2 &&
           d->H == d->Ho && d->W == d->Wo && d->H % 16 == 0 && d->W % 16 == 0 && d->Cpad % 32 == 0;
}

int conv3x3_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                   const float* scale, const float* shift, float* stats, unsigned src_bytes, unsigned wpk_bytes,
                   hipStream_t s) {
    C3Args a;
    a.d = *d;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = bias; a.scale = scale; a.shift = shift; a.stats = stats;
    a.rev = next_traversal();
    a.nchunk = d->Cpad / 32;
    a.src_bytes = src_bytes;
    a.wpk_bytes = wpk_bytes;
    a.tiles_x = d->W / 16;
    a.nowide = getenv("AAU_NO_WIDE_STORE") != nullptr;
    a.tiles_y = d->H / 16;
    a.nopair = getenv("AAU_RESW_NOPAIR") != nullptr;
    // 48 / 96 channels in and out: strips with register-resident weights (conv3x3s.hip)
    if (conv3x3s_applicable(d, src, dst)) return conv3x3s_launch(a, s);
    const bool narrow = d->Cout <= 48;
    const int BQ = narrow ? 48 : 96;
    // resident-weight persistent variant: small weight matrix, many patches, no read-modify-write epilogue
    {
        const int npatch = a.tiles_x * a.tiles_y * d->N;
        // two halo buffers + all weight tiles, the latter rounded up to whole staging rounds
        const size_t wbytes = ((size_t)a.nchunk * 9 * BQ * 64 + 8191) / 8192 * 8192;   // whole staging rounds of 512 threads
        const size_t lds = (size_t)2 * 384 * 64 + wbytes;
        const int ntq = (d->Cout + BQ - 1) / BQ;
        // two patch streams sharing the weights (2 waves per SIMD) where four halo buffers fit beside them
        {
            const size_t wb2 = ((size_t)a.nchunk * 9 * 48 * 64 + 8191) / 8192 * 8192;
            const size_t lds2 = (size_t)4 * 384 * 64 + wb2;
            const int gx2 = npatch / 2 < 256 ? npatch / 2 : 256;
            if (narrow && ntq == 1 && !d->accumulate && lds2 <= 160 * 1024 && npatch >= 2048 && npatch % (2 * gx2) == 0 &&
                d->src_split_c <= 0 && d->dst_split_c <= 0 && !getenv("AAU_NO_RESW") && !getenv("AAU_NO_RESW2")) {
                static bool attr2 = false;
                if (!attr2) {
                    hipFuncSetAttribute((const void*)conv3x3_resw2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr2 = true;
                }
                prof_tag("conv3x3_resw2");
                if (const char* e = getenv("AAU_RESW_ABL")) a.rev |= atoi(e) & 14;
                hipLaunchKernelGGL(conv3x3_resw2_kernel, dim3(gx2, 1), dim3(512), lds2, s, a, npatch);
                return check_launch("aau_conv_igemm(3x3 resident weights, two patch streams)");
            }
        }
        const bool split = d->src_split_c > 0 || d->dst_split_c > 0;
        if (!d->accumulate && lds <= 160 * 1024 && npatch >= 1024 && ntq <= 2 && (split || !getenv("AAU_NO_RESW"))) {
            static bool attr48 = false, attr96 = false;
            if (narrow && !attr48) {
                hipFuncSetAttribute((const void*)conv3x3_resw_kernel<48, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute((const void*)conv3x3_resw_kernel<48, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr48 = true;
            }
            if (!narrow && !attr96) {
                hipFuncSetAttribute((const void*)conv3x3_resw_kernel<96, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute((const void*)conv3x3_resw_kernel<96, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr96 = true;
            }
            const int gx = npatch < 256 / ntq ? npatch : 256 / ntq;
            // 8 waves on one patch (2 per SIMD): -10...20 % against 4 waves on every layer that takes this path
            const bool w8 = getenv("AAU_RESW_W4") == nullptr;
            prof_tag(narrow ? "conv3x3_resw<48>" : "conv3x3_resw<96>");
            if (narrow && w8) hipLaunchKernelGGL((conv3x3_resw_kernel<48, 8>), dim3(gx, ntq), dim3(512), lds, s, a, npatch);
            else if (narrow) hipLaunchKernelGGL((conv3x3_resw_kernel<48, 4>), dim3(gx, ntq), dim3(256), lds, s, a, npatch);
            else if (w8) hipLaunchKernelGGL((conv3x3_resw_kernel<96, 8>), dim3(gx, ntq), dim3(512), lds, s, a, npatch);
            else hipLaunchKernelGGL((conv3x3_resw_kernel<96, 4>), dim3(gx, ntq), dim3(256), lds, s, a, npatch);
            return check_launch("aau_conv_igemm(3x3 resident weights)");
        }
    }
    if (d->src_split_c > 0 || d->dst_split_c > 0) {
        set_error("aau_conv_igemm: two-plane operands are only served by the resident-weight 3x3 kernel (aau_conv_split_ok)");
        return AAU_E_INVALID;
    }
    // 16 x 32 patches with 8 waves (one workgroup per CU) halve the LDS-DMA instructions per wave, but
    // measured 8-12 % SLOWER than two 4-wave workgroups per CU (A/B on one device): opt-in only
    const bool wide_patch = !narrow && d->W % 32 == 0 && getenv("AAU_C3_PW2");
    if (wide_patch) a.tiles_x = d->W / 32;
    const int64_t grid = (int64_t)((d->Cout + BQ - 1) / BQ) * a.tiles_x * a.tiles_y * d->N;
    if (grid <= 0 || grid > 0x7fffffff) { set_error("conv3x3: grid out of range"); return AAU_E_INVALID; }
    // loader-wave variant: measured 25-30 % SLOWER than conv3x3g on every layer (A/B in one gpurun call, 2- and 3-slot
    // weight rings alike), so it is opt-in (experiments only)
    if (!wide_patch && !getenv("AAU_C3_NOGROUP") && getenv("AAU_C3_LOADER")) {
        prof_tag(narrow ? "conv3x3l<48>" : "conv3x3l<96>");
        static bool attrl = false;
        if (!attrl) {
            hipFuncSetAttribute((const void*)conv3x3l_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void*)conv3x3l_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attrl = true;
        }
        const size_t ldsl = (size_t)(2 * 336 * 32 + 3 * 2 * BQ * 32) * 2;
        if (narrow) hipLaunchKernelGGL((conv3x3l_kernel<48>), dim3((unsigned)grid), dim3(320), ldsl, s, a);
        else hipLaunchKernelGGL((conv3x3l_kernel<96>), dim3((unsigned)grid), dim3(320), ldsl, s, a);
        return check_launch("aau_conv_igemm(3x3 halo, grouped taps, loader wave)");
    }
    if (!wide_patch && !narrow && !getenv("AAU_C3_NOH") && !getenv("AAU_C3_NOGROUP") && !getenv("AAU_C3_LOADER")) {
        static bool attrh = false;
        if (!attrh) {
            hipFuncSetAttribute((const void*)conv3x3h_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attrh = true;
        }
        const size_t ldsh = (size_t)(2 * 336 * 32 + 2 * 3 * 96 * 32 + 512) * 2;
        if (!getenv("AAU_C3_NOPERSIST") && !getenv("AAU_C3_NOSTORE") && !getenv("AAU_C3_ABL") && grid < 0x7fffffff) {
            const int ntq = (d->Cout + 95) / 96;
            int G = 512 / ntq * ntq;                        // two workgroups per CU, a multiple of the channel-tile count
            if (G > grid) G = (int)grid;
            static bool attrp = false;
            if (!attrp) {
                hipFuncSetAttribute((const void*)conv3x3p_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attrp = true;
            }
            prof_tag("conv3x3p<96>");
            hipLaunchKernelGGL((conv3x3p_kernel<96>), dim3((unsigned)G), dim3(256), ldsh, s, a, (int)grid);
            return check_launch("aau_conv_igemm(3x3 halo, column steps, persistent)");
        }
        prof_tag("conv3x3h<96>");
        if (getenv("AAU_C3_NOSTORE")) a.rev |= 2;
        if (const char* e = getenv("AAU_C3_ABL")) a.rev |= atoi(e) & 12;
        hipLaunchKernelGGL((conv3x3h_kernel<96>), dim3((unsigned)grid), dim3(256), ldsh, s, a);
        return check_launch("aau_conv_igemm(3x3 halo, column steps)");
    }
    prof_tag(narrow ? "conv3x3g<48>" : "conv3x3g<96>");
    if (!wide_patch && !getenv("AAU_C3_NOGROUP")) {
        if (narrow) hipLaunchKernelGGL((conv3x3g_kernel<48>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3g_kernel<96>), dim3((unsigned)grid), dim3(256), 0, s, a);
        return check_launch("aau_conv_igemm(3x3 halo, grouped taps)");
    }
    if (narrow) hipLaunchKernelGGL((conv3x3_kernel<48, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else if (wide_patch) hipLaunchKernelGGL((conv3x3_kernel<96, 2>), dim3((unsigned)grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_kernel<96, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return check_launch("aau_conv_igemm(3x3 halo)");
}

}  // namespace aau

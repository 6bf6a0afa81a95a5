// Implicit-GEMM convolution on MFMA for gfx950 (forward and data-gradient).
//
//   dst[m][q] = epi( sum_{tap,c} src[gather(m,tap)][c] * wpk[q][tap][c] )
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = taps x Cin.  The MFMA A operand is
// the WEIGHT tile (rows = output channels) and the B operand the gathered ACTIVATION
// tile (columns = pixels), so v_mfma_f32_16x16x32_bf16 leaves each lane with 4
// consecutive output channels of one pixel: the NHWC store is an 8-byte packed write
// and the per-channel BatchNorm statistics reduce over the 16 lanes of a DPP row.
//
// Staging: both tiles go global -> LDS with global_load_lds_dwordx4 (16 B per lane, no
// VGPR round trip).  The LDS image is lane-linear, so the bank-conflict swizzle is
// applied to the per-lane SOURCE address and undone on the ds_read_b128 side (same
// XOR).  Out-of-image taps, rows past M and channels past Cin read a zero page.
// Two LDS buffers: the loads of K-step t+1 are in flight during the MFMAs of step t.
//
// Replaces: Conv2d inside ConvBNReLU (pipeline:63), the ASPP 1x1 / dilated 3x3 / pool /
// projection convs (:71-78), the gate 1x1 convs (:88-89), ConvTranspose2d(2,2) (:101,
// as a GEMM with N = 4*Co and a pixel-shuffle store) and their input gradients.
#include "common.h"
#ifdef AAU_NO_MFMA_PIN         /* A/B build (build.py -DAAU_NO_MFMA_PIN --tag=nopin): the scheduler's own order */
#define AAU_PIN_SB()
#else
#define AAU_PIN_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#include <type_traits>

namespace aau {


// physical 16-B slot of logical chunk `lc` in LDS row `row` (an involution)
template <int BK>
__device__ __forceinline__ int swz(int row, int lc) {
    if constexpr (BK == 64) {
        return lc ^ ((row >> 1) & 7);
    } else {
        return lc ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
    }
}

struct IgemmArgs {
    aau_conv_desc d;
    const unsigned short* src;
    const unsigned short* wpk;
    unsigned short* dst;
    const float* bias;
    const float* scale;
    const float* shift;
    float* stats;
    int M;
    int nchunk;  // Cpad / BK
    unsigned src_bytes;  // extent of the gather source (buffer descriptor range; out-of-range reads return 0)
    unsigned wpk_bytes;
    int rev;             // 1: walk the tiles from the end (aau_traverse)
};

// SMALL = half-height pixel tile (64 x 96): for the 32x32-resolution layers (M = 8192) the regular tiling
// yields only 256 workgroups (one per CU, 1 wave per SIMD); 512 smaller ones hide twice the latency.
//
// BQ = 192 is the wide tile for the long-K GEMMs of the ASPP bridge and the ConvTranspose layers: 128 pixels x 192
// channels on 8 waves (2 x 4, the same 64 x 48 wave tile), filled THROUGH REGISTERS into two LDS buffers (80 KiB,
// dynamic).  What bounds those layers is the rate at which a CU can fill LDS: a K-step of the 128 x 96 tile moves
// 28.7 KB for 1.57 MFLOP (55 FLOP/B), the 128 x 192 tile 41 KB for 3.15 MFLOP (77 FLOP/B), and LDS-DMA tops out near
// 100 GB/s per CU without overlapping the multiply (see the main loop).
template <int BK, int BQ, bool SMALL = false>
__global__ __launch_bounds__((BQ == 192) ? 512 : 256) void igemm_kernel(const IgemmArgs a) {
    constexpr int NWV = (BQ == 192) ? 8 : 4;  // waves
    constexpr bool WIDE = (BQ == 192);        // register-staged main loop, 8 waves
    constexpr int BP = SMALL ? 64 : ((BQ == 96 || BQ == 192) ? 128 : 256);
    static_assert(!SMALL || BQ == 96, "the small tile is a 2x2 wave layout");
    static_assert(BQ != 192 || BK == 64, "the wide tile is built for 64-channel K-steps");
    constexpr int SLOTS = BK / 8;            // 16-B slots per LDS row
    constexpr int RPI = 64 / SLOTS;          // rows covered by one wave-wide glds
    constexpr int NA = BP / (NWV * RPI);     // activation loads per thread per K-step
    constexpr int NW = (BQ + NWV * RPI - 1) / (NWV * RPI);  // weight loads per thread (last may be partial)
    constexpr int MI = SMALL ? 2 : 4, NI = 3; // wave tile: 64 (32) pixels x 48 channels
    constexpr int WPX = MI * 16;
    constexpr int KSUB = BK / 32;

#ifdef AAU_IGEMM_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
    unsigned long long st_loop1 = 0;
#endif
    __shared__ __attribute__((aligned(16))) unsigned short ssmem[(BQ == 192) ? 8 : 2 * (BQ + BP) * BK];
    extern __shared__ __attribute__((aligned(16))) unsigned short dsmem[];
    unsigned short* const smem = (BQ == 192) ? dsmem : ssmem;
    auto sW = [&](int buf) -> unsigned short* { return smem + buf * ((BQ + BP) * BK); };
    auto sA = [&](int buf) -> unsigned short* { return smem + buf * ((BQ + BP) * BK) + BQ * BK; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wp = (BQ == 192) ? (wave >> 2) : (BQ == 96) ? (wave >> 1) : wave;
    const int wq = (BQ == 192) ? (wave & 3) : (BQ == 96) ? (wave & 1) : 0;

    // XCD-aware tile order: consecutive tile ids (same pixel tile, different channel
    // tiles / neighbouring pixel tiles) share an XCD's L2.  Bijective remap.
    const int ntq = (d.Cout + BQ - 1) / BQ;
    const int nwg = gridDim.x;
    int bid = (a.rev & 1) ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tq = bid % ntq;
    const int tp = bid / ntq;
    const int q0 = tq * BQ;
    const int m0 = tp * BP;

    const int HoWo = d.Ho * d.Wo;
    constexpr unsigned OOB = 0x80000000u;  // beyond any descriptor range: the load returns zeros

    // Buffer descriptors (wave-uniform): hardware range check = free zero fill for padding taps,
    // rows past M and channels past Cin; 32-bit offsets keep the per-load address math to ~1 VALU op.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // ---- per-thread row bookkeeping for the activation gather ----
    int pix0[NA];   // n*H*W
    int yx0[NA];    // (y0 << 16) | (x0 & 0xffff), y0/x0 = out*stride - pad; 0x80000000 marks "row past M"
    const int slot = lane % SLOTS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (i * NWV + wave) * RPI + lane / SLOTS;
        const int m = m0 + row;
        if (m < a.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int yo = rem / d.Wo;
            const int xo = rem - yo * d.Wo;
            pix0[i] = n * d.H * d.W;
            yx0[i] = ((yo * d.stride - d.pad) << 16) | ((xo * d.stride - d.pad) & 0xffff);
        } else {
            pix0[i] = 0;
            yx0[i] = (int)0x80000000;
        }
    }
    // logical chunk this lane fetches for each of its rows (swizzle on the source side)
    int lcA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) lcA[i] = swz<BK>((i * NWV + wave) * RPI + lane / SLOTS, slot);
    const int T = d.KH * d.KW;
    unsigned wbase[NW];  // byte offset of (row q, tap 0, this lane's logical chunk) in the packed weights
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int row = (j * NWV + wave) * RPI + lane / SLOTS;
        const int lc = swz<BK>(row, slot);
        wbase[j] = (row < BQ && q0 + row < d.Cout) ? (unsigned)(((q0 + row) * T * d.Cpad + lc * 8) * 2) : OOB;
    }
    // channel tail: in the last chunk of a tap, chunks that start at or beyond Cin read zeros
    const int tail_c0 = (a.nchunk - 1) * BK;
    bool tail_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) tail_ok[i] = tail_c0 + lcA[i] * 8 < d.Cin;
    const bool has_tail = d.Cpad != d.Cin;

    // ---- active taps: a tap whose every row of this tile is out of the image is skipped ----
    unsigned tapmask = (T >= 32) ? 0xffffffffu : ((1u << T) - 1u);
    if (d.dil > 1) {
        unsigned mine = 0;
        for (int t = 0; t < T; ++t) {
            const int dy = (t / d.KW) * d.dil, dx = (t % d.KW) * d.dil;
            bool any = false;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int y = (yx0[i] >> 16) + dy, x = (short)(yx0[i] & 0xffff) + dx;
                any |= (yx0[i] != (int)0x80000000) && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            }
            if (any) mine |= 1u << t;
        }
        __shared__ unsigned s_mask;
        if (tid == 0) s_mask = 0;
        __syncthreads();
        // one LDS atomic per WAVE: 512 lanes adding to the same word serialise (the stamps put 4.5 us of a 7.4-us prologue here)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor(mine, o, 64);
        if (lane == 0 && mine) atomicOr(&s_mask, mine);
        __syncthreads();
        tapmask = __builtin_amdgcn_readfirstlane(s_mask);
        __syncthreads();
        if (tapmask == 0) tapmask = 1;  // still run one (all-zero) step so the epilogue sees zeros
    }

    unsigned abase[NA];  // byte offset of (gathered pixel, this lane's logical chunk) for the current tap
    auto set_tap = [&](int tap) {
        const int dy = (tap / d.KW) * d.dil, dx = (tap % d.KW) * d.dil;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int y = (yx0[i] >> 16) + dy, x = (short)(yx0[i] & 0xffff) + dx;
            const bool ok = (yx0[i] != (int)0x80000000) && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            abase[i] = ok ? (unsigned)(((pix0[i] + y * d.W + x) * d.src_pitch + lcA[i] * 8) * 2) : OOB;
        }
    };

    auto stage = [&](int buf, int tap, int chunk) {
        const int soffA = chunk * BK * 2;                      // scalar byte offsets
        const int soffW = (tap * d.Cpad + chunk * BK) * 2;
        const bool last = has_tail && chunk == a.nchunk - 1;   // uniform
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned v = ((last && !tail_ok[i]) || (a.rev & 4)) ? OOB : abase[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA(buf) + (i * NWV + wave) * RPI * BK), 16, (int)v,
                                                     soffA, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if ((j * NWV + wave) * RPI < BQ)  // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(sW(buf) + (j * NWV + wave) * RPI * BK), 16,
                                                         (a.rev & 8) ? (int)OOB : (int)wbase[j], soffW, 0, 0);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;   // fragment row (pixel for B operand, channel for A operand)
    const int fk = lane >> 4;   // k-group
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
            bf16x8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wq * 48 + ni * 16 + fr;
                wf[ni] = *(const bf16x8*)(sW(buf) + row * BK + swz<BK>(row, kk * 4 + fk) * 8);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wp * WPX + mi * 16 + fr;
                af[mi] = *(const bf16x8*)(sA(buf) + row * BK + swz<BK>(row, kk * 4 + fk) * 8);
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
    };

    // ---- main loop over (active tap, channel chunk) ----
    unsigned mask = tapmask;
    int tap = __builtin_ctz(mask);
    mask &= mask - 1;
    int chunk = 0;
    set_tap(tap);
    if constexpr (WIDE) {
        // REGISTER-staged fill: global -> VGPR (buffer_load_dwordx4) -> ds_write_b128, two LDS buffers, the loads of
        // K-steps t+1 and t+2 in flight in two register sets.  The LDS-DMA form of this loop (three-deep ring) ran the
        // bridge GEMMs at 58-60 us: 0.40 us per K-step to push 40 wave-wide DMA instructions (1 KiB each, ~24 clk apiece
        // even when every address is out of range) PLUS 0.42 us to multiply -- the two did not overlap, whether the
        // DMA instructions were issued in a burst or between the MFMAs.  Through registers the data returns on the
        // vector-memory path and enters LDS at 128 B/clk, beside the ds_read traffic.
        constexpr int NL = NA + NW;
#ifndef AAU_WIDE_MPW
#define AAU_WIDE_MPW 2
#endif
        constexpr int WIDE_MPW = AAU_WIDE_MPW;      // MFMAs of the first group in front of each LDS write
        static_assert(BQ % (NWV * RPI) == 0, "uniform load count per step");
        const int nsteps = __builtin_popcount(tapmask) * a.nchunk;
        u32x4 R0[NL], R1[NL];
        auto gload = [&](u32x4 (&R)[NL]) {       // fetch the step at the cursor, then advance the cursor
            const bool last = has_tail && chunk == a.nchunk - 1;
            const int soffA = chunk * BK * 2, soffW = (tap * d.Cpad + chunk * BK) * 2;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned v = ((last && !tail_ok[i]) || (a.rev & 4)) ? OOB : abase[i];
                R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)v, soffA, 0);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j)
                R[NA + j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (a.rev & 8) ? (int)OOB : (int)wbase[j], soffW, 0);
            if (++chunk == a.nchunk) {
                chunk = 0;
                if (mask) {
                    tap = __builtin_ctz(mask);
                    mask &= mask - 1;
                    set_tap(tap);
                }
            }
        };
        auto lwrite = [&](int buf, const u32x4 (&R)[NL]) {   // lane-linear image, as the DMA form writes it
#pragma unroll
            for (int i = 0; i < NA; ++i) *(u32x4*)(sA(buf) + (i * NWV + wave) * RPI * BK + lane * 8) = R[i];
#pragma unroll
            for (int j = 0; j < NW; ++j) *(u32x4*)(sW(buf) + (j * NWV + wave) * RPI * BK + lane * 8) = R[NA + j];
        };
        // fragments of one 32-channel sub-step: read one sub-step AHEAD of the MFMAs that use them.  Read right before
        // use, the 56 wave-wide ds_read_b128 of a sub-step (8 waves x 7) all queue behind the same barrier and every
        // wave idles ~220 clk for its data, twice per K-step (measured: 0.74 us per step with no memory traffic at all
        // against 0.32 us of MFMA work).
        struct Frag { bf16x8 w[NI], a[MI]; };
        auto read_frags = [&](int buf, int kk, Frag& f) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wq * 48 + ni * 16 + fr;
                f.w[ni] = *(const bf16x8*)(sW(buf) + row * BK + swz<BK>(row, kk * 4 + fk) * 8);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wp * WPX + mi * 16 + fr;
                f.a[mi] = *(const bf16x8*)(sA(buf) + row * BK + swz<BK>(row, kk * 4 + fk) * 8);
            }
        };
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = AAU_MFMA16(f.w[ni], f.a[mi], acc[ni][mi], 0, 0, 0);
        };
        // Iteration t enters with F0 = fragments (t, sub-step 0), step t in LDS buffer t & 1, step t+1 in flight in
        // Rnext:   read F1 <- (t, 1); fetch step t+2 -> Rfree; MFMA(F0); write step t+1 into buffer (t+1) & 1 (last read
        // as (t-1, 1), complete before the barrier of iteration t-1); barrier; read F0 <- (t+1, 0); MFMA(F1).
        // The steady state is branch-free (the compiler's vmcnt bookkeeping then waits for the OLDER register set only
        // and leaves the younger fetch in flight); the last two steps run without a fetch.
        Frag F0, F1;
#ifdef AAU_IGEMM_STAMP
        // diagnostic build only (build.py -DAAU_IGEMM_STAMP --tag=stamp; scripts/probes/igemm_stamp.py): s_memtime at the three
        // points of an iteration where the LDS counter is (nearly) drained anyway; phase sums per wave -> a.shift (scale null)
        unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_last = 0;
        auto stamp = [&]() -> unsigned long long {
            unsigned long long v;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
            return v;
        };
#define AAU_ST(acc)                                          \
        {                                                    \
            const unsigned long long now_ = stamp();         \
            acc += now_ - st_last;                           \
            st_last = now_;                                  \
        }
#else
#define AAU_ST(acc)
#endif
        auto iter = [&](int t, u32x4 (&Rnext)[NL], u32x4 (&Rfree)[NL], auto fetch, auto write) {
            // sched_barrier: the reads must ISSUE ahead of the MFMAs they hide behind (left alone, the scheduler sinks
            // them to just before their first use, one sub-step later)
            AAU_ST(st_b)                          // since the barrier: first fragment reads of this step + second MFMA group
            read_frags(t & 1, 1, F1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(fetch)::value) gload(Rfree);
            AAU_PIN_SB();
#ifdef AAU_IGEMM_STAMP2
            AAU_ST(st_d)                          // (perturbing: drains the fragment reads) top -> fetch issued, fragments landed
            AAU_PIN_SB();
#endif
            mma(F0);
            if constexpr (decltype(write)::value) {
                lwrite((t + 1) & 1, Rnext);
                // Left alone, the scheduler sinks ten of the twelve MFMAs of this group BELOW the barrier (they are pure
                // register operations): the wait for the fetched step, its five ds_write_b128, the wait for them and the
                // barrier then run with the matrix pipe idle, every step.  Two MFMAs, one write, five times, two MFMAs:
#ifndef AAU_NO_MFMA_PIN      /* A/B build: the scheduler's own order */
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, WIDE_MPW, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // DS write
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NI * MI - NL * WIDE_MPW, 0);
#endif
                AAU_PIN_SB();
                AAU_ST(st_c)                      // since the top: second fragment reads, fetch, first MFMA group + LDS writes
                __syncthreads();
                AAU_ST(st_a)                      // wait for the writes + barrier
                read_frags((t + 1) & 1, 0, F0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(F1);
        };
        using Y = std::integral_constant<bool, true>;
        using N = std::integral_constant<bool, false>;
        gload(R0);
        if (nsteps > 1) gload(R1);
        lwrite(0, R0);
        __syncthreads();
        read_frags(0, 0, F0);
#ifdef AAU_IGEMM_STAMP
        st_last = stamp();
        const unsigned long long st_begin = st_last, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        int t = 0;
        for (; t + 3 < nsteps; t += 2) {          // steps t+2 and t+3 exist
            iter(t, R1, R0, Y{}, Y{});
            iter(t + 1, R0, R1, Y{}, Y{});
        }
        // 1, 2 or 3 steps left; the register set holding step t+1 is R1
        if (t + 2 < nsteps) {                     // three left
            iter(t, R1, R0, Y{}, Y{});
            iter(t + 1, R0, R1, N{}, Y{});
            iter(t + 2, R1, R0, N{}, N{});
        } else if (t + 1 < nsteps) {              // two left
            iter(t, R1, R0, N{}, Y{});
            iter(t + 1, R0, R1, N{}, N{});
        } else {
            iter(t, R1, R0, N{}, N{});
        }
#ifdef AAU_IGEMM_STAMP
        if (lane == 0 && a.scale == nullptr && a.shift != nullptr && blockIdx.x < 64) {
            unsigned long long* dbg = (unsigned long long*)a.shift + ((size_t)blockIdx.x * NWV + wave) * 10;
            dbg[0] = st_a; dbg[1] = st_b; dbg[2] = st_c; dbg[3] = stamp() - st_begin; dbg[4] = (unsigned long long)nsteps;
            dbg[5] = __builtin_amdgcn_s_memrealtime() - st_rt0;
            dbg[6] = st_begin - st_entry;        // prologue
            dbg[7] = st_d;
        }
        st_loop1 = stamp();
#endif
    } else {
    stage(0, tap, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    while (true) {
        int ntap = tap, nchk = chunk + 1;
        bool more = true;
        if (nchk == a.nchunk) {
            nchk = 0;
            if (mask) {
                ntap = __builtin_ctz(mask);
                mask &= mask - 1;
                set_tap(ntap);
            } else {
                more = false;
            }
        }
        if (more) stage(buf ^ 1, ntap, nchk);
        compute(buf);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        tap = ntap;
        chunk = nchk;
    }
    }

    // ---- epilogue ----
    // lane holds acc[ni][mi][r] = D[channel q0 + wq*48 + ni*16 + 4*fk + r][pixel m0 + wp*64 + mi*16 + fr]
    const bool want_stats = a.stats != nullptr;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;

    // 16-byte epilogue stores (common.h: swap_pair8) when the destination allows them: the quads of pixel rows mi, mi+1
    // are exchanged across k-groups so that a lane owns 8 consecutive channels of one pixel
    const bool wide = ((uintptr_t)a.dst & 15) == 0 && d.dst_pitch % 8 == 0 && (!d.shuffle2x2 || (d.Cout >> 2) % 8 == 0) &&
                      !(a.rev & 16);
    if (wide) {
        static_assert(MI % 2 == 0, "pixel rows are stored in pairs");
        const int Co = d.Cout >> 2;
#pragma unroll
        for (int mp = 0; mp < MI; mp += 2) {
            const int ml = m0 + wp * WPX + (mp + (fk & 1)) * 16 + fr;      // the pixel this lane stores after the swap
            int n = 0, yo = 0, xo = 0;
            if (d.shuffle2x2 && ml < a.M) {
                n = ml / HoWo;
                const int rem = ml - n * HoWo;
                yo = rem / d.Wo;
                xo = rem - yo * d.Wo;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int q = q0 + wq * 48 + ni * 16 + 4 * fk;
                float va[4], vb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[r] = acc[ni][mp][r]; vb[r] = acc[ni][mp + 1][r]; }
                if (q < d.Cout) {
                    if (want_stats) {      // rows past M hold zeros (their loads were out of range)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {     // in the order of the 8-byte form: same bits
                            s1[ni][r] += va[r]; s2[ni][r] += va[r] * va[r];
                            s1[ni][r] += vb[r]; s2[ni][r] += vb[r] * vb[r];
                        }
                    }
                    const int qv = d.shuffle2x2 ? q % Co : q;
                    if (a.bias) {
                        const f32x4 b = *(const f32x4*)(a.bias + qv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { va[r] += b[r]; vb[r] += b[r]; }
                    }
                    if (a.scale) {
                        const f32x4 sc = *(const f32x4*)(a.scale + qv);
                        const f32x4 sh = *(const f32x4*)(a.shift + qv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { va[r] = va[r] * sc[r] + sh[r]; vb[r] = vb[r] * sc[r] + sh[r]; }
                    }
                }
                float w[8];
                swap_pair8(va, vb, w);                                     // every lane takes part
                const int qw = q0 + wq * 48 + ni * 16 + 8 * (fk >> 1);
                if (qw >= d.Cout || ml >= a.M) continue;
                unsigned short* out;
                if (d.shuffle2x2) {
                    const int pos = qw / Co;
                    const int64_t op = ((int64_t)n * (2 * d.Ho) + (2 * yo + (pos >> 1))) * (2 * d.Wo) + (2 * xo + (pos & 1));
                    out = a.dst + op * d.dst_pitch + (qw - pos * Co);
                } else {
                    out = a.dst + (int64_t)ml * d.dst_pitch + qw;
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
                *(u32x4*)out = pack8(w);
            }
        }
    } else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wp * WPX + mi * 16 + fr;
        const bool mok = m < a.M;
        int64_t pixel = m;
        int n = 0, yo = 0, xo = 0;
        if (d.shuffle2x2 && mok) {
            n = m / HoWo;
            const int rem = m - n * HoWo;
            yo = rem / d.Wo;
            xo = rem - yo * d.Wo;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + wq * 48 + ni * 16 + 4 * fk;
            if (!mok || q >= d.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
            if (want_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[ni][r] += v[r]; s2[ni][r] += v[r] * v[r]; }
            }
            // per-channel vectors are indexed by the real output channel (co for the pixel-shuffle store)
            const int qv = d.shuffle2x2 ? q % (d.Cout >> 2) : q;
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
            if (d.shuffle2x2) {
                const int Co = d.Cout >> 2;
                const int pos = q / Co, co = q - pos * Co;
                const int64_t op = ((int64_t)n * (2 * d.Ho) + (2 * yo + (pos >> 1))) * (2 * d.Wo) + (2 * xo + (pos & 1));
                out = a.dst + op * d.dst_pitch + co;
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

    if (want_stats) {
        // per-wave row sums (DPP), combined across the waves in LDS (the tiles are dead now),
        // then ONE global atomic per channel and workgroup
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)smem;                      // [NWV][2][BQ]
        __syncthreads();                                // every wave is done reading the LDS tiles
        for (int i = tid; i < NWV * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[wq * 48 + ni * 16 + 4 * fk + r] = x1;
                    mine[BQ + wq * 48 + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, NWV, BQ, tid, q0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
#ifdef AAU_IGEMM_STAMP
    if constexpr (WIDE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && a.scale == nullptr && a.shift != nullptr && blockIdx.x < 64)
            ((unsigned long long*)a.shift)[((size_t)blockIdx.x * NWV + wave) * 10 + 8] = __builtin_amdgcn_s_memtime() - st_loop1;   // epilogue
    }
#endif
}

// conv3x3.hip
bool conv3x3_applicable(const aau_conv_desc* d);
bool conv3x3_split_ok(const aau_conv_desc* d);
int conv3x3_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                   const float* scale, const float* shift, float* stats, unsigned src_bytes, unsigned wpk_bytes,
                   hipStream_t s);

bool conv1x1_resw_applicable(const aau_conv_desc* d, bool want_stats);
int conv1x1_resw_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                        const float* scale, const float* shift, float* stats, unsigned src_bytes, unsigned wpk_bytes,
                        hipStream_t s);

template <int BK, int BQ, bool SMALL = false>
static int launch(const IgemmArgs& a, hipStream_t s) {
    constexpr int BP = SMALL ? 64 : ((BQ == 96 || BQ == 192) ? 128 : 256);
    const int ntq = (a.d.Cout + BQ - 1) / BQ;
    const int ntp = (a.M + BP - 1) / BP;
    const int64_t grid = (int64_t)ntq * ntp;
    if (grid <= 0 || grid > 0x7fffffff) {
        set_error("aau_conv_igemm: grid %lld out of range", (long long)grid);
        return AAU_E_INVALID;
    }
    {
        char tag[AAU_PROF_TAG_LEN];
        snprintf(tag, sizeof(tag), "igemm<%d,%d,%d>%s", BK, BQ, SMALL ? 1 : 0, a.d.KH * a.d.KW > 1 ? (a.d.dil > 1 ? " dilated" : " taps") : "");
        prof_tag(tag);
    }
    if constexpr (BQ == 192) {
        constexpr size_t lds = (size_t)2 * (BQ + BP) * BK * 2;
        static bool attr = false;
        if (!attr) {
            // the kernel also has a few static words (tap mask): ask for what the ring needs, not for all 160 KiB
            if (hipFuncSetAttribute((const void*)igemm_kernel<BK, BQ, SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                set_error("aau_conv_igemm: cannot reserve %zu bytes of LDS", lds);
                return AAU_E_INVALID;
            }
            attr = true;
        }
        hipLaunchKernelGGL((igemm_kernel<BK, BQ, SMALL>), dim3((unsigned)grid), dim3(512), lds, s, a);
    } else {
        hipLaunchKernelGGL((igemm_kernel<BK, BQ, SMALL>), dim3((unsigned)grid), dim3(256), 0, s, a);
    }
    return check_launch("aau_conv_igemm");
}

}  // namespace aau

static int conv_dispatch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst,
                         const float* bias, const float* scale, const float* shift, float* stats, void* stream) {
    using namespace aau;
    AAU_REQUIRE(d && src && wpk && dst, "aau_conv_igemm: null pointer");
    AAU_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0, "aau_conv_igemm: Cin=%d must be a positive multiple of 8", d->Cin);
    AAU_REQUIRE(d->Cout > 0 && d->Cout % 8 == 0, "aau_conv_igemm: Cout=%d must be a positive multiple of 8", d->Cout);
    AAU_REQUIRE(d->Cpad >= d->Cin && d->Cpad % 32 == 0, "aau_conv_igemm: Cpad=%d must be a multiple of 32 >= Cin", d->Cpad);
    {   // a pixel's channels fit in its row (each plane's share when the source has two planes)
        const int row_c = d->src_split_c > 0 ? (d->src_split_c > d->Cin - d->src_split_c ? d->src_split_c : d->Cin - d->src_split_c) : d->Cin;
        AAU_REQUIRE(d->src_pitch >= row_c && d->src_pitch % 8 == 0, "aau_conv_igemm: src_pitch=%d", d->src_pitch);
    }
    AAU_REQUIRE(d->dst_pitch % 4 == 0, "aau_conv_igemm: dst_pitch=%d must be a multiple of 4", d->dst_pitch);
    AAU_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= 16, "aau_conv_igemm: taps %dx%d", d->KH, d->KW);
    AAU_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "aau_conv_igemm: empty shape");
    AAU_REQUIRE(d->H < 32768 && d->W < 32768, "aau_conv_igemm: spatial dims must be < 32768");
    AAU_REQUIRE((int64_t)d->N * d->H * d->W < 0x7fffffff && (int64_t)d->N * d->Ho * d->Wo < 0x7fffffff,
                "aau_conv_igemm: pixel count overflows int32");
#ifndef ABL_STAMP
#ifndef AAU_IGEMM_STAMP      /* the diagnostic build takes its stamp buffer through `shift` */
    AAU_REQUIRE((scale == nullptr) == (shift == nullptr), "aau_conv_igemm: scale and shift come together");
#endif
#endif
    AAU_REQUIRE(!d->shuffle2x2 || (d->Cout % 32 == 0), "aau_conv_igemm: shuffle2x2 needs Cout %% 32 == 0");
    AAU_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)wpk & 15) == 0 && ((uintptr_t)dst & 7) == 0,
                "aau_conv_igemm: pointers must be 16-byte (src, wpk) / 8-byte (dst) aligned");
    IgemmArgs a;
    a.d = *d;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = bias; a.scale = scale; a.shift = shift; a.stats = stats;
    a.M = d->N * d->Ho * d->Wo;
    const bool split = d->src_split_c > 0 || d->dst_split_c > 0;
    if (split) {
        AAU_REQUIRE(conv3x3_applicable(d) && conv3x3_split_ok(d),
                    "aau_conv_igemm: two-plane operands are only served by the resident-weight 3x3 kernel (aau_conv_split_ok)");
        AAU_REQUIRE(d->src_split_c % 8 == 0 && d->src_split_c < d->Cin && d->src_split_off % 8 == 0 && d->src_split_off >= 0 &&
                        d->dst_split_c % 4 == 0 && d->dst_split_c < d->Cout && d->dst_split_off % 4 == 0 && d->dst_split_off >= 0,
                    "aau_conv_igemm: split_c / split_off must be aligned (8 source, 4 destination elements) and inside the channel range");
    }
    const int64_t src_bytes = (((int64_t)d->N * d->H * d->W - 1) * d->src_pitch + d->Cin +
                               (d->src_split_c > 0 ? d->src_split_off - d->src_split_c : 0)) * 2;
    const int64_t wpk_bytes = (int64_t)d->Cout * d->KH * d->KW * d->Cpad * 2;
    AAU_REQUIRE(src_bytes < 0x7fffffff && wpk_bytes < 0x7fffffff,
                "aau_conv_igemm: source (%lld B) / packed weights (%lld B) must stay below 2 GiB", (long long)src_bytes,
                (long long)wpk_bytes);
    a.src_bytes = (unsigned)src_bytes;
    a.wpk_bytes = (unsigned)wpk_bytes;
    const bool bk64 = (d->Cpad % 64 == 0);
    a.nchunk = d->Cpad / (bk64 ? 64 : 32);
    const double flops = 2.0 * a.M * (double)d->Cout * d->Cin * d->KH * d->KW;
    ProfScope prof(0, flops, (hipStream_t)stream);
    // algorithmic HBM bytes: every input / output element and every weight once (bf16)
    prof_tag(nullptr, 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)a.M * d->Cout * (d->accumulate ? 2 : 1) +
                             (double)d->Cout * d->KH * d->KW * d->Cin));
    a.rev = 0;
    if (conv3x3_applicable(d))
        return conv3x3_launch(d, src, wpk, dst, bias, scale, shift, stats, a.src_bytes, a.wpk_bytes, (hipStream_t)stream);
    if (conv1x1_resw_applicable(d, stats != nullptr))
        return conv1x1_resw_launch(d, src, wpk, dst, bias, scale, shift, stats, a.src_bytes, a.wpk_bytes, (hipStream_t)stream);
    a.rev = next_traversal();
    if (getenv("AAU_NO_WIDE_STORE")) a.rev |= 16;     // experiment: 8-byte epilogue stores
    const bool narrow = d->Cout <= 48;
    // long-K, few-tile problems (bridge at 32x32): halve the pixel tile to double the workgroup count
    const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((d->Cout + 95) / 96);
    // long-K problems whose channel count fills whole 192-wide tiles and that still yield a workgroup per CU: the wide
    // tile (L2 -> LDS fill is what bounds them, see the kernel's header).  AAU_IGEMM_WIDE=0 / 1 forces the choice.
    {
        const int64_t tiles192 = (int64_t)((a.M + 127) / 128) * (d->Cout / 192);
        bool wide = bk64 && d->Cout % 192 == 0 && tiles192 >= 224 && a.nchunk * d->KH * d->KW >= 6;
        if (const char* e = getenv("AAU_IGEMM_WIDE")) wide = bk64 && d->Cout % 192 == 0 && atoi(e) != 0;
        if (wide) {
            if (const char* e = getenv("AAU_IGEMM_ABL")) a.rev |= atoi(e) & 14;   // timing ablation, results are wrong
            return launch<64, 192>(a, (hipStream_t)stream);
        }
    }
    if (bk64 && !narrow && tiles128 <= 384 && a.nchunk * d->KH * d->KW >= 16) return launch<64, 96, true>(a, (hipStream_t)stream);
    if (bk64) return narrow ? launch<64, 48>(a, (hipStream_t)stream) : launch<64, 96>(a, (hipStream_t)stream);
    return narrow ? launch<32, 48>(a, (hipStream_t)stream) : launch<32, 96>(a, (hipStream_t)stream);
}

extern "C" int aau_conv_igemm(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk,
                              aau_bf16* dst, const float* bias, const float* scale, const float* shift,
                              aau_stat* stats, int64_t stats_bytes, void* stream) {
    AAU_REQUIRE(d != nullptr, "aau_conv_igemm: null descriptor");
    AAU_CHECK_STAT("aau_conv_igemm", stats, stats_bytes, d->Cout);
    return conv_dispatch(d, src, wpk, dst, bias, scale, shift, (float*)stats, stream);
}

// wgrad3x3.hip
namespace aau { bool wgrad3x3_applicable(const aau_conv_desc* d); }

extern "C" int aau_conv_split_ok(const aau_conv_desc* d, int mode) {
    if (!d) return 0;
    if (mode == 0) return aau::conv3x3_applicable(d) && aau::conv3x3_split_ok(d) ? 1 : 0;
    return aau::wgrad3x3_applicable(d) && d->dst_split_c <= 0 ? 1 : 0;
}

extern "C" int aau_conv_is_halo3x3(const aau_conv_desc* d) {
    return d && aau::conv3x3_applicable(d) && !d->accumulate ? 1 : 0;
}

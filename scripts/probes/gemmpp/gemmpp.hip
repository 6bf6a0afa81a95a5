// Ping-pong GEMM for the long-K 1x1 convolutions of the low-resolution levels (gfx950): the ASPP projection 3840 -> 768 and
// its input gradient (pipeline:78), ConvTranspose2d(2,2) forward (pixel-shuffle store) and input gradient (2x2 / stride-2
// gather) of the deep decoder levels (pipeline:101).
//
//   dst[m][q] = sum_{tap, c} src[pixel(m, tap)][c] * wpk[q][tap][c]  (+ bias[q])
//
// The wide tile of igemm.hip (128 x 192 on eight waves in lockstep, filled through registers) runs these at 0.42-0.76
// PFLOP/s: every wave issues its fills, then every wave multiplies.  Here the workgroup is TWO four-wave groups that work
// on the SAME 128 x 192 output tile half a step apart (the form of wgradL.hip / wgrad3x3r.hip): group g takes the K-steps
// g, g + 2, ... with its own three-slot ring and its own accumulators; between the two barriers of a tick one group
// multiplies (10 fragment reads + 24 MFMAs per wave) while the other issues the five LDS-DMAs of its step t + 2 and waits,
// counted, for its step t.  A K-step's DMA instructions carry the step in their SCALAR offset (tap offset + 64 bytes per
// 32-channel chunk); the per-lane part is a constant of the tile with the row mask folded in as an out-of-range offset.
// At the end the groups exchange halves of their accumulators through the dead ring (a fixed order: own + other) and each
// finishes half of the tile: bias, BatchNorm statistics (DPP row sums, per-wave LDS blocks, one fixed-point add per
// channel and workgroup), 16-byte stores after the cross-lane swap of common.h.
// 128 x 192 divides every shape it serves into whole rounds of the 256 CUs (256 / 512 / 1024 / 1280 tiles); LDS: 6 x 20 KB.
#include <stdlib.h>
#include "c3args.h"

namespace aau {

struct GPArgs {
    aau_conv_desc d;
    const unsigned short* src;
    const unsigned short* wpk;
    unsigned short* dst;
    const float* bias;
    float* stats;
    int M, nk, cpt;              // output pixels, K-steps (taps x chunks), chunks per tap (Cpad / 32)
    int ntn;                     // channel tiles (Cout / 192)
    unsigned src_bytes, wpk_bytes;
    unsigned tapoff[4];          // byte offset of tap t's source pixel relative to tap 0's
    int taps2;                   // 1: 2x2 / stride-2 gather (ConvTranspose input gradient), 0: 1x1
};

constexpr int GP_BM = 128, GP_BN = 192, GP_BK = 32;
constexpr int GP_AB = GP_BM * GP_BK * 2, GP_BB = GP_BN * GP_BK * 2, GP_SLOT = GP_AB + GP_BB;   // 8 + 12 = 20 KB
constexpr int GP_NS = 3;                                 // ring slots per group
constexpr int GP_STAT_OFF = 2 * 12 * 256 * 16;           // behind the 96 KB exchange area

__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GPArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];   // [2 groups][3 slots][A | B]
    constexpr unsigned OOB = 0x80000000u;
    constexpr int MI = 4, NI = 6;
    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = wave >> 2, wl = wave & 3;
    const int wm = wl >> 1, wn = wl & 1;
    const int gtid = tid & 255;
    unsigned char* const ring = gp_smem + grp * GP_NS * GP_SLOT;

    // tile of this workgroup: channel tile fastest, XCD-aware bijective remap (the workgroups of an XCD share pixel rows)
    int bid = (int)blockIdx.x;
    {
        const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tn = bid % a.ntn, tm = bid / a.ntn;
    const int m0 = tm * GP_BM, n0 = tn * GP_BN;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, a.wpk_bytes, 0x00020000);

    // ---- fill roles of this thread inside its group: 2 pixel pieces + 3 weight pieces of 16 bytes per K-step ----
    unsigned va[2], vw[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = i * 256 + gtid;
        const int row = p >> 2, lc = swz32(row, p & 3);
        const int m = m0 + row;
        unsigned off;
        if (a.taps2) {       // output pixel (n, y, x) of the coarse grid reads (2y + dy, 2x + dx) of the fine one
            const unsigned x = (unsigned)m % (unsigned)d.Wo, t = (unsigned)m / (unsigned)d.Wo;
            const unsigned y = t % (unsigned)d.Ho, n = t / (unsigned)d.Ho;
            off = (unsigned)((((n * d.H + 2 * y) * d.W + 2 * x) * d.src_pitch + lc * 8) * 2);
        } else {
            off = (unsigned)((m * d.src_pitch + lc * 8) * 2);
        }
        va[i] = m < a.M ? off : OOB;
    }
    const int Kp = d.KH * d.KW * d.Cpad;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = i * 256 + gtid;
        const int row = p >> 2, lc = swz32(row, p & 3);
        vw[i] = (n0 + row) < d.Cout ? (unsigned)(((n0 + row) * Kp + lc * 8) * 2) : OOB;
    }
    // issue iterator of this group over the K-steps grp, grp + 2, ...
    int ik = grp, itap = 0, ichunk = grp;
    while (ichunk >= a.cpt) { ichunk -= a.cpt; ++itap; }
    auto issue = [&](int slot) {
        unsigned char* sA = ring + slot * GP_SLOT;
        unsigned char* sB = sA + GP_AB;
        const bool live = ik < a.nk;                                     // past the end: zeros (keeps the vmcnt counts constant)
        const unsigned soA = live ? a.tapoff[itap & 3] + (unsigned)(ichunk * 64) : 0u;
        const unsigned soW = live ? (unsigned)(ik * 64) : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + (i * 4 + wl) * 1024), 16, (int)(live ? va[i] : OOB), (int)soA, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(sB + (i * 4 + wl) * 1024), 16, (int)(live ? vw[i] : OOB), (int)soW, 0, 0);
        ik += 2; ichunk += 2;
        if (ichunk >= a.cpt) { ichunk -= a.cpt; ++itap; if (ichunk >= a.cpt) { ichunk -= a.cpt; ++itap; } }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    int aoff[MI], boff[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * 64 + mi * 16 + fr;
        aoff[mi] = row * 64 + swz32(row, fk) * 16;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * 96 + ni * 16 + fr;
        boff[ni] = GP_AB + row * 64 + swz32(row, fk) * 16;
    }
    auto compute = [&](int slot) {
        const unsigned char* base = ring + slot * GP_SLOT;
        bf16x8 wf[NI], af[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8*)(base + boff[ni]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8*)(base + aoff[mi]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = AAU_MFMA16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    };

    // ---- ticks (wgradL.hip): between b1 and b2 group 0 multiplies its step t while group 1 issues its step t + 2 and waits
    // for its step t; behind b2 they swap.  RAW: a group's waves wait (all but their youngest 5 / 10 DMAs landed) in front
    // of the barrier that opens their multiply half.  WAR: a DMA goes into the slot whose reads ended a whole tick earlier. ----
    const int ticks = (a.nk + 1) >> 1;
    if (grp == 0) {
        issue(0);
        issue(1);
        int slot = 0, islot = 2;
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        for (int t = 0; t < ticks; ++t) {
            __builtin_amdgcn_s_barrier();               // b1
            compute(slot);
            __builtin_amdgcn_s_barrier();               // b2
            issue(islot);
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            slot = slot == 2 ? 0 : slot + 1;
            islot = islot == 2 ? 0 : islot + 1;
        }
    } else {
        issue(0);
        issue(1);
        int slot = 0, islot = 2;
        for (int t = 0; t < ticks; ++t) {
            __builtin_amdgcn_s_barrier();               // b1
            issue(islot);                               // step t + 2, into the slot of step t - 1 (read before b1)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // step t has landed; t + 1 and t + 2 stay in flight
            __builtin_amdgcn_s_barrier();               // b2
            compute(slot);
            slot = slot == 2 ? 0 : slot + 1;
            islot = islot == 2 ? 0 : islot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- exchange: group g finishes the pixel tiles mi = 2g, 2g + 1 of every wave tile; it hands the other two to the
    // other group through the dead ring ([giver][ni * 2 + j][thread of the group] float4, 2 x 48 KB) ----
    f32x4* xch = (f32x4*)gp_smem;
    __syncthreads();
    f32x4 fin[NI][2];
    if (grp == 0) {          // (two branches: the accumulator indices stay compile-time constants)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) xch[(ni * 2 + j) * 256 + gtid] = acc[ni][2 + j];
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) xch[(12 + ni * 2 + j) * 256 + gtid] = acc[ni][j];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) fin[ni][j] = acc[ni][j] + xch[(12 + ni * 2 + j) * 256 + gtid];
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) fin[ni][j] = xch[(ni * 2 + j) * 256 + gtid] + acc[ni][2 + j];     // group 0's share first
    }

    // ---- epilogue of this group's half: pixels m0 + wm * 64 + (2 grp + j) * 16 + fr ----
    const bool want_stats = a.stats != nullptr;
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, 0x7fffffff, 0x00020000);
    const int Co = d.shuffle2x2 ? d.Cout >> 2 : d.Cout;
    // after the swap this lane owns channels qw .. qw + 7 of pixel tile j = fk & 1, row fr
    const int mrow = m0 + wm * 64 + (2 * grp + (fk & 1)) * 16 + fr;
    unsigned pix_off;           // byte offset of this lane's destination pixel (channel 0)
    if (d.shuffle2x2) {
        const int pos = n0 / Co;                                          // the 192-channel tile lies inside one position
        const unsigned x = (unsigned)mrow % (unsigned)d.W, t = (unsigned)mrow / (unsigned)d.W;
        const unsigned y = t % (unsigned)d.H, n = t / (unsigned)d.H;
        pix_off = (unsigned)((((n * 2 * d.H + 2 * y + (pos >> 1)) * (2 * d.W) + 2 * x + (pos & 1)) * d.dst_pitch) * 2);
    } else {
        pix_off = (unsigned)(mrow * d.dst_pitch * 2);
    }
    const bool prow_ok = mrow < a.M;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int q = n0 + wn * 96 + ni * 16 + 4 * fk;                    // MFMA layout: channels q .. q + 3 of pixels (j, fr)
        float va_[4], vb_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { va_[r] = fin[ni][0][r]; vb_[r] = fin[ni][1][r]; }
        if (q < d.Cout) {
            if (want_stats) {
                const bool okA = m0 + wm * 64 + (2 * grp) * 16 + fr < a.M, okB = m0 + wm * 64 + (2 * grp + 1) * 16 + fr < a.M;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (okA) { s1[ni][r] += va_[r]; s2[ni][r] += va_[r] * va_[r]; }
                    if (okB) { s1[ni][r] += vb_[r]; s2[ni][r] += vb_[r] * vb_[r]; }
                }
            }
            if (a.bias) {
                const int qb = d.shuffle2x2 ? q % Co : q;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float b = a.bias[qb + r]; va_[r] += b; vb_[r] += b; }
            }
        }
        float w[8];
        swap_pair8(va_, vb_, w);                                         // every lane takes part
        const int qw = n0 + wn * 96 + ni * 16 + 8 * (fk >> 1);
        const int cq = d.shuffle2x2 ? qw % Co : qw;
        const unsigned voff = (prow_ok && qw < d.Cout) ? pix_off + (unsigned)(cq * 2) : OOB;
        store_b128_soff(pack8(w), rsD, voff, 0u);
    }
    if (want_stats) {
        float* sst = (float*)(gp_smem + GP_STAT_OFF);                    // [8 waves][2][192]
        float* mine = sst + wave * 2 * GP_BN;
        for (int i = lane; i < 2 * GP_BN; i += 64) mine[i] = 0.f;
        // (each wave writes its own block: no barrier needed in between)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = row16_sum(s1[ni][r]), x2 = row16_sum(s2[ni][r]);
                if (fr == 0) {
                    mine[wn * 96 + ni * 16 + 4 * fk + r] = x1;
                    mine[GP_BN + wn * 96 + ni * 16 + 4 * fk + r] = x2;
                }
            }
        }
        __syncthreads();
        stats_publish(sst, 8, GP_BN, tid, n0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 1 when the ping-pong GEMM serves this problem: 1x1 or the 2x2 / stride-2 gather, whole 128 x 192 tiles, at least a
// workgroup per CU, a K loop long enough to pay for the exchange at its end
bool gemm_pp_applicable(const aau_conv_desc* d, const void* dst, const float* scale, const float* shift) {
    if (getenv("AAU_NO_GEMM_PP") || scale || shift || d->accumulate || d->relu || d->src_split_c > 0 || d->dst_split_c > 0) return false;
    const bool lin = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
    const bool t2 = d->KH == 2 && d->KW == 2 && d->stride == 2 && d->pad == 0 && d->dil == 1 && d->H == 2 * d->Ho && d->W == 2 * d->Wo && !d->shuffle2x2;
    if (!lin && !t2) return false;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    if (M % GP_BM || d->Cout % GP_BN || d->Cpad % 32 || d->Cin != d->Cpad || d->dst_pitch % 8 || ((uintptr_t)dst & 15)) return false;
    if (d->shuffle2x2 && ((d->Cout >> 2) % GP_BN || !lin)) return false;
    const int nk = d->KH * d->KW * (d->Cpad / 32);
    const int64_t tiles = M / GP_BM * (d->Cout / GP_BN);
    const int64_t dst_bytes = M * (d->shuffle2x2 ? 4 : 1) * (int64_t)d->dst_pitch * 2;
    return nk >= 12 && tiles >= 224 && dst_bytes < 0x7fffffff;
}

int gemm_pp_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                   float* stats, unsigned src_bytes, unsigned wpk_bytes, hipStream_t s) {
    GPArgs a;
    a.d = *d;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = bias; a.stats = stats;
    a.M = d->N * d->Ho * d->Wo;
    a.cpt = d->Cpad / 32;
    a.nk = d->KH * d->KW * a.cpt;
    a.ntn = d->Cout / GP_BN;
    a.src_bytes = src_bytes; a.wpk_bytes = wpk_bytes;
    a.taps2 = d->KH == 2 ? 1 : 0;
    for (int t = 0; t < 4; ++t) a.tapoff[t] = a.taps2 ? (unsigned)((((t >> 1) * d->W + (t & 1)) * d->src_pitch) * 2) : 0u;
    const int64_t grid = (int64_t)(a.M / GP_BM) * a.ntn;
    prof_tag(a.taps2 ? "gemm_pp<128,192> taps" : "gemm_pp<128,192>");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    constexpr int lds = 2 * GP_NS * GP_SLOT + 8 * 2 * GP_BN * 4;       // ring (120 KB; the exchange reuses 96 KB of it) + statistics
    static_assert(GP_STAT_OFF + 8 * 2 * GP_BN * 4 <= lds, "statistics blocks behind the exchange area");
    hipLaunchKernelGGL(gemm_pp_kernel, dim3((unsigned)grid), dim3(512), lds, s, a);
    return check_launch("aau_conv_igemm(ping-pong GEMM)");
}

}  // namespace aau

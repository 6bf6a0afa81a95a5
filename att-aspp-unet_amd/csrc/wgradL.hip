// Grouped big-tile weight gradient on MFMA for gfx950: the ASPP bridge (pipeline:67-83) in ONE launch.
//
//   dw_p[q][tap][c] += sum_m dz_p[m][q] * x_p[gather_p(m, tap)][c]          for every problem p of the group
//
// Why a second weight-gradient kernel: at the bridge (M = 8 x 32 x 32 = 8192 pixels, 384 -> 768 channels, three
// dilated 3x3 branches + 1x1 + 3840 -> 768 projection) the generic kernel (wgrad.hip: 96 x 96 tiles, 18 MFMAs per
// wave between two vmcnt(0) barriers, 8-way split-K with 85 MB of partial slabs per branch) ran at 17-19 % of the
// MFMA peak.  Here
//   * a TILE is 192(q) x 192(c) outputs of ONE tap over the FULL pixel range (336 tiles at base_c 48 with the deepest
//     ConvTranspose2d riding along): no split-K, no slabs, no reduce pass, results bitwise reproducible;
//   * K-steps are 32-pixel row segments: a step whose source row (y + dy) lies outside the image is skipped as a
//     whole (dilation 18 at 32 x 32: 56 % of the steps of the off-centre tap rows), so tiles differ in length;
//   * both operands stay [pixel][channel] in LDS as the LDS-DMA delivers them and are read transposed
//     (ds_read_b64_tr_b16); rows are 384 B, the 32-B granule is XOR-ed with bits 1-2 of the row on the SOURCE side of
//     the DMA and on the read side: the 8 rows of a 32-lane half hit 8 different 32-B bank windows (conflict free).
// Rounds 2-3 ran one four-wave workgroup per tile, two per CU.  A tile's K-steps are a serial chain (counted wait ->
// barrier -> six LDS-DMA issues per wave -> 24 transposed reads -> 36 MFMAs) that only a second chain on the CU
// overlaps: one tile alone took 0.74-0.9 us per step, two co-resident ones 0.98-1.1 us per step pair -- but then the
// scheduling unit is HALF a CU, and 336 tiles of unequal length cannot be dealt evenly to 512 half-CUs that all start at
// once: 80 CUs got two tiles, 176 one, and the launch (253-284 us) ran at the pace of the former.  Round 4
// (wgradL_pp_kernel below): ONE persistent eight-wave workgroup per CU whose two four-wave groups work on the same
// tile half a step apart, tiles pulled from per-XCD work queues longest first: 201-204 us for the same launch.
// Measured on the way (profiles/NOTES.md): an eight-wave workgroup in lockstep (48 x 96 per wave, 6-slot ring) is no
// faster per tile than four waves; with the multiply halves removed a step still cost 0.48 us in address arithmetic
// and branches around the DMA issues (now scalar offsets: 0.37 us, the LDS-DMA issue itself); the order of the 24
// reads (first MFMA after 4 instead of 18) and a single chip-wide queue change nothing.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace aau {

constexpr int WL_MAXP = 8;
constexpr int WL_T = 192;              // tile edge (channels)
constexpr int WL_ROWB = WL_T * 2;      // LDS row bytes
constexpr int WL_TILEB = 32 * WL_ROWB; // one operand tile of a K-step: 12 KiB
constexpr int WL_STAGEB = 2 * WL_TILEB;

struct WLProb {
    const unsigned short* src;
    const unsigned short* dz;
    float* dw;
    int H, W, Ho, Wo, Cin, Cout, src_pitch, dst_pitch, KW, T, stride, pad, dil;
    int M, ksteps;             // output pixels, 32-pixel K-steps
    int ntq, ntc;
    int item_begin;            // first tile of this problem in the grid
    int linear;                // 1: 1x1 / stride 1 / pad 0 (source pixel == output pixel)
    unsigned src_bytes, dz_bytes;
};

constexpr int WL_MAXITEMS = 1024;      // tiles of one grouped launch (the order table travels in the kernel arguments)
constexpr int WL_QSTRIDE = 16;         // ints between two queue heads (a 64-byte line each)

struct WLArgs {
    WLProb p[WL_MAXP];
    int nprob, nitems;
    int flags;     // timing experiments only: 1 = no LDS-DMA, 2 = no LDS reads / MFMAs, 4 = no epilogue
    int* queue;    // [8][WL_QSTRIDE]: head of the queue of XCD x at queue[x * WL_QSTRIDE]; ZERO when the launch starts
    int bin_begin[9];                      // queue x holds order[bin_begin[x] .. bin_begin[x + 1])
    unsigned short order[WL_MAXITEMS];     // tile ids, every queue longest tile first
};

// byte offset of (row, channel ch [multiple of 4]) inside a [32][192] bf16 tile
__device__ __forceinline__ int wl_off(int row, int ch) {
    return row * WL_ROWB + ((((ch >> 4) ^ ((row >> 1) & 3))) << 5) + (ch & 15) * 2;
}

#define WL_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define WL_FRAG(lo, hi) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))

// ------------------------------------------------------------------------------------------------------------------
// Ping-pong form: ONE persistent workgroup per CU, eight waves in two four-wave groups that work on the SAME tile.
// Group g takes the tile's K-steps g, g + 2, ... with its own 3-slot ring and its own 96 x 96 accumulators per wave
// (the layout of the four-wave form), and the two groups run HALF A STEP APART: between the two barriers of a tick one
// group issues its LDS-DMA and reads its fragments (the "load" half) while the other one multiplies (36 MFMAs per wave),
// then they swap.  Why: a tile's step is a serial chain (wait -> barrier -> DMA issue -> transposed reads -> MFMAs) that
// only a second, independent chain on the CU overlaps -- the four-wave form gets that from a second WORKGROUP, but then
// a tile is the scheduling unit of HALF a CU and 336 tiles of unequal length cannot be dealt evenly to 512 half-CUs
// that all start at once (80 CUs got two tiles, 176 one: the launch ran at the pace of the former).  Here the unit is a
// whole CU, tiles outnumber the 256 workgroups, and the queues (longest first) balance them.  At the end of a tile group
// 1's accumulators are added to group 0's through the staging area of the epilogue, in a fixed order.
__global__ __launch_bounds__(512, 2) void wgradL_pp_kernel(const WLArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_smem[];     // [2 groups][3 slots] | 16 bytes
    constexpr int NS = 3, NPW = 6;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = wave >> 2, wl = wave & 3;
    const int wq = wl >> 1, wc = wl & 1;
    const int gtid = tid & 255;
    unsigned char* const ring = wl_smem + grp * NS * WL_STAGEB;
    int* const deq = (int*)(wl_smem + 2 * NS * WL_STAGEB);
    const int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    int probe = 0;

  for (;;) {
    if (tid == 0) {
        int item = -1;
        while (probe < 8) {
            const int b = (xcc + probe) & 7;
            const int n = a.bin_begin[b + 1] - a.bin_begin[b];
            if (n > 0) {
                const int i = __hip_atomic_fetch_add(a.queue + b * WL_QSTRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i < n) { item = a.order[a.bin_begin[b] + i]; break; }
            }
            ++probe;
        }
        *deq = item;
    }
    __syncthreads();
    const int bid = __builtin_amdgcn_readfirstlane(*deq);
    __syncthreads();
    if (bid < 0) break;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WL_MAXP; ++i)
        if (i < a.nprob && a.p[i].item_begin <= bid) pi = i;
    const WLProb P = a.p[pi];
    int local = bid - P.item_begin;
    const int tc = local % P.ntc; local /= P.ntc;
    const int tap = local % P.T;
    const int tq = local / P.T;
    const int q0 = tq * WL_T, c0 = tc * WL_T;
    const int dy = (tap / P.KW) * P.dil - P.pad, dx = (tap % P.KW) * P.dil - P.pad;

    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)P.dz, 0, P.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)P.src, 0, P.src_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int segs = P.linear ? (P.M + 31) >> 5 : P.Wo >> 5;
    int ylo = 0, yhi = 1, nimg = 1;
    if (!P.linear) {
        ylo = dy < 0 ? (-dy + P.stride - 1) / P.stride : 0;
        yhi = P.H - 1 - dy < 0 ? 0 : (P.H - 1 - dy) / P.stride + 1;
        if (yhi > P.Ho) yhi = P.Ho;
        nimg = P.M / (P.Ho * P.Wo);
    }
    const int total = yhi > ylo ? nimg * (yhi - ylo) * segs : 0;      // valid K-steps of this tile
    const int ticks = (total + 1) >> 1;                               // steps of either group (group 1's last may be empty)

    // the three 16-B pieces per operand this thread stages every K-step of its group (the four-wave form's roles)
    unsigned ybase[3], xbase[3];
    int prow[3], xcol[3], pch[3];
    bool yok[3], xok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = (i * 4 + wl) * 64 + lane;
        const int row = p / 24, hg = p - row * 24;
        const int ch = ((((hg >> 1) ^ ((row >> 1) & 3)) << 1) | (hg & 1)) * 8;
        prow[i] = row; pch[i] = ch;
        yok[i] = q0 + ch < P.Cout;
        xok[i] = c0 + ch < P.Cin;
        ybase[i] = (unsigned)((row * P.dst_pitch + q0 + ch) * 2);
        xcol[i] = row * P.stride + dx;
        xbase[i] = (unsigned)(((P.linear ? row : xcol[i]) * P.src_pitch + c0 + ch) * 2);
        if (!P.linear && segs == 1) xok[i] = xok[i] && (unsigned)xcol[i] < (unsigned)P.W;
    }
    const bool ragged = (P.M & 31) != 0;
    const bool multiseg = !P.linear && segs > 1;
    // Fast issue path (every step of the tile is a whole 32-pixel segment of a single-segment row: the bridge's shapes):
    // the per-lane part of a piece's address -- with every mask folded in as an out-of-range offset -- is a constant of
    // the tile, the step only changes the SCALAR offset of the instruction, so a K-step issues its six LDS-DMAs with no
    // vector arithmetic (the general path below spent ~150 cycles per piece on predicates and branches: with the
    // multiply halves removed a step still took 0.48 us).  The x resource starts `shiftb` bytes in front of the tensor
    // so that the scalar part stays non-negative for taps that reach to the left (dx < 0).
    const bool fast = !ragged && !multiseg;
    const unsigned shiftb = (unsigned)(P.pad * P.src_pitch * 2);
    const __amdgpu_buffer_rsrc_t rsXs = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)P.src - shiftb), 0,
                                                                          P.src_bytes + shiftb, 0x00020000);
    unsigned vy[3], vx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vy[i] = yok[i] ? ybase[i] : OOB;
        vx[i] = xok[i] ? (unsigned)(((P.linear ? prow[i] : prow[i] * P.stride) * P.src_pitch + c0 + pch[i]) * 2) : OOB;
    }

    // issue iterator of this group over the tile's steps grp, grp + 2, ...
    int in_ = 0, iy = ylo, ix0 = 0, istep = 0;
    auto adv = [&]() {
        ++istep;
        ix0 += 32;
        if (ix0 >= (P.linear ? P.M : P.Wo)) { ix0 = 0; if (++iy == yhi) { iy = ylo; ++in_; } }
    };
    if (grp) adv();
    auto issue = [&](int slot) {
        unsigned char* sY = ring + slot * WL_STAGEB;
        unsigned char* sX = sY + WL_TILEB;
        const bool live = istep < total && !(a.flags & 1);      // past the end: zero rows (keeps the vmcnt count constant)
        const int m0 = P.linear ? ix0 : (in_ * P.Ho + iy) * P.Wo + ix0;
        if (fast) {
            const unsigned sy = (unsigned)m0 * (unsigned)(P.dst_pitch * 2);
            const unsigned sx = P.linear ? (unsigned)m0 * (unsigned)(P.src_pitch * 2)
                                         : (unsigned)(((in_ * P.H + iy * P.stride + dy) * P.W + ix0 * P.stride + dx) * P.src_pitch * 2) + shiftb;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sY + (i * 4 + wl) * 1024), 16, (int)(live ? vy[i] : OOB), (int)sy, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXs, LDS_PTR(sX + (i * 4 + wl) * 1024), 16, (int)(live ? vx[i] : OOB), (int)sx, 0, 0);
            adv(); adv();
            return;
        }
        const unsigned yoff = (unsigned)m0 * (unsigned)(P.dst_pitch * 2);
        const int xs0 = ix0 * P.stride;
        const unsigned xoff = P.linear ? (unsigned)m0 * (unsigned)(P.src_pitch * 2)
                                       : (unsigned)(((in_ * P.H + iy * P.stride + dy) * P.W + xs0) * P.src_pitch * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = live && yok[i];
            if (ragged) ok = ok && m0 + prow[i] < P.M;
            const unsigned v = ok ? ybase[i] + yoff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sY + (i * 4 + wl) * 1024), 16, (int)v, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = live && xok[i];
            if (ragged) ok = ok && m0 + prow[i] < P.M;
            if (multiseg) ok = ok && (unsigned)(xs0 + xcol[i]) < (unsigned)P.W;
            const unsigned v = ok ? xbase[i] + xoff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(sX + (i * 4 + wl) * 1024), 16, (int)v, 0, 0, 0);
        }
        adv(); adv();
    };

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g16 = lane >> 4, li = lane & 15;
    const int rrow = 4 * g16 + (li >> 2), cp = (li & 3) * 4;
    int yoffs[6], xoffs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        yoffs[i] = wl_off(rrow, wq * 96 + i * 16 + cp);
        xoffs[i] = WL_TILEB + wl_off(rrow, wc * 96 + i * 16 + cp);
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring;
    // multiply half of a step: the 24 transposed reads of slot `slot` and the 36 MFMAs.  While this group multiplies, the
    // other one only issues DMAs, so nothing else feeds the matrix pipe during the reads' latency: the reads are issued
    // in "growing square" order (A0 B0 A1 B1 ... A5 B5; LDS returns in order) and the products max(i, j) = k start as
    // soon as A_k and B_k are back -- the first MFMA after 4 of the 24 reads instead of 18.
    auto compute = [&](int slot) {
        if (a.flags & 2) return;
        const unsigned base = lds0 + slot * WL_STAGEB;
        u32x2 alo[6], ahi[6], blo[6], bhi[6];
#define WL_READ(K)                                                                                                        \
        { const unsigned ay = base + yoffs[K], ax = base + xoffs[K];                                                      \
          WL_TR(alo[K], ay, 0); WL_TR(ahi[K], ay, 6144); WL_TR(blo[K], ax, 0); WL_TR(bhi[K], ax, 6144); }
        bf16x8 af[6], bf[6];
#define WL_STAGE(K, CNT)                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(alo[K]), "+v"(ahi[K]), "+v"(blo[K]), "+v"(bhi[K]));               \
        af[K] = WL_FRAG(alo[K], ahi[K]);                                                                                  \
        bf[K] = WL_FRAG(blo[K], bhi[K]);                                                                                  \
        _Pragma("unroll") for (int i = 0; i < K; ++i) acc[i][K] = AAU_MFMA16(af[i], bf[K], acc[i][K], 0, 0, 0);          \
        _Pragma("unroll") for (int j = 0; j <= K; ++j) acc[K][j] = AAU_MFMA16(af[K], bf[j], acc[K][j], 0, 0, 0);      \
        __builtin_amdgcn_sched_barrier(0);      /* hipcc would sink every MFMA below the last wait (guide 5.4 rule 18) */
        // (lgkmcnt is a 4-bit field: at most 15 can be named, so the last two read groups follow the first two stages)
        WL_READ(0) WL_READ(1) WL_READ(2) WL_READ(3)
        WL_STAGE(0, 12)
        WL_READ(4)
        WL_STAGE(1, 12)
        WL_READ(5)
        WL_STAGE(2, 12)
        WL_STAGE(3, 8)
        WL_STAGE(4, 4)
        WL_STAGE(5, 0)
#undef WL_READ
#undef WL_STAGE
    };

    // ---- ticks: two barriers each.  Between b1 and b2 group 0 multiplies its step t (transposed reads + 36 MFMAs per
    // wave) while group 1 issues the LDS-DMA of its step t + 2 and waits for its step t; between b2 and the next b1 they
    // swap (group 0 issues step t + 2, waits for step t + 1).  RAW: a group's waves wait (all but their youngest 6 DMAs
    // landed) in front of the barrier that opens their multiply half.  WAR: a DMA goes into the slot whose reads ended at
    // least a whole tick earlier (3 slots per group). ----
    if (ticks > 0) {
        if (grp == 0) {
            issue(0);
            issue(1);
            int slot = 0, islot = 2;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            for (int t = 0; t < ticks; ++t) {
                __builtin_amdgcn_s_barrier();               // b1
                compute(slot);
                __builtin_amdgcn_s_barrier();               // b2
                issue(islot);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
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
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // step t has landed; t + 1 and t + 2 stay in flight
                __builtin_amdgcn_s_barrier();               // b2
                compute(slot);
                slot = slot == 2 ? 0 : slot + 1;
                islot = islot == 2 ? 0 : islot + 1;
            }
        }
    }
    if (a.flags & 4) continue;

    // ---- epilogue: dw[q][tap][c0 ..] += D0 + D1 through LDS (96 q-rows x 192 fp32 per half): group 1 stores its
    // accumulators, group 0 adds its own on top (same lane -> element map: every element is touched by one thread of each
    // group), then all 512 threads copy out 16-B vectors along contiguous 768-B rows ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* stg = (float*)wl_smem;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wq == h && grp == 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(i * 16 + 4 * g16 + r) * WL_T + wc * 96 + j * 16 + li] = acc[i][j][r];
        }
        __syncthreads();
        if (wq == h && grp == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(i * 16 + 4 * g16 + r) * WL_T + wc * 96 + j * 16 + li] += acc[i][j][r];
        }
        __syncthreads();
        for (int v = tid; v < 96 * 48; v += 512) {
            const int row = v / 48, c4 = (v - row * 48) * 4;
            const int q = q0 + h * 96 + row, c = c0 + c4;
            if (q < P.Cout && c < P.Cin) {
                float* d = P.dw + ((int64_t)q * P.T + tap) * P.Cin + c;
                const f32x4 add = *(const f32x4*)(stg + row * WL_T + c4);
                f32x4 old = *(const f32x4*)d;
                old += add;
                *(f32x4*)d = old;
            }
        }
    }
    (void)gtid;
  }   // next tile
}

}  // namespace aau

using namespace aau;

// true when one problem can go into a grouped launch
static bool wl_ok(const aau_conv_desc* d) {
    const bool linear = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
    if (!linear && d->Wo % 32 != 0) return false;
    if (d->Cin % 8 || d->Cout % 8 || d->src_pitch % 8 || d->dst_pitch % 8) return false;
    if (d->KH * d->KW > 16 || d->Cin < 96 || d->Cout < 96) return false;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    return M >= 1024 && M * d->dst_pitch * 2 < 0x7fffffff && (int64_t)d->N * d->H * d->W * d->src_pitch * 2 < 0x7fffffff;
}

// 1 when this problem may be a MEMBER of a grouped launch (the group as a whole is judged by aau_conv_wgrad_group_ok)
extern "C" int aau_conv_wgrad_group_member_ok(const aau_conv_desc* d) {
    return d && !getenv("AAU_NO_WGRAD_GROUP") && wl_ok(d) ? 1 : 0;
}

extern "C" int aau_conv_wgrad_group_ok(const aau_conv_desc* descs, int n) {
    if (!descs || n < 1 || n > WL_MAXP || getenv("AAU_NO_WGRAD_GROUP")) return 0;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!wl_ok(&descs[i])) return 0;
        tiles += (int64_t)((descs[i].Cout + WL_T - 1) / WL_T) * ((descs[i].Cin + WL_T - 1) / WL_T) * descs[i].KH * descs[i].KW;
    }
    // fewer tiles than ~2/3 of the CUs: the split-K kernels fill the chip better; more than the order table holds: not served
    return tiles >= 160 && tiles <= WL_MAXITEMS ? 1 : 0;
}

extern "C" int64_t aau_conv_wgrad_group_queue_bytes(void) { return (int64_t)8 * WL_QSTRIDE * sizeof(int); }

// valid K-steps of tile (problem P, tap): the kernel's own arithmetic
static int wl_tile_steps(const WLProb& P, int tap) {
    if (P.linear) return (P.M + 31) >> 5;
    const int dy = (tap / P.KW) * P.dil - P.pad;
    const int ylo = dy < 0 ? (-dy + P.stride - 1) / P.stride : 0;
    int yhi = P.H - 1 - dy < 0 ? 0 : (P.H - 1 - dy) / P.stride + 1;
    if (yhi > P.Ho) yhi = P.Ho;
    return yhi > ylo ? (P.M / (P.Ho * P.Wo)) * (yhi - ylo) * (P.Wo >> 5) : 0;
}

extern "C" int aau_conv_wgrad_group(const aau_conv_desc* descs, const aau_bf16* const* srcs, const aau_bf16* const* dzs,
                                    float* const* dws, int n, int32_t* queue, int64_t queue_bytes, void* stream) {
    AAU_REQUIRE(descs && srcs && dzs && dws && n >= 1 && n <= WL_MAXP, "aau_conv_wgrad_group: bad args (n=%d, at most %d problems)", n, WL_MAXP);
    AAU_REQUIRE(queue && queue_bytes >= aau_conv_wgrad_group_queue_bytes() && ((uintptr_t)queue & 3) == 0,
                "aau_conv_wgrad_group: the work-queue heads need %lld zeroed bytes (aau_conv_wgrad_group_queue_bytes), got %lld",
                (long long)aau_conv_wgrad_group_queue_bytes(), (long long)queue_bytes);
    WLArgs a;
    a.nprob = n;
    a.queue = queue;
    int items = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const aau_conv_desc* d = &descs[i];
        AAU_REQUIRE(wl_ok(d), "aau_conv_wgrad_group: problem %d is outside the kernel's range (aau_conv_wgrad_group_ok)", i);
        AAU_REQUIRE(srcs[i] && dzs[i] && dws[i], "aau_conv_wgrad_group: null pointer in problem %d", i);
        AAU_REQUIRE(((uintptr_t)srcs[i] & 15) == 0 && ((uintptr_t)dzs[i] & 15) == 0, "aau_conv_wgrad_group: 16-byte alignment");
        WLProb& P = a.p[i];
        P.src = srcs[i]; P.dz = dzs[i]; P.dw = dws[i];
        P.H = d->H; P.W = d->W; P.Ho = d->Ho; P.Wo = d->Wo; P.Cin = d->Cin; P.Cout = d->Cout;
        P.src_pitch = d->src_pitch; P.dst_pitch = d->dst_pitch; P.KW = d->KW; P.T = d->KH * d->KW;
        P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
        P.M = d->N * d->Ho * d->Wo;
        P.ksteps = (P.M + 31) / 32;
        P.ntq = (d->Cout + WL_T - 1) / WL_T;
        P.ntc = (d->Cin + WL_T - 1) / WL_T;
        P.item_begin = items;
        P.linear = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
        P.src_bytes = (unsigned)((((int64_t)d->N * d->H * d->W - 1) * d->src_pitch + d->Cin) * 2);
        P.dz_bytes = (unsigned)((((int64_t)P.M - 1) * d->dst_pitch + d->Cout) * 2);
        items += P.ntq * P.ntc * P.T;
        flops += 2.0 * P.M * (double)d->Cout * d->Cin * P.T;
        bytes += 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)P.M * d->Cout) + 4.0 * d->Cout * (double)P.T * d->Cin;
    }
    a.nitems = items;
    AAU_REQUIRE(items <= WL_MAXITEMS, "aau_conv_wgrad_group: %d tiles, at most %d (aau_conv_wgrad_group_ok)", items, WL_MAXITEMS);
    a.flags = getenv("AAU_WL_FLAGS") ? atoi(getenv("AAU_WL_FLAGS")) : 0;
    // ---- work queues: (problem, q-tile) groups dealt to the 8 XCD queues longest group first onto the lightest queue;
    // inside a queue longest tile first (stable: tiles of a group with equal length stay together) ----
    struct Item { unsigned short id; int steps; int group; };
    struct Group { int first, count; int64_t steps; };
    Item it[WL_MAXITEMS];
    Group gr[WL_MAXITEMS];
    int ni = 0, ng = 0;
    for (int i = 0; i < n; ++i) {
        const WLProb& P = a.p[i];
        for (int tq = 0; tq < P.ntq; ++tq) {
            Group g{ni, 0, 0};
            for (int tap = 0; tap < P.T; ++tap) {
                const int steps = wl_tile_steps(P, tap);
                if (steps == 0) continue;                      // the tap never meets the image: nothing to add
                for (int tc = 0; tc < P.ntc; ++tc) {
                    it[ni++] = Item{(unsigned short)(P.item_begin + (tq * P.T + tap) * P.ntc + tc), steps, ng};
                    g.steps += steps; ++g.count;
                }
            }
            if (g.count) gr[ng++] = g;
        }
    }
    if (ni == 0) return AAU_OK;
    int gorder[WL_MAXITEMS];
    for (int i = 0; i < ng; ++i) gorder[i] = i;
    std::stable_sort(gorder, gorder + ng, [&](int x, int y) { return gr[x].steps > gr[y].steps; });
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int bin_of[WL_MAXITEMS];
    for (int k = 0; k < ng; ++k) {
        int b = 0;
        for (int x = 1; x < 8; ++x) if (load[x] < load[b]) b = x;
        bin_of[gorder[k]] = b;
        load[b] += gr[gorder[k]].steps;
    }
    int pos = 0;
    for (int b = 0; b < 8; ++b) {
        a.bin_begin[b] = pos;
        int idx[WL_MAXITEMS], m = 0;
        for (int i = 0; i < ni; ++i) if (bin_of[it[i].group] == b) idx[m++] = i;
        std::stable_sort(idx, idx + m, [&](int x, int y) { return it[x].steps > it[y].steps; });
        for (int i = 0; i < m; ++i) a.order[pos++] = it[idx[i]].id;
    }
    a.bin_begin[8] = pos;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(1, flops, s);
    prof_tag("wgradL<192,192> grouped", bytes);
    constexpr size_t lds = (size_t)6 * WL_STAGEB + 16;       // [2 groups][3 slots] | the dequeued tile id
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)wgradL_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    // one persistent workgroup per CU; fewer tiles than CUs: one workgroup per tile
    hipLaunchKernelGGL(wgradL_pp_kernel, dim3((unsigned)(ni < 256 ? ni : 256)), dim3(512), lds, s, a);
    return check_launch("aau_conv_wgrad_group");
}

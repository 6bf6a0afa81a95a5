// Grouped big-tile weight gradient on MFMA for gfx950: the ASPP bridge (pipeline:67-83) in ONE launch.
//
//   dw_p[q][tap][c] += sum_m dz_p[m][q] * x_p[gather_p(m, tap)][c]          for every problem p of the group
//
// Why a second weight-gradient kernel: at the bridge (M = 8 x 32 x 32 = 8192 pixels, 384 -> 768 channels, three
// dilated 3x3 branches + 1x1 + 3840 -> 768 projection) the generic kernel (wgrad.hip: 96 x 96 tiles, 18 MFMAs per
// wave between two vmcnt(0) barriers, 8-way split-K with 85 MB of partial slabs per branch) ran at 17-19 % of the
// MFMA peak.  Here
//   * a workgroup owns a 192(q) x 192(c) tile of ONE tap (4 waves x 96 x 96 = 36 accumulator tiles each): 96 FLOP
//     per staged byte instead of 48, 36 MFMAs per wave between barriers;
//   * the five problems are tiled into one grid (304 tiles at base_c 48), so every tile runs the FULL pixel range:
//     no split-K, no slabs, no reduce pass, results bitwise reproducible;
//   * K-steps are 32-pixel row segments: a step whose source row (y + dy) lies outside the image is skipped as a
//     whole (dilation 18 at 32 x 32: 56 % of the steps of the off-centre tap rows);
//   * 3-slot LDS ring filled by buffer_load ... lds with a counted s_waitcnt vmcnt(6) + one raw s_barrier per step
//     (the next step's tiles stay in flight across the barrier), 72 KiB -> two workgroups per CU.  Measured on the
//     way (MI355X, 256 tiles): a 6-slot ring and an extra wave that warms L2 eight steps ahead changed nothing --
//     the step time is the serial sum of LDS-DMA issue (~500 cycles per wave), transposed reads + MFMAs (~850) and
//     loop overhead, which two workgroups per CU overlap;
//   * both operands stay [pixel][channel] in LDS as they arrive and are read transposed (ds_read_b64_tr_b16); rows are
//     384 B, the 32-B granule is XOR-ed with bits 1-2 of the row on the SOURCE side of the LDS-DMA and on the read
//     side, which makes the 8 rows of a 32-lane half hit 8 different 32-B bank windows (conflict free).
// Scheduling (round 4).  A tile's K-steps form a serial chain (counted wait -> barrier -> six LDS-DMA issues -> 24
// transposed reads -> 36 MFMAs): ONE tile alone on a CU runs at 0.74 us per step, two co-resident tiles at 0.98 us per
// step PAIR -- only a second workgroup on the CU overlaps the chain.  Tiles differ in length (dilation 18 at 32 x 32:
// 44 % of the rows of an off-centre tap), and a plain grid of 336 tiles put two workgroups on 80 CUs and one on 176:
// the launch ran at the pace of the doubly occupied CUs while the others idled after 0.4-1 tile.  Now the kernel is
// PERSISTENT: 512 workgroups (two per CU, the LDS ring admits exactly two) pull tiles from work queues, longest first,
// so every CU runs two chains until the queues are empty and the tail is made of the shortest tiles.  One queue per
// XCD (the workgroup reads HW_REG_XCC_ID): the host deals (problem, q-tile) groups -- the 18 tiles that stream the same
// dz columns -- to the eight queues by greedy longest-first bin packing, so a group's operand slices stay in one L2;
// a workgroup whose queue is empty takes from the others'.  Placement only steers speed: any workgroup may run any
// tile, every tile is run exactly once (one returning agent-scope atomic add per dequeue), tiles write disjoint blocks
// of dw, so the result does not depend on who ran what.
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

// WL_NS = LDS ring slots (3: 72 KiB, two workgroups per CU).
template <int WL_NS>
__global__ __launch_bounds__(256, 2) void wgradL_kernel(const WLArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_smem[];     // ring | 16 bytes: the dequeued tile id
    static_assert(WL_NS == 3, "the epilogue stages 96 x 192 fp32 through the 72-KiB ring");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wq = wave >> 1, wc = wave & 1;
    int* const deq = (int*)(wl_smem + WL_NS * WL_STAGEB);
    // HW_REG_XCC_ID (id 20), bits 3:0: which XCD this workgroup runs on -- picks the queue to start with, nothing else
    const int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    int probe = 0;                 // queues this workgroup has found empty (thread 0's copy is the one that counts)

  for (;;) {
    // ---- next tile: own XCD's queue first, then the others' ----
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
    __syncthreads();               // also: every wave has finished reading the previous tile's staging area
    const int bid = __builtin_amdgcn_readfirstlane(*deq);
    __syncthreads();
    if (bid < 0) break;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WL_MAXP; ++i)
        if (i < a.nprob && a.p[i].item_begin <= bid) pi = i;
    const WLProb P = a.p[pi];      // by value: one batch of scalar loads, then SGPRs (a reference re-reads the kernarg
                                   // segment inside the K loop: 0.3 us per step)
    int local = bid - P.item_begin;
    const int tc = local % P.ntc; local /= P.ntc;
    const int tap = local % P.T;
    const int tq = local / P.T;
    const int q0 = tq * WL_T, c0 = tc * WL_T;
    const int dy = (tap / P.KW) * P.dil - P.pad, dx = (tap % P.KW) * P.dil - P.pad;

    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)P.dz, 0, P.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)P.src, 0, P.src_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- K-steps = 32-pixel segments of output rows; only rows whose source row y*stride+dy is inside the image ----
    // (linear problems: one "row" of M pixels per launch)
    const int segs = P.linear ? (P.M + 31) >> 5 : P.Wo >> 5;          // steps per row
    int ylo = 0, yhi = 1, nimg = 1;
    if (!P.linear) {
        ylo = dy < 0 ? (-dy + P.stride - 1) / P.stride : 0;
        yhi = P.H - 1 - dy < 0 ? 0 : (P.H - 1 - dy) / P.stride + 1;
        if (yhi > P.Ho) yhi = P.Ho;
        nimg = P.M / (P.Ho * P.Wo);
    }
    const int total = yhi > ylo ? nimg * (yhi - ylo) * segs : 0;      // valid K-steps of this tile

    // ---- the three 16-B pieces per operand this thread stages every K-step ----
    // piece p = (i*4 + wave)*64 + lane -> LDS offset p*16 (lane-linear); row = p / 24, physical half-granule p % 24
    unsigned ybase[3], xbase[3];   // byte offset of (row, channel) relative to the step's first pixel (may wrap below 0)
    int prow[3], xcol[3];          // xcol: source column of the row relative to x0*stride
    bool yok[3], xok[3];           // channel inside the tensor (tile tails); xok also: column inside the image when
                                   // the row has a single segment (then it does not depend on the step)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = (i * 4 + wave) * 64 + lane;
        const int row = p / 24, hg = p - row * 24;
        const int ch = ((((hg >> 1) ^ ((row >> 1) & 3)) << 1) | (hg & 1)) * 8;
        prow[i] = row;
        yok[i] = q0 + ch < P.Cout;
        xok[i] = c0 + ch < P.Cin;
        ybase[i] = (unsigned)((row * P.dst_pitch + q0 + ch) * 2);
        xcol[i] = row * P.stride + dx;
        xbase[i] = (unsigned)(((P.linear ? row : xcol[i]) * P.src_pitch + c0 + ch) * 2);
        if (!P.linear && segs == 1) xok[i] = xok[i] && (unsigned)xcol[i] < (unsigned)P.W;
    }
    const bool ragged = (P.M & 31) != 0;       // only linear problems can end inside a step
    const bool multiseg = !P.linear && segs > 1;

    // issue iterator (n, y, x0): plain counters, no validity tests inside the loop
    int in_ = 0, iy = ylo, ix0 = 0, issued = 0;
    auto issue = [&](int slot) {
        unsigned char* sY = wl_smem + slot * WL_STAGEB;
        unsigned char* sX = sY + WL_TILEB;
        const bool live = issued < total && !(a.flags & 1);     // past the end: zero rows (keeps the vmcnt count constant)
        const int m0 = P.linear ? ix0 : (in_ * P.Ho + iy) * P.Wo + ix0;
        const unsigned yoff = (unsigned)m0 * (unsigned)(P.dst_pitch * 2);
        const int xs0 = ix0 * P.stride;
        const unsigned xoff = P.linear ? (unsigned)m0 * (unsigned)(P.src_pitch * 2)
                                       : (unsigned)(((in_ * P.H + iy * P.stride + dy) * P.W + xs0) * P.src_pitch * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = live && yok[i];
            if (ragged) ok = ok && m0 + prow[i] < P.M;
            const unsigned v = ok ? ybase[i] + yoff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sY + (i * 4 + wave) * 1024), 16, (int)v, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = live && xok[i];
            if (ragged) ok = ok && m0 + prow[i] < P.M;
            if (multiseg) ok = ok && (unsigned)(xs0 + xcol[i]) < (unsigned)P.W;
            const unsigned v = ok ? xbase[i] + xoff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(sX + (i * 4 + wave) * 1024), 16, (int)v, 0, 0, 0);
        }
        ++issued;
        ix0 += 32;
        if (ix0 >= (P.linear ? P.M : P.Wo)) { ix0 = 0; if (++iy == yhi) { iy = ylo; ++in_; } }
    };

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: lane 4*rq+p of a 16-lane group supplies row rq, columns 4p..4p+3; group g16 owns
    // rows 4*g16 .. 4*g16+3 (lo) and 16 + the same (hi):  k = 8*g16 + 4*h + e  <->  pixel 16*h + 4*g16 + e
    const int g16 = lane >> 4, li = lane & 15;
    const int rrow = 4 * g16 + (li >> 2), cp = (li & 3) * 4;
    int yoffs[6], xoffs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        yoffs[i] = wl_off(rrow, wq * 96 + i * 16 + cp);
        xoffs[i] = WL_TILEB + wl_off(rrow, wc * 96 + i * 16 + cp);
    }
    // The transposed reads are inline asm: hipcc puts s_waitcnt vmcnt(0) in front of the ds_read_tr BUILTIN whenever an
    // LDS-DMA is in flight (it cannot tell the ring slots apart), which serialises the whole pipeline; it does not see
    // inside an asm statement, so the counted vmcnt above the barrier is the only wait.  The waits for the reads are
    // explicit and name every destination ("+v"), which keeps the MFMAs below them (guide 5.7, form ii).
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)wl_smem;
#define WL_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define WL_FRAG(lo, hi) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))
    auto compute = [&](int slot) {
        if (a.flags & 2) return;
        const unsigned base = lds0 + slot * WL_STAGEB;
        u32x2 alo[6], ahi[6], blo[6], bhi[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned ad = base + yoffs[i];
            WL_TR(alo[i], ad, 0);
            WL_TR(ahi[i], ad, 6144);          // 16 rows further: 16 * WL_ROWB
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned ad = base + xoffs[j];
            WL_TR(blo[j], ad, 0);
            WL_TR(bhi[j], ad, 6144);
        }
        // LDS returns in order: all but the last 6 reads (blo/bhi[3..5]) are back
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(alo[4]), "+v"(alo[5]), "+v"(ahi[0]),
                       "+v"(ahi[1]), "+v"(ahi[2]), "+v"(ahi[3]), "+v"(ahi[4]), "+v"(ahi[5]), "+v"(blo[0]), "+v"(blo[1]),
                       "+v"(blo[2]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]));
        bf16x8 af[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) af[i] = WL_FRAG(alo[i], ahi[i]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bf16x8 bf = WL_FRAG(blo[j], bhi[j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i][j] = AAU_MFMA16(af[i], bf, acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[3]), "+v"(blo[4]), "+v"(blo[5]), "+v"(bhi[3]), "+v"(bhi[4]), "+v"(bhi[5]));
#pragma unroll
        for (int j = 3; j < 6; ++j) {
            const bf16x8 bf = WL_FRAG(blo[j], bhi[j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i][j] = AAU_MFMA16(af[i], bf, acc[i][j], 0, 0, 0);
        }
    };

    // ---- pipeline: two steps in flight; every step issues 6 LDS-DMA per thread (steps past the end load zero rows), so
    // the wait is one constant: all but the youngest 6 have landed ----
    if (total > 0) {
        issue(0);
        issue(1);
        int slot = 0, islot = 2;
        for (int t = 0; t < total; ++t) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(islot);                       // the slot read in the previous iteration
            compute(slot);
            slot = slot == 2 ? 0 : slot + 1;
            islot = islot == 2 ? 0 : islot + 1;
        }
    }
    if (a.flags & 4) continue;

    // ---- epilogue: dw[q][tap][c0 ..] += D.  The accumulators go through LDS (96 q-rows x 192 fp32 = the whole ring)
    // so that global memory sees 16-B vectors along contiguous 768-B rows instead of 4-B pieces ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* stg = (float*)wl_smem;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wq == h) {
            // acc[i][j][r] = D[q = q0 + wq*96 + i*16 + 4*g16 + r][c = c0 + wc*96 + j*16 + li]
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(i * 16 + 4 * g16 + r) * WL_T + wc * 96 + j * 16 + li] = acc[i][j][r];
        }
        __syncthreads();
        for (int v = tid; v < 96 * 48; v += 256) {
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
    constexpr size_t lds = 3 * WL_STAGEB + 16;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)wgradL_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    // two workgroups per CU (the ring admits exactly two); fewer tiles than that: one workgroup per tile
    const int grid = ni < 512 ? ni : 512;
    hipLaunchKernelGGL((wgradL_kernel<3>), dim3((unsigned)grid), dim3(256), lds, s, a);
    return check_launch("aau_conv_wgrad_group");
}

// Weight gradient of the 3x3 (stride 1, pad 1, dilation 1) convolutions, ROW-REUSE variant (gfx950).
//
//   dw[q][tap][c] += sum_m dz[m][q] * x[m + off(tap)][c]           (pipeline:63 backward, via aau_conv_wgrad)
//
// wgrad3x3.hip gives each wave 3 q tiles x 7 (tap, channel group) column tiles and re-reads, per 32-pixel sub-step,
// 6 dz fragments + 14 x fragments out of LDS for 21 MFMAs: 488 LDS bytes per MFMA, which with two workgroups per CU
// is 640 LDS cycles against 672 matrix cycles per SIMD -- the LDS pipe, not the MFMA, is what the kernel runs at.
//
// Here a wave owns ONE 16-channel group of x and ALL nine taps for it (3 q tiles x 9 column tiles = 27 accumulator
// tiles).  The operand of tap (ty, tx) at sub-step ss (patch rows 2ss, 2ss+1) is the pair of halo rows
// (2ss + ty, 2ss + ty + 1) at column shift tx, so for a fixed tx the three vertical taps of the four sub-steps of an
// 8-row patch are served by the 10 halo rows read ONCE each (a sliding window of 4 rows: 24 operand halves from 10
// reads).  Per patch a wave reads 24 dz + 30 x fragments for 108 MFMAs: 256 LDS bytes per MFMA (-48 %).
//
// Workgroup tile (4 waves):  <QT=3, CJ=4>: 48 q x 64 c   (waves = the four channel groups)
//                            <QT=6, CJ=2>: 96 q x 32 c   (waves = 2 q halves x 2 channel groups)
// both stage 12/24 KB of dz + 22.5/11.25 KB of x halo per patch (2 stages, 2 workgroups per CU) for 432 MFMAs.
// LDS rows stay [pixel][channel] as the LDS-DMA delivers them; operands come out of ds_read_b64_tr_b16 (see
// wgrad3x3.hip for the k <-> pixel map).  128-B / 64-B halo rows would put 8 / 4 of the 16 rows of a transposed read on
// the same banks, so the 32-B channel granule is XOR-swizzled with the pixel index on the SOURCE side of the DMA
// (granule' = j ^ ((pix >> 1) & 3) resp. j ^ ((pix >> 2) & 1)): any 8 consecutive pixels cover all 8 bank groups.
// The swizzle has period 8 in the pixel index, so 8 per-lane base registers + a compile-time row offset address all 30
// (row, shift) fragments without per-read address arithmetic.
// Split-K over patch ranges; partial sums leave in register layout and wg_reduce_kernel<2> adds them in a fixed order.
#include <stdlib.h>
#include "common.h"

namespace aau {

struct W3RArgs {
    aau_conv_desc d;
    const unsigned short* src;   // x   [N][H][W] pitch src_pitch, Cin channels
    const unsigned short* dz;    // dz  [N][H][W] pitch dst_pitch, Cout channels
    float* dw;
    float* ws;                   // split-K slabs [workgroup][27 acc tiles][256 threads][4] (null: fp32 atomics)
    unsigned src_bytes, dz_bytes;
    int npatch;                  // N * (H/8) * (W/16)
    int patches_per_block, nsplit;
    int tiles_x, tiles_y;        // W/16, H/8
    int rev;
    int noremap;                 // experiment (AAU_W3_NOREMAP): the round-1 order, split fastest, no XCD remap
    const float* in_scale;       // BNIN: x = relu(src * in_scale + in_shift) applied on the halo tile in LDS (per Cin channel)
    const float* in_shift;
};

#define AAU_TR16O(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

// KG = 2: the workgroup is TWO such four-wave groups on alternate patches of its K-range, each with its own pair of
// staging buffers, combined through LDS (group 1 -> group 0, a fixed order) before the slab leaves.  The same eight
// waves per CU as two KG = 1 workgroups, but half as many split-K slabs: every launch used to write 56 MB of partial
// sums (512 workgroups x 110 KB, whatever the layer) that wg_reduce_kernel read back.
// BNIN (round 4, aau_conv_wgrad_bnin): the source is the producing layer's RAW conv output and relu(z * scale + shift) is
// applied on the halo tile in LDS: each lane rewrites the 16-byte pieces ITS OWN LDS-DMA fetched, behind the vmcnt wait
// that says they have landed and in front of the barrier that publishes the tile -- in the LOAD half of the ping-pong,
// while the other group multiplies.  Pieces outside the image stay zero (the padding of the ACTIVATION, not of z).
template <int QT, int CJ, int KG, bool BNIN = false>
__global__ __launch_bounds__(256 * KG, 2) void wgrad3x3r_kernel(const W3RArgs a) {
    static_assert(CJ * (QT / 3) == 4 && QT % 3 == 0, "four waves: CJ channel groups x QT/3 q groups");
    constexpr int BQ = QT * 16, BC = CJ * 16;
    constexpr int PR = 8, NPX = PR * 16;                  // pixels per K-step
    constexpr int QS = BQ / 8;                            // 16-B slots per dz row
    constexpr int YPITCH = BQ * 2;                        // 96 or 192 bytes
    constexpr int NLY = (NPX * QS + 255) / 256;           // LDS-DMA instructions per wave per K-step (dz tile): 3 / 6
    constexpr int YB = NLY * 4 * 1024;
    constexpr int XROWS = (PR + 2) * 18;                  // 180 halo pixels
    constexpr int XS = CJ * 2, XPITCH = CJ * 32;          // 16-B slots per halo row, bytes per halo row
    constexpr int NLX = (XROWS * XS + 255) / 256;         // 6 / 3
    constexpr int XB = NLX * 4 * 1024;
    constexpr int STAGE = YB + XB;
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];

    const aau_conv_desc& d = a.d;
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = (KG == 2) ? (wave8 >> 2) : 0;           // K-group
    const int wave = wave8 & 3;                             // wave within the group
    const int tid = (int)threadIdx.x & 255;                 // thread within the group
    const int lane = tid & 63;
    unsigned char* smem = smem_r + grp * 2 * STAGE;
    const int jw = wave % CJ, qg = wave / CJ;

    const int ntc = (d.Cin + BC - 1) / BC;
    // Workgroup order: the K-split (patch range) is the SLOWEST index and the bijective XCD remap (igemm.hip) gives each
    // XCD a contiguous run of logical ids, i.e. the workgroups that share an XCD's L2 are different (q, c) tiles of the
    // SAME patch range: they read the same x halos (once per q tile) and dz tiles (once per c tile).  With the split
    // fastest and no remap, neighbours on an XCD shared nothing and every operand came over the fabric once per tile:
    // 529 MB per launch of L2 misses against 115 MB algorithmic (profiles/r02_pmc_traffic.json).
    int bid = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    if (!a.noremap) {
        const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int ntile = (int)gridDim.x / a.nsplit;
    const int split = a.noremap ? bid % a.nsplit : bid / ntile;
    const int tile = a.noremap ? bid / a.nsplit : bid - split * ntile;
    const int lbid = tile * a.nsplit + split;      // slab index (wg_reduce walks the splits of a tile)
    const int tc = tile % ntc;
    const int tq = tile / ntc;
    const int q0 = tq * BQ, c0 = tc * BC;
    const int p_begin = split * a.patches_per_block;
    const int p_end = min(a.npatch, p_begin + a.patches_per_block);
    if (p_begin >= p_end) {   // never taken with the host's split sizes, but a slab must not stay unwritten
        if (a.ws && grp == 0)
            for (int v = 0; v < 27; ++v)
                *(f32x4*)(a.ws + ((int64_t)lbid * 27 * 256 + v * 256 + tid) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }

    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);

    // ---- fixed staging roles (wave-major: each wave-instruction is 1 KiB linear in LDS) ----
    int yrel[NLY];     // element offset of (patch row, column, channel slot) relative to the patch origin, or -1
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
        const int p = (i * 4 + wave) * 64 + lane;
        const int px = p / QS, s = p - px * QS;
        const int r = px >> 4, cx = px & 15;
        // 192-byte rows: XOR the 32-B granule with bit 2 of the row (source side), see wgrad.hip
        const int sl = (QT == 6) ? ((((s >> 1) ^ ((px >> 2) & 1)) << 1) | (s & 1)) : s;
        yrel[i] = (px < NPX && q0 + sl * 8 < d.Cout) ? (r * d.W + cx) * d.dst_pitch + q0 + sl * 8 : -1;
    }
    int xhy[NLX], xhx[NLX], xch[NLX];
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int p = (i * 4 + wave) * 64 + lane;
        const int px = p / XS, s = p - px * XS;
        const int hy = px / 18, hx = px - hy * 18;
        const int gsw = (CJ == 4) ? ((px >> 1) & 3) : ((px >> 2) & 1);
        const int ch = c0 + ((s >> 1) ^ gsw) * 16 + (s & 1) * 8;
        xhy[i] = (px < XROWS && ch < d.Cin) ? hy : -100000;
        xhx[i] = hx;
        xch[i] = ch;
    }
    // BNIN: all pieces of a lane carry the SAME eight channels (slot = lane % XS and the swizzle bit come from the lane
    // alone: the piece stride 256 is a multiple of 8 pixels x XS slots), so one scale / shift octet per lane serves them
    float bsc[8], bsh[8];
    unsigned xv0 = 0, xv1 = 0;          // valid bits of the pieces last staged into buffer 0 / 1
    if constexpr (BNIN) {
        static_assert((256 / XS) % 8 == 0, "the pieces of a lane share their channel octet");
        const bool okc = xch[0] < d.Cin;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bsc[j] = okc ? a.in_scale[xch[0] + j] : 0.f;
            bsh[j] = okc ? a.in_shift[xch[0] + j] : 0.f;
        }
    }

    auto stage = [&](int buf, int patch) {
        const int pxi = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int pyi = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = pyi * PR, x0 = pxi * 16;
        const int org = ((n * d.H + y0) * d.W + x0);          // pixel index of the patch origin (scalar)
        const bool live = patch < p_end;                      // a group's padding trip (KG = 2, odd patch count): zeros
        unsigned char* sy = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NLY; ++i) {
            const unsigned v = (live && yrel[i] >= 0) ? (unsigned)((org * d.dst_pitch + yrel[i]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sy + (i * 4 + wave) * 1024), 16, (int)v, 0, 0, 0);
        }
        unsigned vb = 0;
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int y = y0 - 1 + xhy[i], x = x0 - 1 + xhx[i];
            const bool ok = live && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            const unsigned v = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + xch[i]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(sy + YB + (i * 4 + wave) * 1024), 16, (int)v, 0, 0, 0);
            if (ok) vb |= 1u << i;
        }
        if constexpr (BNIN) { if (buf == 0) xv0 = vb; else xv1 = vb; }
    };
    // BNIN: relu(z * scale + shift) on this lane's pieces of buffer `buf` (they have landed: the caller's vmcnt wait)
    auto xform = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (BNIN) {
            const unsigned vb = buf == 0 ? xv0 : xv1;
#pragma unroll
            for (int i = 0; i < NLX; ++i) {
                if ((vb >> i) & 1u) {
                    u32x4* q = (u32x4*)(smem + buf * STAGE + YB + (i * 4 + wave) * 1024 + lane * 16);
                    float f[8];
                    unpack8(*q, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * bsc[j] + bsh[j], 0.f);
                    *q = pack8(f);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the rewritten pieces are in LDS before the barrier
        }
    };

    // acc[i][tx][ty]: q tile qg*3 + i, tap (ty, tx), channels c0 + jw*16 ..
    f32x4 acc[3][3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[i][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g16 = lane >> 4, li = lane & 15;
    const int rq = li >> 2, cp = (li & 3) * 4;   // transposed read: lane supplies row rq, columns cp..cp+3
    const int yrow = 4 * g16 + rq;               // this lane's pixel inside a 16-pixel row
    const unsigned lds_base = AAU_LDS_ADDR(smem);
    // dz fragments: byte address of (pixel yrow of patch row 0, q tile qg*3 + i); + patch row * 16 * YPITCH (immediate)
    unsigned abase[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int gi = qg * 3 + i;
        abase[i] = lds_base + yrow * YPITCH + cp * 2 + ((QT == 6) ? ((gi ^ ((yrow >> 2) & 1)) * 32) : gi * 32);
    }
    // x fragments: halo pixel index = yrow + k with k = hy * 18 + tx; the granule swizzle depends on (yrow + k) mod 8
    unsigned xb8[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int pix = yrow + m;
        const int gsw = (CJ == 4) ? ((pix >> 1) & 3) : ((pix >> 2) & 1);
        xb8[m] = lds_base + YB + yrow * XPITCH + cp * 2 + (jw ^ gsw) * 32;
    }

    u32x2 A[4][6];       // [ss][i*2 + h]
    u32x2 B[3][10];      // [tx][halo row]
    auto read_a = [&](const int buf, const int ss) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            AAU_TR16O(A[ss][2 * i], abase[i], buf * STAGE + (2 * ss) * 16 * YPITCH);
            AAU_TR16O(A[ss][2 * i + 1], abase[i], buf * STAGE + (2 * ss + 1) * 16 * YPITCH);
        }
    };
    auto read_b = [&](const int buf, const int hy) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) AAU_TR16O(B[t][hy], xb8[(hy * 18 + t) & 7], buf * STAGE + (hy * 18 + t) * XPITCH);
    };
    auto compute = [&](const int buf) __attribute__((always_inline)) {
        read_a(buf, 0);
        read_b(buf, 0); read_b(buf, 1); read_b(buf, 2); read_b(buf, 3);
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            if (ss < 3) {           // next sub-step's operands: 6 + 6 reads stay in flight behind the MFMAs below
                read_a(buf, ss + 1);
                read_b(buf, 2 * ss + 4); read_b(buf, 2 * ss + 5);
            }
            // LDS returns in order: everything but the 12 reads just issued is back
            if (ss == 0) {
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[0][4]), "+v"(A[0][5]),
                             "+v"(B[0][0]), "+v"(B[1][0]), "+v"(B[2][0]), "+v"(B[0][1]), "+v"(B[1][1]), "+v"(B[2][1]),
                             "+v"(B[0][2]), "+v"(B[1][2]), "+v"(B[2][2]), "+v"(B[0][3]), "+v"(B[1][3]), "+v"(B[2][3]));
            } else if (ss < 3) {
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(A[ss][0]), "+v"(A[ss][1]), "+v"(A[ss][2]), "+v"(A[ss][3]), "+v"(A[ss][4]), "+v"(A[ss][5]),
                             "+v"(B[0][2 * ss + 2]), "+v"(B[1][2 * ss + 2]), "+v"(B[2][2 * ss + 2]),
                             "+v"(B[0][2 * ss + 3]), "+v"(B[1][2 * ss + 3]), "+v"(B[2][2 * ss + 3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[ss][0]), "+v"(A[ss][1]), "+v"(A[ss][2]), "+v"(A[ss][3]), "+v"(A[ss][4]), "+v"(A[ss][5]),
                             "+v"(B[0][2 * ss + 2]), "+v"(B[1][2 * ss + 2]), "+v"(B[2][2 * ss + 2]),
                             "+v"(B[0][2 * ss + 3]), "+v"(B[1][2 * ss + 3]), "+v"(B[2][2 * ss + 3]));
            }
            bf16x8 af[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = AAU_FRAG8(A[ss][2 * i], A[ss][2 * i + 1]);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const bf16x8 bf = AAU_FRAG8(B[t][2 * ss + u], B[t][2 * ss + u + 1]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i][t][u] = AAU_MFMA16(af[i], bf, acc[i][t][u], 0, 0, 0);
                }
        }
    };

    // two patches per loop trip so that the stage index is a compile-time constant in every LDS address
    // group g takes patches p_begin + g, p_begin + g + KG, ...; both groups run the same number of trips (the barriers
    // are workgroup wide), a trip past p_end stages zeros
    const int ntrip = (p_end - p_begin + KG - 1) / KG;
    int patch = p_begin + grp, k = 0;
    if constexpr (KG == 2) {
        // PING-PONG (round 4): the two groups run half a trip apart.  Between the two barriers of a trip group 0 multiplies
        // (transposed reads + 108 MFMAs per wave) while group 1 issues the LDS-DMA of its next patch and waits for the
        // current one; behind the second barrier they swap.  In lockstep (round 3) both groups issued their DMAs, then both
        // multiplied: the matrix pipes idled through every issue phase (the wgradL kernel measured 1.5x between the two
        // forms).  A group's buffer b is restaged only after the barrier that ends its multiply half; waits are counted
        // (all but the youngest NLY + NLX = 9 DMAs of this wave have landed), trips past the range stage zeros.
        static_assert(NLY + NLX == 9, "the counted waits below");
        if (grp == 0) {
            stage(0, patch);
            stage(1, patch + KG);
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            xform(0);
            while (true) {
                __builtin_amdgcn_s_barrier();
                compute(0);
                __builtin_amdgcn_s_barrier();
                stage(0, patch + 2 * KG);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");      // buffer 1 (staged a trip ago) has landed
                xform(1);
                patch += KG;
                if (++k == ntrip) break;
                __builtin_amdgcn_s_barrier();
                compute(1);
                __builtin_amdgcn_s_barrier();
                stage(1, patch + 2 * KG);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                xform(0);
                patch += KG;
                if (++k == ntrip) break;
            }
        } else {
            stage(0, patch);
            while (true) {
                __builtin_amdgcn_s_barrier();
                stage(1, patch + KG);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");      // buffer 0 has landed
                xform(0);
                __builtin_amdgcn_s_barrier();
                compute(0);
                patch += KG;
                if (++k == ntrip) break;
                __builtin_amdgcn_s_barrier();
                stage(0, patch + KG);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                xform(1);
                __builtin_amdgcn_s_barrier();
                compute(1);
                patch += KG;
                if (++k == ntrip) break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero stages past the range: the combine reuses the buffers
    } else {
    stage(0, patch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xform(0);
    __builtin_amdgcn_s_barrier();
    while (true) {
        bool more = k + 1 < ntrip;
        if (more) stage(1, patch + KG);
        compute(0);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform(1);
        __builtin_amdgcn_s_barrier();
        patch += KG; ++k;
        more = k + 1 < ntrip;
        if (more) stage(0, patch + KG);
        compute(1);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform(0);
        __builtin_amdgcn_s_barrier();
        patch += KG; ++k;
    }
    }
    if constexpr (KG == 2) {
        if (a.ws) {     // group 1 hands its 27 accumulator tiles to group 0 through LDS (the staging buffers are dead)
            f32x4* comb = (f32x4*)smem_r;
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < 3; ++u) comb[((i * 3 + t) * 3 + u) * 256 + tid] = acc[i][t][u];
            }
            __syncthreads();
            if (grp == 1) return;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[i][t][u] += comb[((i * 3 + t) * 3 + u) * 256 + tid];
        }
    }

    // acc[i][tx][ty][r] = D[q = q0 + (qg*3 + i)*16 + 4*g16 + r][tap = ty*3 + tx][c = c0 + jw*16 + li]
    if (a.ws) {   // split-K partial in register layout (wg_reduce_kernel<2>)
        float* slab = a.ws + (int64_t)lbid * (27 * 256 * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 3; ++u) *(f32x4*)(slab + (((i * 3 + t) * 3 + u) * 256 + tid) * 4) = acc[i][t][u];
        return;
    }
    const int c = c0 + jw * 16 + li;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + (qg * 3 + i) * 16 + 4 * g16 + r;
                    if (q < d.Cout && c < d.Cin) atomicAdd(a.dw + ((int64_t)q * 9 + (u * 3 + t)) * d.Cin + c, acc[i][t][u][r]);
                }
}

// Which kernel: 0 = wgrad3x3.hip (48 q x 48 c), 1 = <3,4> (48 q x 64 c), 2 = <6,2> (96 q x 32 c).  Never a shape that
// pads more than the 48 x 48 tiling.  Measured per layer (bs 8, base_c 48, A/B on one device, XCD-aware workgroup
// order): <6,2> is the fastest or tied wherever it fits (Cout % 96 == 0, Cin % 32 == 0: -10...20 % against wgrad3x3 on
// d2.1, d3.*, d4.*, u2-u4), <3,4> where only it fits; the 48-channel layers of level 1 keep wgrad3x3.
int wgrad3x3r_variant(const aau_conv_desc* d) {
    if (getenv("AAU_W3_NOR")) return 0;
    if (const char* e = getenv("AAU_W3_R")) return atoi(e);          // experiment: force a variant
    auto eff = [&](int bq, int bc) {
        return (double)d->Cout / ((d->Cout + bq - 1) / bq * bq) * (double)d->Cin / ((d->Cin + bc - 1) / bc * bc);
    };
    const double e0 = eff(48, 48), e1 = eff(48, 64), e2 = eff(96, 32);
    if (e2 >= 0.99 * e0) return 2;
    if (e1 >= 0.99 * e0) return 1;
    return 0;
}

template <int QT, int CJ, int KG, bool BNIN = false>
static int launch_w3r(W3RArgs& a, const aau_conv_desc* d, float* ws, int64_t ws_bytes, int64_t* need, hipStream_t s) {
    constexpr int BQ = QT * 16, BC = CJ * 16;
    constexpr int YB = ((128 * (BQ / 8) + 255) / 256) * 4096, XB = ((180 * CJ * 2 + 255) / 256) * 4096;
    a.tiles_x = d->W / 16;
    a.tiles_y = d->H / 8;
    a.npatch = d->N * a.tiles_x * a.tiles_y;
    const int ntc = (d->Cin + BC - 1) / BC;
    const int64_t tiles = (int64_t)((d->Cout + BQ - 1) / BQ) * ntc;
    int64_t target = 512 / KG;                              // one resident round (8 waves per CU), see wgrad3x3.hip
    if (const char* e = getenv("AAU_W3_TARGET")) target = atoi(e);   // experiment
    int64_t nsplit = target / tiles;                         // never more workgroups than resident slots: a second
                                                             // round of a few workgroups doubles the launch time
    const int64_t maxsplit = (a.npatch + 4 * KG - 1) / (4 * KG);   // at least 4 K-steps per four-wave group
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    a.patches_per_block = (int)((a.npatch + nsplit - 1) / nsplit);
    a.nsplit = (int)((a.npatch + a.patches_per_block - 1) / a.patches_per_block);
    const int64_t grid = tiles * a.nsplit;
    if (grid > 0x7fffffff) { set_error("aau_conv_wgrad: grid too large"); return AAU_E_INVALID; }
    const int64_t bytes = grid * (27 * 256 * 4) * (int64_t)sizeof(float);
    if (need) { *need = bytes; return AAU_OK; }
    if (ws && ws_bytes < bytes) {
        set_error("aau_conv_wgrad: workspace of %lld B, need %lld B (aau_conv_wgrad_ws_bytes)", (long long)ws_bytes, (long long)bytes);
        return AAU_E_INVALID;
    }
    a.ws = ws;
    a.rev = next_traversal();
    a.noremap = getenv("AAU_W3_NOREMAP") ? 1 : 0;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)wgrad3x3r_kernel<QT, CJ, KG, BNIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    static_assert(KG == 1 || KG * 2 * (YB + XB) >= 27 * 256 * 16, "the combine needs 27 x 256 float4 of LDS");
    hipLaunchKernelGGL((wgrad3x3r_kernel<QT, CJ, KG, BNIN>), dim3((unsigned)grid), dim3(256 * KG), KG * 2 * (YB + XB), s, a);
    if (!ws) return check_launch("aau_conv_wgrad(3x3 row reuse)");
    WRedArgs r;
    r.ws = ws; r.dw = a.dw;
    r.nsplit = a.nsplit; r.NV = 27; r.sub = 1; r.kwaves = 1;
    r.TQ = QT; r.TC = CJ; r.ntc = ntc; r.T = 9; r.Cout = d->Cout; r.Cin = d->Cin;
    r.nslots = tiles * 27 * 256;
    return wg_reduce_launch(2, r, s);
}

int wgrad3x3r_launch(int variant, const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                     int64_t ws_bytes, int64_t* need, hipStream_t s, const float* in_scale, const float* in_shift) {
    W3RArgs a;
    a.d = *d;
    a.in_scale = in_scale; a.in_shift = in_shift;
    if (in_scale && variant != 2) { set_error("aau_conv_wgrad_bnin: only the 96 x 32 row-reuse tiling applies BatchNorm on load"); return AAU_E_INVALID; }
    a.src = src; a.dz = dz; a.dw = dw; a.ws = nullptr;
    const int64_t npix = (int64_t)d->N * d->H * d->W;
    const int64_t sb = ((npix - 1) * d->src_pitch + d->Cin) * 2, zb = ((npix - 1) * d->dst_pitch + d->Cout) * 2;
    if (sb >= 0x7fffffff || zb >= 0x7fffffff) { set_error("aau_conv_wgrad: tensors must stay below 2 GiB"); return AAU_E_INVALID; }
    a.src_bytes = (unsigned)sb;
    a.dz_bytes = (unsigned)zb;
    static const bool kg1env = getenv("AAU_W3_KG1") != nullptr;    // experiment: one four-wave group per workgroup
    // two K-groups only where whole K-ranges still fill the chip: with 96 tiles (768 -> 384 channels) 256 / 96 = 2 ranges
    // leave a quarter of the CUs idle, the five ranges of the one-group form 6 %
    const int BQv = variant == 1 ? 48 : 96, BCv = variant == 1 ? 64 : 32;
    const int64_t tiles_v = (int64_t)((d->Cout + BQv - 1) / BQv) * ((d->Cin + BCv - 1) / BCv);
    const bool kg1 = kg1env || (256 / tiles_v) * tiles_v < 230;
    if (in_scale) return kg1 ? launch_w3r<6, 2, 1, true>(a, d, ws, ws_bytes, need, s) : launch_w3r<6, 2, 2, true>(a, d, ws, ws_bytes, need, s);
    if (kg1) {
        if (variant == 1) return launch_w3r<3, 4, 1>(a, d, ws, ws_bytes, need, s);
        return launch_w3r<6, 2, 1>(a, d, ws, ws_bytes, need, s);
    }
    if (variant == 1) return launch_w3r<3, 4, 2>(a, d, ws, ws_bytes, need, s);
    return launch_w3r<6, 2, 2>(a, d, ws, ws_bytes, need, s);
}

}  // namespace aau

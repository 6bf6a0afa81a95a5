// Weight gradient of the 3x3 (stride 1, pad 1, dilation 1) convolutions on MFMA, all 9 taps
// per workgroup (gfx950).
//
//   dw[q][tap][c] += sum_m dz[m][q] * x[m + off(tap)][c]
//
// The generic kernel (wgrad.hip) gives each workgroup ONE tap, so per 128-pixel K-step it
// stages 24 KB for 36 MFMAs (24 FLOP per staged byte: L2->LDS fill bound).  Here a workgroup
// owns a 48(q) x 48(c) slice of the gradient for ALL taps: the GEMM is
// [48 q] x [9 taps * 48 c = 432 columns] with K = pixels.  Per K-step (an 8 x 16 pixel patch)
// it stages the dz tile (12 KB) and the patch's 10 x 18 halo of x (17 KB, once for all 9 taps)
// for 324 MFMAs: 183 FLOP per staged byte.
//
// Both tiles stay [pixel][channel] in LDS as the LDS-DMA delivers them; the MFMA operands
// (8 consecutive pixels of one channel per lane) come out of ds_read_b64_tr_b16.  The
// k <-> pixel map of a 32-pixel sub-step is k = 8g + 4h + e <-> (patch row h, column 4g + e):
// each transposed read covers 4 consecutive LDS rows at any tap shift and a half-wave covers
// 8 consecutive 96-byte rows -- conflict free.
// 4 waves split the 27 column tiles (tap, 16-channel group) 7/7/7/6; each keeps 3 x 7
// accumulator tiles.  Two LDS stages (58 KB) -> 2 workgroups per CU.  Split-K over patch
// ranges, fp32 atomics into the channels_last gradient [Cout][9][Cin].
#include <stdlib.h>
#include "common.h"

namespace aau {

// timing-only ablation: -DABL_NOATOMIC turns the split-K adds into plain stores (wrong sums)
#ifdef ABL_NOATOMIC
#define WG_ADD(p, v) (*(p) = (v))
#else
#define WG_ADD(p, v) atomicAdd((p), (v))
#endif

struct W3Args {
    aau_conv_desc d;
    const unsigned short* src;   // x   [N][H][W] pitch src_pitch, Cin channels
    const unsigned short* dz;    // dz  [N][H][W] pitch dst_pitch, Cout channels
    float* dw;
    float* ws;                   // split-K slabs [workgroup][QT*7 acc tiles][256 threads][4] (null: fp32 atomics)
    unsigned src_bytes, dz_bytes;
    int npatch;                  // N * (H/8) * (W/16)
    int patches_per_block, nsplit;
    int tiles_x, tiles_y;        // W/16, H/8
    int rev;                     // 1: workgroups take the split ranges from the end (aau_traverse)
    int noremap;                 // experiment (AAU_W3_NOREMAP): the round-1 order, split fastest, no XCD remap
    // aau_conv_wgrad_bnin: src is the RAW conv output z of the producing BatchNorm -> ReLU layer; the kernel applies
    // x = relu(z * in_scale + in_shift) in LDS on the pieces each lane fetched itself (see conv3x3s.hip, BNIN)
    const float* in_scale = nullptr;
    const float* in_shift = nullptr;
};

// QT = 16-channel q tiles per workgroup (3: 48 channels, 6: 96 channels); PR = patch rows per K-step.
// <3, 8> (default): 21 accumulator tiles per wave, 20 transposed reads per 21 MFMAs, 2 waves per SIMD;
// <6, 4> (opt-in): 42 tiles per wave, half the LDS reads per FLOP, but 200 VGPRs -> 1 wave per SIMD.
// NG = 2 (with QT = 6): 8 waves, two groups of four that share the staged x halo; group g owns q tiles 3g..3g+2 of the
// 96-channel dz tile.  Same accumulators per wave and waves per SIMD as <3, 8, 1> at two workgroups per CU, but the
// halo is staged once per 96 output channels instead of once per 48: -29 % L2 -> LDS fill bytes per FLOP.
template <int QT, int PR, int NG = 1, bool BNIN = false>
__global__ __launch_bounds__(256 * NG) void wgrad3x3_kernel(const W3Args a) {
    constexpr int BQ = QT * 16;
    constexpr int NW = 4 * NG;                            // waves
    constexpr int QW = QT / NG;                           // q tiles per wave
    constexpr int NPX = PR * 16;                          // pixels per K-step
    constexpr int QS = BQ / 8;                            // 16-B slots per dz row
    constexpr int YPITCH = BQ * 2;                        // 96 or 192 bytes
    constexpr int NLY = (NPX * QS + 64 * NW - 1) / (64 * NW);   // LDS-DMA instructions per wave per K-step (dz tile)
    constexpr int YB = NLY * NW * 1024;                    // staged bytes (pieces past the tile are zero fill)
    constexpr int XROWS = (PR + 2) * 18;                  // halo pixels
    constexpr int NLX = (XROWS * 6 + 64 * NW - 1) / (64 * NW);  // LDS-DMA instructions per wave per K-step (x halo)
    constexpr int XB = NLX * NW * 1024;                    // staged bytes (rows >= XROWS are zero fill)
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (YB + XB) + (BNIN ? 2 * 48 * 4 : 0)];
    float* const s_in = (float*)(smem + 2 * (YB + XB));        // BNIN: [2][48] scale | shift of this workgroup's channel tile
    auto sY = [&](int b) -> unsigned char* { return smem + b * (YB + XB); };
    auto sX = [&](int b) -> unsigned char* { return smem + b * (YB + XB) + YB; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wc = wave & 3, grp = wave >> 2;   // column-tile wave, q group

    const int ntc = (d.Cin + 47) / 48;
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
    const int q0 = tq * BQ, c0 = tc * 48;
    const int p_begin = split * a.patches_per_block;
    const int p_end = min(a.npatch, p_begin + a.patches_per_block);
    if (p_begin >= p_end) {   // never taken with the host's split sizes, but a slab must not stay unwritten
        if (a.ws)
            for (int v = grp * QW * 7; v < (grp + 1) * QW * 7; ++v)
                *(f32x4*)(a.ws + ((int64_t)lbid * QT * 7 * 256 + v * 256 + (threadIdx.x & 255)) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }

    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);

    // ---- fixed staging roles: 16-B piece p = instr*256... (wave-major so that each wave-instruction is 1 KiB linear) ----
    int yrel[NLY];     // element offset of (patch row, column, channel slot) relative to the patch origin, or -1
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
        const int p = (i * NW + wave) * 64 + lane;
        const int px = p / QS, s = p - px * QS;
        const int r = px >> 4, cx = px & 15;
        // 192-byte rows: XOR the 32-B granule with bit 2 of the row (source side), see wgrad.hip
        const int sl = (QT == 6) ? ((((s >> 1) ^ ((px >> 2) & 1)) << 1) | (s & 1)) : s;
        yrel[i] = (px < NPX && q0 + sl * 8 < d.Cout) ? (r * d.W + cx) * d.dst_pitch + q0 + sl * 8 : -1;
    }
    int xhy[NLX], xhx[NLX], xch[NLX];
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int p = (i * NW + wave) * 64 + lane;
        const int px = p / 6, s = p - px * 6;
        const int hy = px / 18, hx = px - hy * 18;
        xhy[i] = (px < XROWS && c0 + s * 8 < d.Cin) ? hy : -100000;
        xhx[i] = hx;
        xch[i] = c0 + s * 8;
        if (d.src_split_c > 0 && xch[i] >= d.src_split_c) xch[i] += d.src_split_off - d.src_split_c;   // second plane (aau.h)
    }

    unsigned xv = 0;        // BNIN: bit i = x piece i of the buffer staged last lies inside the image (this lane)
    auto stage = [&](int buf, int patch) {
        const int pxi = patch % a.tiles_x;
        const int t2 = patch / a.tiles_x;
        const int pyi = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = pyi * PR, x0 = pxi * 16;
        const int org = ((n * d.H + y0) * d.W + x0);          // pixel index of the patch origin (scalar)
#pragma unroll
        for (int i = 0; i < NLY; ++i) {
            const unsigned v = yrel[i] >= 0 ? (unsigned)((org * d.dst_pitch + yrel[i]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sY(buf) + (i * NW + wave) * 1024), 16, (int)v, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int y = y0 - 1 + xhy[i], x = x0 - 1 + xhx[i];
            const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            const unsigned v = ok ? (unsigned)((((n * d.H + y) * d.W + x) * d.src_pitch + xch[i]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(sX(buf) + (i * NW + wave) * 1024), 16, (int)v, 0, 0, 0);
            if constexpr (BNIN) xv = i == 0 ? (ok ? 1u : 0u) : (xv | ((ok ? 1u : 0u) << i));
        }
    };
    // BNIN: x = relu(z * scale + shift) on the pieces of buffer `buf` this lane fetched itself (its own vmcnt(0) says they
    // have landed; the barrier behind it publishes them).  Pieces outside the image stay the zeros the range check
    // delivered: the padding of the activation is zero, not relu(shift).  aau_bn_act's arithmetic, bit for bit.
    auto xform = [&](int buf) {
        if constexpr (BNIN) {
#pragma unroll
            for (int i = 0; i < NLX; ++i) {
                if ((xv >> i) & 1u) {
                    const int p = (i * NW + wave) * 64 + lane;
                    const int s8 = (p % 6) * 8;
                    u32x4* pz = (u32x4*)(sX(buf) + (i * NW + wave) * 1024 + lane * 16);
                    float f[8], sc[8], sh[8];
                    *(f32x4*)(sc) = *(const f32x4*)(s_in + s8); *(f32x4*)(sc + 4) = *(const f32x4*)(s_in + s8 + 4);
                    *(f32x4*)(sh) = *(const f32x4*)(s_in + 48 + s8); *(f32x4*)(sh + 4) = *(const f32x4*)(s_in + 48 + s8 + 4);
                    unpack8(*pz, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
                    *pz = pack8(f);
                }
            }
        }
    };
    if constexpr (BNIN) {
        for (int i = tid; i < 48; i += 256 * NG) {
            const bool in = c0 + i < d.Cin;
            s_in[i] = in ? a.in_scale[c0 + i] : 0.f;
            s_in[48 + i] = in ? a.in_shift[c0 + i] : 0.f;
        }
        // (published by the barrier behind the first stage below)
    }

    // ---- this wave's column tiles: ct = 7*wave + n, n = 0..6 (tap = ct/3, channel group j = ct%3) ----
    f32x4 acc[QW][7];
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int n = 0; n < 7; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int coff[7];    // byte offset inside the halo image of (tap shift, channel group) for column tile n
    const int nct = (wc == 3) ? 6 : 7;
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        int ct = 7 * wc + n;
        if (ct > 26) ct = 26;
        const int tap = ct / 3, j = ct - tap * 3;
        const int ty = tap / 3, tx = tap - ty * 3;
        coff[n] = (ty * 18 + tx) * 96 + j * 32;
    }

    const int g16 = lane >> 4, li = lane & 15;
    const int rq = li >> 2, cp = (li & 3) * 4;   // transposed read: lane supplies row rq, columns cp..cp+3
    const int yrow = 4 * g16 + rq;                               // dz row inside a 16-pixel patch row
    const int xbase = (4 * g16 + rq) * 96 + cp * 2;              // + (2*ss + h)*18*96 + coff[n]
    // Transposed reads in asm form (common.h, AAU_TR16): with the builtin hipcc drained the LDS-DMA of the NEXT patch
    // (issued just above) with a vmcnt(0) before the first read, so staging and MFMAs never overlapped.
    const unsigned lds_base = AAU_LDS_ADDR(smem);
    auto compute = [&](int buf) {
        const unsigned by = lds_base + buf * (YB + XB);
        const unsigned bx = by + YB;
#pragma unroll
        for (int ss = 0; ss < PR / 2; ++ss) {
            u32x2 alo[QW], ahi[QW], blo[7], bhi[7];
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const int r0 = (2 * ss) * 16 + yrow, r1 = r0 + 16;
                const int gi = grp * QW + i;          // q tile inside the staged dz tile
                int o0, o1;
                if constexpr (QT == 3) {
                    o0 = r0 * 96 + gi * 32 + cp * 2;
                    o1 = r1 * 96 + gi * 32 + cp * 2;
                } else {   // granule swizzle of the 192-byte rows
                    o0 = r0 * 192 + ((gi ^ ((r0 >> 2) & 1)) * 32) + cp * 2;
                    o1 = r1 * 192 + ((gi ^ ((r1 >> 2) & 1)) * 32) + cp * 2;
                }
                AAU_TR16(alo[i], by + o0);
                AAU_TR16(ahi[i], by + o1);
            }
#pragma unroll
            for (int n = 0; n < 7; ++n) {      // the 4th wave's 7th tile is a duplicate (ct clamped): read, never used
                const unsigned o = bx + xbase + (2 * ss) * 18 * 96 + coff[n];
                AAU_TR16(blo[n], o);
                AAU_TR16(bhi[n], o + 18 * 96);
            }
            // LDS returns in order: all but the youngest 8 reads (column tiles 3..6) are back
            if constexpr (QW == 3) {
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(ahi[0]), "+v"(ahi[1]),
                             "+v"(ahi[2]), "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]));
            } else {
                static_assert(QW == 3 || QW == 6, "operand lists of the waits");
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3 % QW]), "+v"(alo[4 % QW]),
                             "+v"(alo[5 % QW]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]), "+v"(ahi[3 % QW]), "+v"(ahi[4 % QW]),
                             "+v"(ahi[5 % QW]), "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]));
            }
            bf16x8 af[QW];
#pragma unroll
            for (int i = 0; i < QW; ++i) af[i] = AAU_FRAG8(alo[i], ahi[i]);
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const bf16x8 bf = AAU_FRAG8(blo[n], bhi[n]);
#pragma unroll
                for (int i = 0; i < QW; ++i) acc[i][n] = AAU_MFMA16(af[i], bf, acc[i][n], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[3]), "+v"(blo[4]), "+v"(blo[5]), "+v"(blo[6]), "+v"(bhi[3]),
                         "+v"(bhi[4]), "+v"(bhi[5]), "+v"(bhi[6]));
#pragma unroll
            for (int n = 3; n < 7; ++n) {
                if (n < nct) {   // wave-uniform
                    const bf16x8 bf = AAU_FRAG8(blo[n], bhi[n]);
#pragma unroll
                    for (int i = 0; i < QW; ++i) acc[i][n] = AAU_MFMA16(af[i], bf, acc[i][n], 0, 0, 0);
                }
            }
        }
    };

    int patch = p_begin;
    stage(0, patch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BNIN) { __syncthreads(); xform(0); }      // (s_in is visible behind the first barrier)
    __syncthreads();
    int buf = 0;
    while (true) {
        const bool more = patch + 1 < p_end;
        if (more) stage(buf ^ 1, patch + 1);
        compute(buf);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        ++patch;
    }

    // acc[i][n][r] = D[q = q0 + i*16 + 4*g16 + r][tap, c = c0 + j*16 + li]
    if (a.ws) {   // split-K partial in register layout (see wgrad.hip, wg_reduce_kernel<1>)
        float* slab = a.ws + (int64_t)lbid * (QT * 7 * 256 * 4);
#pragma unroll
        for (int n = 0; n < 7; ++n) {
            if (n >= nct) continue;
#pragma unroll
            for (int i = 0; i < QW; ++i) *(f32x4*)(slab + (((grp * QW + i) * 7 + n) * 256 + (tid & 255)) * 4) = acc[i][n];
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        if (n >= nct) continue;
        const int ct = 7 * wc + n;
        const int tap = ct / 3, j = ct - tap * 3;
        const int c = c0 + j * 16 + li;
#pragma unroll
        for (int i = 0; i < QW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + (grp * QW + i) * 16 + 4 * g16 + r;
                if (q < d.Cout && c < d.Cin) WG_ADD(a.dw + ((int64_t)q * 9 + tap) * d.Cin + c, acc[i][n][r]);
            }
    }
}

bool wgrad3x3_applicable(const aau_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && d->H == d->Ho &&
           d->W == d->Wo && d->H % 8 == 0 && d->W % 16 == 0;
}

template <int QT, int PR, int NG = 1, bool BNIN = false>
static int launch_w3(W3Args& a, const aau_conv_desc* d, float* ws, int64_t ws_bytes, int64_t* need, hipStream_t s) {
    constexpr int BQ = QT * 16;
    a.tiles_x = d->W / 16;
    a.tiles_y = d->H / PR;
    a.npatch = d->N * a.tiles_x * a.tiles_y;
    const int64_t tiles = (int64_t)((d->Cout + BQ - 1) / BQ) * ((d->Cin + 47) / 48);
    // Every workgroup ends with BQ x 432 fp32 atomics, which execute at the memory side at ~1.3 TB/s chip-wide and
    // are NOT hidden behind other workgroups once the whole grid is resident: a plain-store ablation of this
    // epilogue (-DABL_NOATOMIC) is 20-35 us faster per launch.  One resident round (2 workgroups per CU) is the
    // measured optimum for every layer; 1024 cost +15-30 % on the 174-GFLOP layers, 256 +30 % on the others.
    int64_t target = 512 / NG;                                       // NG = 2: one 8-wave workgroup per CU
    if (const char* e = getenv("AAU_W3_TARGET")) target = atoi(e) / NG;   // experiment
    int64_t nsplit = (target + tiles / 2) / tiles;
    const int64_t maxsplit = (a.npatch + 3) / 4;             // at least 4 K-steps per workgroup
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    a.patches_per_block = (int)((a.npatch + nsplit - 1) / nsplit);
    a.nsplit = (int)((a.npatch + a.patches_per_block - 1) / a.patches_per_block);
    const int64_t grid = tiles * a.nsplit;
    if (grid > 0x7fffffff) { set_error("aau_conv_wgrad: grid too large"); return AAU_E_INVALID; }
    const int64_t bytes = grid * (QT * 7 * 256 * 4) * (int64_t)sizeof(float);
    if (need) { *need = bytes; return AAU_OK; }
    if (ws && ws_bytes < bytes) {
        set_error("aau_conv_wgrad: workspace of %lld B, need %lld B (aau_conv_wgrad_ws_bytes)", (long long)ws_bytes, (long long)bytes);
        return AAU_E_INVALID;
    }
    a.ws = ws;
    a.rev = next_traversal();
    a.noremap = getenv("AAU_W3_NOREMAP") ? 1 : 0;
    hipLaunchKernelGGL((wgrad3x3_kernel<QT, PR, NG, BNIN>), dim3((unsigned)grid), dim3(256 * NG), 0, s, a);
    if (!ws) return check_launch("aau_conv_wgrad(3x3)");
    WRedArgs r;
    r.ws = ws; r.dw = a.dw;
    r.nsplit = a.nsplit; r.NV = QT * 7; r.sub = 1; r.kwaves = 1;
    r.TQ = 1; r.TC = 1; r.ntc = (d->Cin + 47) / 48; r.T = 9; r.Cout = d->Cout; r.Cin = d->Cin;
    r.nslots = tiles * (QT * 7) * 256;
    return wg_reduce_launch(1, r, s);
}

int wgrad3x3_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                    int64_t ws_bytes, int64_t* need, hipStream_t s, const float* in_scale, const float* in_shift) {
    W3Args a;
    a.d = *d;
    a.src = src; a.dz = dz; a.dw = dw; a.ws = nullptr;
    a.in_scale = in_scale; a.in_shift = in_shift;
    const int64_t npix = (int64_t)d->N * d->H * d->W;
    const int64_t sb = ((npix - 1) * d->src_pitch + d->Cin + (d->src_split_c > 0 ? d->src_split_off - d->src_split_c : 0)) * 2;
    const int64_t zb = ((npix - 1) * d->dst_pitch + d->Cout) * 2;
    if (sb >= 0x7fffffff || zb >= 0x7fffffff) { set_error("aau_conv_wgrad: tensors must stay below 2 GiB"); return AAU_E_INVALID; }
    a.src_bytes = (unsigned)sb;
    a.dz_bytes = (unsigned)zb;
    // The 96-channel variant halves the LDS reads per FLOP but needs 200 VGPRs (1 wave per SIMD): measured
    // 1.7-1.9x SLOWER than <3, 8> at 2 waves per SIMD on the same device, so it stays opt-in (experiments).
    if (d->Cout > 48 && getenv("AAU_W3_WIDE")) return launch_w3<6, 4>(a, d, ws, ws_bytes, need, s);
    if (getenv("AAU_W3_PR4")) return launch_w3<3, 4>(a, d, ws, ws_bytes, need, s);   // experiment: 4 workgroups per CU
    if (d->Cout > 48 && getenv("AAU_W3_NG2")) return launch_w3<6, 8, 2>(a, d, ws, ws_bytes, need, s);   // experiment
    if (in_scale) return launch_w3<3, 8, 1, true>(a, d, ws, ws_bytes, need, s);
    return launch_w3<3, 8>(a, d, ws, ws_bytes, need, s);
}

}  // namespace aau

// Weight-gradient implicit GEMM on MFMA for gfx950.
//
//   dw[q][tap][c] += sum_m dz[m][q] * src[gather(m,tap)][c]
//
// GEMM view: rows = output channels q, columns = input channels c (one tap per
// workgroup), reduction K = pixels m.  Both operands are pixel-major in HBM (NHWC), but
// the MFMA wants 8 consecutive K per lane, i.e. 8 consecutive PIXELS of one channel:
// both LDS tiles are kept [pixel][channel] exactly as they arrive (global_load_lds,
// 16 B per lane) and read back transposed with ds_read_b64_tr_b16 -- no transposition
// pass, no scalar LDS reads.  Since the two operands use the same k <-> pixel map, the
// map is free: k = 8g + 4h + e  <->  pixel 16h + 4g + e of a 32-pixel sub-step, which
// makes every half-wave of a transposed read touch 8 consecutive LDS rows.
// Rows are 96 B (48-channel tile: conflict free as is) or 192 B (96-channel tile: the
// 32-B granule is XOR-ed with bit 2 of the row on the SOURCE side of the LDS-DMA and
// on the read side).
//
// Split-K over pixel ranges; partial tiles are added with fp32 atomics (contiguous 64-B
// runs) into the channels_last gradient [Cout][taps][Cin], which the caller zeroes.
//
// Replaces the weight part of ATen convolution_backward for pipeline:63,71-78,88-89,101.
#include <stdlib.h>
#include "common.h"

namespace aau {

// timing-only ablation: -DABL_NOATOMIC turns the split-K adds into plain stores (wrong sums)
#ifdef ABL_NOATOMIC
#define WG_ADD(p, v) (*(p) = (v))
#else
#define WG_ADD(p, v) atomicAdd((p), (v))
#endif


struct WgradArgs {
    aau_conv_desc d;
    const unsigned short* src;
    const unsigned short* dz;
    float* dw;
    int M;
    int pix_per_split;  // multiple of the K-step
    int nsplit;
    int linear;         // 1: gather(m) == m (1x1, stride 1, pad 0)
    int rev;            // 1: workgroups take the split ranges from the end (aau_traverse)
    float* ws;          // split-K slabs [workgroup][9 acc tiles][256 threads][4] (null: fp32 atomics into dw)
    unsigned src_bytes, dz_bytes;   // extents for the buffer descriptors of the fast issue path
    int fast_ok;        // both tensors below 2 GiB (32-bit buffer offsets); AAU_WG_NOFAST=1 switches the path off (A/B)
    // aau_conv_wgrad_bnin_dz: the `dz` operand is a raw conv output; relu(dz * dz_scale + dz_shift) is applied on its tile in
    // LDS by the lanes that fetched the pieces (fast issue path only: every row of every step is a real pixel)
    const float* dz_scale = nullptr;
    const float* dz_shift = nullptr;
};

// byte offset of (row, channel ch [multiple of 4]) in a [rows][48*TT] bf16 tile
template <int TT>
__device__ __forceinline__ int tile_off(int row, int ch) {
    if constexpr (TT == 1) {
        return row * 96 + ch * 2;
    } else {
        const int g = (ch >> 4) ^ ((row >> 2) & 1);
        return row * 192 + g * 32 + (ch & 15) * 2;
    }
}

template <int TQ, int TC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int KWAVES = 4 / (TQ * TC);
    constexpr int KSUBW = (TQ * TC == 1) ? 1 : 2;
    constexpr int BKP = 32 * KWAVES * KSUBW;          // pixels per K-step
    constexpr int NLY = BKP * 6 * TQ / 256;           // 16-B loads per thread, dz tile
    constexpr int NLX = BKP * 6 * TC / 256;           // 16-B loads per thread, src tile
    constexpr int YB = BKP * 96 * TQ, XB = BKP * 96 * TC;  // tile bytes
    static_assert(BKP * 6 * TQ % 256 == 0 && BKP * 6 * TC % 256 == 0, "tile/threads");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (YB + XB)];
    __shared__ __attribute__((aligned(16))) float s_dz[2 * 96];         // BNIN-dz: scale | shift of this workgroup's q tile
    auto sY = [&](int buf) -> unsigned char* { return smem + buf * (YB + XB); };
    auto sX = [&](int buf) -> unsigned char* { return smem + buf * (YB + XB) + YB; };

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wq = (TQ == 2) ? ((TC == 2) ? (wave >> 1) : (wave & 1)) : 0;
    const int wc = (TC == 2) ? ((TQ == 2) ? (wave & 1) : (wave & 1)) : 0;
    const int wk = (TQ * TC == 4) ? 0 : ((TQ * TC == 2) ? (wave >> 1) : wave);

    const int T = d.KH * d.KW;
    const int ntq = (d.Cout + 48 * TQ - 1) / (48 * TQ);
    const int ntc = (d.Cin + 48 * TC - 1) / (48 * TC);
    // K-split slowest + bijective XCD remap: the workgroups of one XCD are different (q, c, tap) tiles of the SAME pixel
    // range, so they share their dz / x rows in that XCD's L2 (see wgrad3x3r.hip)
    int bid = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    {
        const int nwg = (int)gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int ntile = (int)gridDim.x / a.nsplit;
    const int split = bid / ntile;
    bid -= split * ntile;
    const int lbid = bid * a.nsplit + split;      // slab index (wg_reduce walks the splits of a tile)
    const int tap = bid % T;
    bid /= T;
    const int tc = bid % ntc;
    const int tq = bid / ntc;
    const int q0 = tq * 48 * TQ, c0 = tc * 48 * TC;

    const int mb = split * a.pix_per_split;
    const int me = min(a.M, mb + a.pix_per_split);
    if (mb >= me) {  // uniform; never taken with the host's split sizes, but a slab must not stay unwritten
        if (a.ws)
            for (int v = 0; v < 9; ++v) *(f32x4*)(a.ws + ((int64_t)lbid * 9 * 256 + v * 256 + threadIdx.x) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }

    const unsigned short* zero = (const unsigned short*)g_zero_page;
    const int dy = (tap / d.KW) * d.dil - d.pad, dx = (tap % d.KW) * d.dil - d.pad;
    const int HoWo = d.Ho * d.Wo;

    // ---- fixed (row, channel) of each 16-B piece this thread stages ----
    int yrow[NLY], ych[NLY];
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
        const int p = tid + 256 * i;
        const int row = p / (6 * TQ), s = p % (6 * TQ);
        yrow[i] = row;
        if constexpr (TQ == 1) ych[i] = s * 8;
        else ych[i] = ((((s >> 1) ^ ((row >> 2) & 1)) << 1) | (s & 1)) * 8;
    }
    int xrow[NLX], xch[NLX];
    int xn[NLX], xy[NLX], xx[NLX];  // decoded output-grid position of pixel mb + row
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int p = tid + 256 * i;
        const int row = p / (6 * TC), s = p % (6 * TC);
        xrow[i] = row;
        if constexpr (TC == 1) xch[i] = s * 8;
        else xch[i] = ((((s >> 1) ^ ((row >> 2) & 1)) << 1) | (s & 1)) * 8;
        const int m = mb + row;
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        xn[i] = n;
        xy[i] = rem / d.Wo;
        xx[i] = rem - xy[i] * d.Wo;
    }

    // Fast issue path (round 4; the form of conv1x1_rs / wgradL): when every K-step of this workgroup is a whole run of BKP
    // consecutive output pixels whose source pixels are equally spaced (1x1, or the 2x2 / stride-2 gather of a ConvTranspose
    // gradient with BKP dividing the output row), the per-lane part of a piece's address is a constant of the workgroup --
    // channel masks folded in as an out-of-range offset -- and the step travels in the SCALAR offset of the LDS-DMA: no
    // 64-bit address arithmetic, predicates or row / column decode per piece and step (the general path below spends
    // more issue cycles on those than the step's MFMAs take: these launches ran at 3-4 TB/s).
    constexpr unsigned OOB = 0x80000000u;
    const bool gather2 = !a.linear && d.KH == 2 && d.KW == 2 && d.stride == 2 && d.pad == 0 && d.dil == 1 && d.H == 2 * d.Ho &&
                         d.W == 2 * d.Wo && d.Wo % BKP == 0;
    const bool fast = a.fast_ok && (a.linear || gather2) && (me - mb) % BKP == 0 && mb % BKP == 0;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    unsigned vy[NLY], vx[NLX];
#pragma unroll
    for (int i = 0; i < NLY; ++i) vy[i] = q0 + ych[i] < d.Cout ? (unsigned)((yrow[i] * d.dst_pitch + q0 + ych[i]) * 2) : OOB;
#pragma unroll
    for (int i = 0; i < NLX; ++i)
        vx[i] = c0 + xch[i] < d.Cin ? (unsigned)(((a.linear ? xrow[i] : 2 * xrow[i]) * d.src_pitch + c0 + xch[i]) * 2) : OOB;
    auto stage_fast = [&](int buf, int mbase) {
        const unsigned sy = (unsigned)mbase * (unsigned)(d.dst_pitch * 2);
        unsigned sx;
        if (a.linear) {
            sx = (unsigned)mbase * (unsigned)(d.src_pitch * 2);
        } else {
            const unsigned xo = (unsigned)mbase % (unsigned)d.Wo, t = (unsigned)mbase / (unsigned)d.Wo;
            const unsigned yo = t % (unsigned)d.Ho, n = t / (unsigned)d.Ho;
            sx = (unsigned)(((n * d.H + 2 * yo + dy) * d.W + 2 * xo + dx) * d.src_pitch * 2);
        }
#pragma unroll
        for (int i = 0; i < NLY; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(sY(buf) + (256 * i + wave * 64) * 16), 16, (int)vy[i], (int)sy, 0, 0);
#pragma unroll
        for (int i = 0; i < NLX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(sX(buf) + (256 * i + wave * 64) * 16), 16, (int)vx[i], (int)sx, 0, 0);
    };
    auto stage = [&](int buf, int mbase) {
        if (fast) { stage_fast(buf, mbase); return; }
#pragma unroll
        for (int i = 0; i < NLY; ++i) {
            const int m = mbase + yrow[i];
            const int q = q0 + ych[i];
            const unsigned short* g = (m < me && q < d.Cout) ? a.dz + (int64_t)m * d.dst_pitch + q : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sY(buf) + (256 * i + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int m = mbase + xrow[i];
            const int c = c0 + xch[i];
            const unsigned short* g = zero;
            if (m < me && c < d.Cin) {
                if (a.linear) {
                    g = a.src + (int64_t)m * d.src_pitch + c;
                } else {
                    const int y = xy[i] * d.stride + dy, x = xx[i] * d.stride + dx;
                    if ((unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W)
                        g = a.src + ((int64_t)(xn[i] * d.H + y) * d.W + x) * d.src_pitch + c;
                }
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sX(buf) + (256 * i + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto advance = [&]() {
        if (a.linear || fast) return;
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            xx[i] += BKP;
            while (xx[i] >= d.Wo) { xx[i] -= d.Wo; ++xy[i]; }
            while (xy[i] >= d.Ho) { xy[i] -= d.Ho; ++xn[i]; }
        }
    };

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: lane 4*rq+p of a 16-lane group supplies row rq, columns 4p..4p+3
    const int g16 = lane >> 4, li = lane & 15;
    const int rq = li >> 2, cp = (li & 3) * 4;
    // transposed reads in asm form (common.h, AAU_TR16): the builtin made hipcc drain the next K-step's LDS-DMA first
    const unsigned lds_base = AAU_LDS_ADDR(smem);
    auto compute = [&](int buf) {
        const unsigned by = lds_base + buf * (YB + XB);
        const unsigned bx = by + YB;
#pragma unroll
        for (int ks = 0; ks < KSUBW; ++ks) {
            const int rbase = (wk * KSUBW + ks) * 32 + 4 * g16 + rq;
            u32x2 alo[3], ahi[3], blo[3], bhi[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ch = wq * 48 + i * 16 + cp;
                AAU_TR16(alo[i], by + tile_off<TQ>(rbase, ch));
                AAU_TR16(ahi[i], by + tile_off<TQ>(rbase + 16, ch));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ch = wc * 48 + j * 16 + cp;
                AAU_TR16(blo[j], bx + tile_off<TC>(rbase, ch));
                AAU_TR16(bhi[j], bx + tile_off<TC>(rbase + 16, ch));
            }
            // in-order LDS returns: everything but the last two column fragments first, then those
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(ahi[0]), "+v"(ahi[1]),
                         "+v"(ahi[2]), "+v"(blo[0]), "+v"(bhi[0]));
            const bf16x8 af0 = AAU_FRAG8(alo[0], ahi[0]), af1 = AAU_FRAG8(alo[1], ahi[1]), af2 = AAU_FRAG8(alo[2], ahi[2]);
            {
                const bf16x8 bf = AAU_FRAG8(blo[0], bhi[0]);
                acc[0][0] = AAU_MFMA16(af0, bf, acc[0][0], 0, 0, 0);
                acc[1][0] = AAU_MFMA16(af1, bf, acc[1][0], 0, 0, 0);
                acc[2][0] = AAU_MFMA16(af2, bf, acc[2][0], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]));
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                const bf16x8 bf = AAU_FRAG8(blo[j], bhi[j]);
                acc[0][j] = AAU_MFMA16(af0, bf, acc[0][j], 0, 0, 0);
                acc[1][j] = AAU_MFMA16(af1, bf, acc[1][j], 0, 0, 0);
                acc[2][j] = AAU_MFMA16(af2, bf, acc[2][j], 0, 0, 0);
            }
        }
    };

    const bool bnz = a.dz_scale != nullptr;
    if (bnz) {
        for (int i = tid; i < 96; i += 256) {
            const bool in = i < 48 * TQ && q0 + i < d.Cout;
            s_dz[i] = in ? a.dz_scale[q0 + i] : 0.f;
            s_dz[96 + i] = in ? a.dz_shift[q0 + i] : 0.f;
        }
        __syncthreads();
    }
    // this thread's own pieces of the dz tile in buffer `buf` (they have landed: the vmcnt(0) in front of every call)
    auto xform_y = [&](int buf) {
        if (!bnz) return;
#pragma unroll
        for (int i = 0; i < NLY; ++i) {
            u32x4* pz = (u32x4*)(sY(buf) + (256 * i + tid) * 16);
            const float* t = s_dz + ych[i];
            float f[8], sc[8], sh[8];
            *(f32x4*)(sc) = *(const f32x4*)(t); *(f32x4*)(sc + 4) = *(const f32x4*)(t + 4);
            *(f32x4*)(sh) = *(const f32x4*)(t + 96); *(f32x4*)(sh + 4) = *(const f32x4*)(t + 100);
            unpack8(*pz, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
            *pz = pack8(f);
        }
    };
    int mbase = mb;
    stage(0, mbase);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xform_y(0);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int mnext = mbase + BKP;
        const bool more = mnext < me;
        if (more) {
            advance();
            stage(buf ^ 1, mnext);
        }
        compute(buf);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform_y(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        mbase = mnext;
    }

    // acc[i][j][r] = D[q = q0 + wq*48 + i*16 + 4*g16 + r][c = c0 + wc*48 + j*16 + li]
    if (a.ws) {
        // split-K partial: the accumulators leave in register layout (16 B per lane, 4 KiB per instruction and
        // workgroup); wg_reduce_kernel sums the slabs of a tile in a fixed order -> bitwise reproducible dw
        float* slab = a.ws + (int64_t)lbid * (9 * 256 * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) *(f32x4*)(slab + ((i * 3 + j) * 256 + tid) * 4) = acc[i][j];
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = c0 + wc * 48 + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + wq * 48 + i * 16 + 4 * g16 + r;
                if (q < d.Cout && c < d.Cin)
                    WG_ADD(a.dw + ((int64_t)q * T + tap) * d.Cin + c, acc[i][j][r]);
            }
        }
}

// Sum of the split-K slabs of wgrad_kernel (MODE 0) / wgrad3x3_kernel (MODE 1), added into dw.
// One thread owns one accumulator vector (tile, v, source lane) and walks over its (split, K-wave) terms in a
// fixed order; P threads share the walk (terms p, p+P, ...) and are combined through LDS in part order.

template <int MODE>
__global__ __launch_bounds__(256) void wg_reduce_kernel(const WRedArgs a) {
    __shared__ f32x4 sm[256];
    const int VPB = 256 / a.P;
    const int part = threadIdx.x / VPB, sl = threadIdx.x - part * VPB;
    const int64_t slot = (int64_t)blockIdx.x * VPB + sl;
    const bool ok = slot < a.nslots;
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
    int lane = 0, s = 0, v = 0, tile = 0;
    if (ok) {
        if constexpr (MODE == 0) {
            lane = (int)(slot & 63);
            int64_t rest = slot >> 6;
            s = (int)(rest % a.sub); rest /= a.sub;
            v = (int)(rest % 9);
            tile = (int)(rest / 9);
            const int nterms = a.nsplit * a.kwaves;
            auto term = [&](int t) -> f32x4 {
                const int split = t / a.kwaves, wk = t - split * a.kwaves;
                return *(const f32x4*)(a.ws + ((((int64_t)tile * a.nsplit + split) * 9 + v) * 256 + (wk * a.sub + s) * 64 + lane) * 4);
            };
            int t = part;
            for (; t + 3 * a.P < nterms; t += 4 * a.P) {
                const f32x4 t0 = term(t), t1 = term(t + a.P), t2 = term(t + 2 * a.P), t3 = term(t + 3 * a.P);
                sum += t0; sum += t1; sum += t2; sum += t3;
            }
            for (; t < nterms; t += a.P) sum += term(t);
        } else {   // MODE 1 / 2
            lane = (int)(slot & 255);          // source thread id
            int64_t rest = slot >> 8;
            v = (int)(rest % a.NV);
            tile = (int)(rest / a.NV);
            // 4 independent loads in flight per thread; the order of the additions stays fixed
            const float* p0 = a.ws + (((int64_t)tile * a.nsplit * a.NV + v) * 256 + lane) * 4;
            const int64_t sstride = (int64_t)a.NV * 256 * 4;
            int split = part;
            for (; split + 3 * a.P < a.nsplit; split += 4 * a.P) {
                const f32x4 t0 = *(const f32x4*)(p0 + (int64_t)split * sstride);
                const f32x4 t1 = *(const f32x4*)(p0 + (int64_t)(split + a.P) * sstride);
                const f32x4 t2 = *(const f32x4*)(p0 + (int64_t)(split + 2 * a.P) * sstride);
                const f32x4 t3 = *(const f32x4*)(p0 + (int64_t)(split + 3 * a.P) * sstride);
                sum += t0; sum += t1; sum += t2; sum += t3;
            }
            for (; split < a.nsplit; split += a.P) sum += *(const f32x4*)(p0 + (int64_t)split * sstride);
        }
    }
    if (a.P > 1) {
        sm[threadIdx.x] = sum;
        __syncthreads();
        if (part != 0) return;
        for (int p = 1; p < a.P; ++p) sum += sm[p * VPB + sl];
    }
    if (!ok) return;
    if constexpr (MODE == 0) {
        const int tap = tile % a.T;
        const int t2 = tile / a.T;
        const int tc = t2 % a.ntc, tq = t2 / a.ntc;
        const int wq = (a.TQ == 2) ? ((a.TC == 2) ? (s >> 1) : s) : 0;
        const int wc = (a.TC == 2) ? ((a.TQ == 2) ? (s & 1) : s) : 0;
        const int i = v / 3, j = v - i * 3;
        const int g16 = lane >> 4, li = lane & 15;
        const int c = tc * 48 * a.TC + wc * 48 + j * 16 + li;
        const int qb = tq * 48 * a.TQ + wq * 48 + i * 16 + 4 * g16;
        if (c < a.Cin)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (qb + r < a.Cout) a.dw[((int64_t)(qb + r) * a.T + tap) * a.Cin + c] += sum[r];
    } else if constexpr (MODE == 2) {   // wgrad3x3r_kernel<QT = TQ, CJ = TC>: v = (i*3 + tx)*3 + ty
        const int wave = lane >> 6, l = lane & 63;
        const int jw = wave % a.TC, qg = wave / a.TC;
        const int i = v / 9, tx = (v / 3) % 3, ty = v % 3;
        const int tc = tile % a.ntc, tq = tile / a.ntc;
        const int g16 = l >> 4, li = l & 15;
        const int c = tc * 16 * a.TC + jw * 16 + li;
        const int qb = tq * a.TQ * 16 + (qg * 3 + i) * 16 + 4 * g16;
        if (c < a.Cin)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (qb + r < a.Cout) a.dw[((int64_t)(qb + r) * 9 + (ty * 3 + tx)) * a.Cin + c] += sum[r];
    } else {
        const int QT = a.NV / 7;
        const int wave = lane >> 6, l = lane & 63;
        const int i = v / 7, n = v - i * 7;
        if (n >= (wave < 3 ? 7 : 6)) return;
        const int ct = 7 * wave + n;
        const int tap = ct / 3, j = ct - tap * 3;
        const int tc = tile % a.ntc, tq = tile / a.ntc;
        const int g16 = l >> 4, li = l & 15;
        const int c = tc * 48 + j * 16 + li;
        const int qb = tq * QT * 16 + i * 16 + 4 * g16;
        if (c < a.Cin)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (qb + r < a.Cout) a.dw[((int64_t)(qb + r) * 9 + tap) * a.Cin + c] += sum[r];
    }
}

int wg_reduce_launch(int mode, WRedArgs& r, hipStream_t s) {
    // enough threads to pull the slabs at memory speed: share each vector's walk among P threads when the
    // tile count is small (level-1 layers: ONE tile, 512 splits)
    int P = 1;
    while (P < 16 && r.nslots * P < 65536 && 2 * P <= r.nsplit * (mode == 0 ? r.kwaves : 1)) P *= 2;
    r.P = P;
    const int VPB = 256 / P;
    const int64_t grid = (r.nslots + VPB - 1) / VPB;
    if (mode == 0) hipLaunchKernelGGL((wg_reduce_kernel<0>), dim3((unsigned)grid), dim3(256), 0, s, r);
    else if (mode == 2) hipLaunchKernelGGL((wg_reduce_kernel<2>), dim3((unsigned)grid), dim3(256), 0, s, r);
    else hipLaunchKernelGGL((wg_reduce_kernel<1>), dim3((unsigned)grid), dim3(256), 0, s, r);
    return check_launch("aau_conv_wgrad(split-K reduce)");
}

// wgrad3x3.hip
bool wgrad3x3_applicable(const aau_conv_desc* d);
int wgrad3x3_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                    int64_t ws_bytes, int64_t* need, hipStream_t s, const float* in_scale = nullptr, const float* in_shift = nullptr);

// wgrad3x3r.hip
int wgrad3x3r_variant(const aau_conv_desc* d);
int wgrad3x3r_launch(int variant, const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                     int64_t ws_bytes, int64_t* need, hipStream_t s, const float* in_scale = nullptr, const float* in_shift = nullptr);

template <int TQ, int TC>
static int launch(WgradArgs& a, float* ws, int64_t ws_bytes, int64_t* need, hipStream_t s) {
    constexpr int KWAVES = 4 / (TQ * TC);
    constexpr int KSUBW = (TQ * TC == 1) ? 1 : 2;
    constexpr int BKP = 32 * KWAVES * KSUBW;
    const aau_conv_desc& d = a.d;
    const int T = d.KH * d.KW;
    const int ntq = (d.Cout + 48 * TQ - 1) / (48 * TQ), ntc = (d.Cin + 48 * TC - 1) / (48 * TC);
    const int64_t tiles = (int64_t)ntq * ntc * T;
    // Split-K: each split ends with a tile of partial sums.  With fp32 atomics (memory side, ~1.3 TB/s chip-wide,
    // slower still when many workgroups hit the same few rows) the split count is a trade against occupancy,
    // measured per shape on one device: 1x1 / 2x2 problems -50 % time going from 2048 to 512 workgroups (256 when
    // the whole matrix is one or two tiles); dilated 3x3 keeps 2048.
    int64_t tgt = T >= 9 ? 2048 : (T == 1 && tiles <= 8 ? 256 : 512);
    if (const char* e = getenv("AAU_WG_TARGET")) tgt = atoi(e);   // experiment
    int64_t want = (tgt + tiles - 1) / tiles;
    int64_t maxsplit = (a.M + 4 * BKP - 1) / (4 * BKP);
    int64_t nsplit = want < 1 ? 1 : (want > maxsplit ? maxsplit : want);
    if (nsplit < 1) nsplit = 1;
    int64_t pps = (a.M + nsplit - 1) / nsplit;
    pps = (pps + BKP - 1) / BKP * BKP;
    nsplit = (a.M + pps - 1) / pps;
    a.pix_per_split = (int)pps;
    a.nsplit = (int)nsplit;
    const int64_t grid = tiles * nsplit;
    if (grid > 0x7fffffff) { set_error("aau_conv_wgrad: grid too large"); return AAU_E_INVALID; }
    const int64_t bytes = grid * (9 * 256 * 4) * (int64_t)sizeof(float);
    if (need) { *need = bytes; return AAU_OK; }
    if (ws && ws_bytes < bytes) {
        set_error("aau_conv_wgrad: workspace of %lld B, need %lld B (aau_conv_wgrad_ws_bytes)", (long long)ws_bytes, (long long)bytes);
        return AAU_E_INVALID;
    }
    a.ws = ws;
    a.rev = next_traversal();
    if (a.dz_scale) {     // the transform needs the fast issue path in EVERY workgroup (no zero-page rows in the dz tile)
        const bool g2 = !a.linear && d.KH == 2 && d.KW == 2 && d.stride == 2 && d.pad == 0 && d.dil == 1 && d.H == 2 * d.Ho &&
                        d.W == 2 * d.Wo && d.Wo % BKP == 0;
        if (!(a.fast_ok && (a.linear || g2) && a.M % BKP == 0)) {
            set_error("aau_conv_wgrad_bnin_dz: descriptor not served (aau_conv_wgrad_bnin_dz_ok)");
            return AAU_E_INVALID;
        }
    }
    {
        char tag[AAU_PROF_TAG_LEN];
        snprintf(tag, sizeof(tag), "wgrad<%d,%d>%s", TQ, TC, T > 1 ? (d.dil > 1 ? " dilated" : " taps") : "");
        prof_tag(tag);
    }
    hipLaunchKernelGGL((wgrad_kernel<TQ, TC>), dim3((unsigned)grid), dim3(256), 0, s, a);
    if (!ws) return check_launch("aau_conv_wgrad");
    WRedArgs r;
    r.ws = ws; r.dw = a.dw;
    r.nsplit = a.nsplit; r.NV = 9; r.sub = TQ * TC; r.kwaves = KWAVES;
    r.TQ = TQ; r.TC = TC; r.ntc = ntc; r.T = T; r.Cout = d.Cout; r.Cin = d.Cin;
    r.nslots = tiles * 9 * (TQ * TC) * 64;
    return wg_reduce_launch(0, r, s);
}

}  // namespace aau

// in_scale / in_shift (aau_conv_wgrad_bnin): src is a raw conv output, the kernel applies relu(src * scale + shift) on it
static int wgrad_dispatch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                          int64_t ws_bytes, int64_t* need, void* stream, const float* in_scale = nullptr,
                          const float* in_shift = nullptr, const float* dz_scale = nullptr, const float* dz_shift = nullptr) {
    using namespace aau;
    AAU_REQUIRE(d, "aau_conv_wgrad: null descriptor");
    AAU_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0 && d->Cout % 8 == 0,
                "aau_conv_wgrad: Cin=%d / Cout=%d must be positive multiples of 8", d->Cin, d->Cout);
    AAU_REQUIRE(d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0, "aau_conv_wgrad: pitches must be multiples of 8");
    AAU_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= 16, "aau_conv_wgrad: taps %dx%d", d->KH, d->KW);
    AAU_REQUIRE((int64_t)d->N * d->H * d->W < 0x7fffffff && (int64_t)d->N * d->Ho * d->Wo < 0x7fffffff,
                "aau_conv_wgrad: pixel count overflows int32");
    if (!need) {
        AAU_REQUIRE(src && dz && dw, "aau_conv_wgrad: null pointer");
        AAU_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dz & 15) == 0 && ((uintptr_t)ws & 15) == 0,
                    "aau_conv_wgrad: 16-byte alignment");
    }
    WgradArgs a;
    a.d = *d;
    a.src = src; a.dz = dz; a.dw = dw; a.ws = nullptr;
    a.dz_scale = dz_scale; a.dz_shift = dz_shift;
    if (dz_scale) AAU_REQUIRE(aau_conv_wgrad_bnin_dz_ok(d), "aau_conv_wgrad_bnin_dz: descriptor not served (aau_conv_wgrad_bnin_dz_ok)");
    a.M = d->N * d->Ho * d->Wo;
    a.linear = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo);
    {
        const int64_t sbytes = (((int64_t)d->N * d->H * d->W - 1) * d->src_pitch + d->Cin) * 2;
        const int64_t zbytes = (((int64_t)a.M - 1) * d->dst_pitch + d->Cout) * 2;
        a.fast_ok = sbytes < 0x7fffffff && zbytes < 0x7fffffff && !getenv("AAU_WG_NOFAST");
        a.src_bytes = (unsigned)(a.fast_ok ? sbytes : 0);
        a.dz_bytes = (unsigned)(a.fast_ok ? zbytes : 0);
    }
    const bool split = d->src_split_c > 0 || d->dst_split_c > 0;
    if (split) {
        AAU_REQUIRE(wgrad3x3_applicable(d) && d->dst_split_c <= 0, "aau_conv_wgrad: a two-plane source is only served by the all-taps 3x3 kernel (aau_conv_split_ok)");
        AAU_REQUIRE(d->src_split_c % 8 == 0 && d->src_split_c < d->Cin && d->src_split_off % 8 == 0 && d->src_split_off >= 0,
                    "aau_conv_wgrad: src_split_c / src_split_off must be multiples of 8 inside the channel range");
    }
    if (in_scale)
        AAU_REQUIRE(aau_conv_wgrad_bnin_ok(d), "aau_conv_wgrad_bnin: descriptor not served (aau_conv_wgrad_bnin_ok)");
    if (wgrad3x3_applicable(d)) {
        const int rv = split ? 0 : wgrad3x3r_variant(d);
        if (need) return rv ? wgrad3x3r_launch(rv, d, src, dz, dw, ws, ws_bytes, need, (hipStream_t)stream)
                            : wgrad3x3_launch(d, src, dz, dw, ws, ws_bytes, need, (hipStream_t)stream);
        const double flops = 2.0 * a.M * (double)d->Cout * d->Cin * d->KH * d->KW;
        ProfScope prof(1, flops, (hipStream_t)stream);
        if (rv) {
            prof_tag(rv == 1 ? "wgrad3x3r<3,4>" : "wgrad3x3r<6,2>",
                     2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)a.M * d->Cout) + 4.0 * d->Cout * 9.0 * d->Cin);
            return wgrad3x3r_launch(rv, d, src, dz, dw, ws, ws_bytes, nullptr, (hipStream_t)stream, in_scale, in_shift);
        }
        prof_tag("wgrad3x3<3,8>", 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)a.M * d->Cout) + 4.0 * d->Cout * 9.0 * d->Cin);
        return wgrad3x3_launch(d, src, dz, dw, ws, ws_bytes, nullptr, (hipStream_t)stream, in_scale, in_shift);
    }
    const bool q2 = d->Cout > 48, c2 = d->Cin > 48;
    if (need) {
        if (q2 && c2) return launch<2, 2>(a, ws, ws_bytes, need, (hipStream_t)stream);
        if (q2) return launch<2, 1>(a, ws, ws_bytes, need, (hipStream_t)stream);
        if (c2) return launch<1, 2>(a, ws, ws_bytes, need, (hipStream_t)stream);
        return launch<1, 1>(a, ws, ws_bytes, need, (hipStream_t)stream);
    }
    const double flops = 2.0 * a.M * (double)d->Cout * d->Cin * d->KH * d->KW;
    ProfScope prof(1, flops, (hipStream_t)stream);
    prof_tag(nullptr, 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)a.M * d->Cout) + 4.0 * d->Cout * (double)(d->KH * d->KW) * d->Cin);
    if (q2 && c2) return launch<2, 2>(a, ws, ws_bytes, nullptr, (hipStream_t)stream);
    if (q2) return launch<2, 1>(a, ws, ws_bytes, nullptr, (hipStream_t)stream);
    if (c2) return launch<1, 2>(a, ws, ws_bytes, nullptr, (hipStream_t)stream);
    return launch<1, 1>(a, ws, ws_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int aau_conv_wgrad(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, float* ws,
                              int64_t ws_bytes, void* stream) {
    return wgrad_dispatch(d, src, dz, dw, ws, ws_bytes, nullptr, stream);
}

extern "C" int aau_conv_wgrad_ws_bytes(const aau_conv_desc* d, int64_t* bytes) {
    AAU_REQUIRE(bytes, "aau_conv_wgrad_ws_bytes: null pointer");
    return wgrad_dispatch(d, nullptr, nullptr, nullptr, nullptr, 0, bytes, nullptr);
}

// 1 when aau_conv_wgrad_bnin_dz serves this descriptor: the generic kernel's fast issue path in every workgroup
extern "C" int aau_conv_wgrad_bnin_dz_ok(const aau_conv_desc* d) {
    using namespace aau;
    if (!d || getenv("AAU_NO_BNIN") || getenv("AAU_WG_NOFAST") || wgrad3x3_applicable(d)) return 0;
    const bool lin = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
    const bool g2 = d->KH == 2 && d->KW == 2 && d->stride == 2 && d->pad == 0 && d->dil == 1 && d->H == 2 * d->Ho && d->W == 2 * d->Wo &&
                    d->Wo % 128 == 0;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    const int64_t sbytes = (((int64_t)d->N * d->H * d->W - 1) * d->src_pitch + d->Cin) * 2, zbytes = ((M - 1) * d->dst_pitch + d->Cout) * 2;
    return (lin || g2) && M % 128 == 0 && sbytes < 0x7fffffff && zbytes < 0x7fffffff && d->src_split_c <= 0 && d->dst_split_c <= 0 &&
           d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0;
}

extern "C" int aau_conv_wgrad_bnin_dz(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, const float* dz_scale,
                                      const float* dz_shift, float* dw, float* ws, int64_t ws_bytes, void* stream) {
    AAU_REQUIRE(dz_scale && dz_shift, "aau_conv_wgrad_bnin_dz: null pointer");
    return wgrad_dispatch(d, src, dz, dw, ws, ws_bytes, nullptr, stream, nullptr, nullptr, dz_scale, dz_shift);
}

// 1 when aau_conv_wgrad_bnin serves this descriptor (the all-taps 3x3 kernel, one source plane)
extern "C" int aau_conv_wgrad_bnin_ok(const aau_conv_desc* d) {
    using namespace aau;
    if (!d || getenv("AAU_NO_BNIN")) return 0;
    const int rv = wgrad3x3r_variant(d);        // 0: wgrad3x3 (48 x 48 tiles), 2: the 96 x 32 row-reuse tiling
    return wgrad3x3_applicable(d) && d->src_split_c <= 0 && d->dst_split_c <= 0 && (rv == 0 || rv == 2) &&
           d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0;
}

// dw += weight gradient with x = relu(src * in_scale + in_shift) as the convolution's input: aau_bn_act followed by
// aau_conv_wgrad, bit for bit, without the activation in memory (the counterpart of aau_conv_igemm_bnin)
extern "C" int aau_conv_wgrad_bnin(const aau_conv_desc* d, const aau_bf16* src, const float* in_scale, const float* in_shift,
                                   const aau_bf16* dz, float* dw, float* ws, int64_t ws_bytes, void* stream) {
    AAU_REQUIRE(in_scale && in_shift, "aau_conv_wgrad_bnin: null pointer");
    return wgrad_dispatch(d, src, dz, dw, ws, ws_bytes, nullptr, stream, in_scale, in_shift);
}

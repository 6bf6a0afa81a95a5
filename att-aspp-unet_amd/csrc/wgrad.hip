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
    int bid = blockIdx.x;
    const int split = bid % a.nsplit;
    bid /= a.nsplit;
    const int tap = bid % T;
    bid /= T;
    const int tc = bid % ntc;
    const int tq = bid / ntc;
    const int q0 = tq * 48 * TQ, c0 = tc * 48 * TC;

    const int mb = split * a.pix_per_split;
    const int me = min(a.M, mb + a.pix_per_split);
    if (mb >= me) return;  // uniform

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

    auto stage = [&](int buf, int mbase) {
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
        if (a.linear) return;
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
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < KSUBW; ++ks) {
            const int rbase = (wk * KSUBW + ks) * 32 + 4 * g16 + rq;
            bf16x8 af[3], bf[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ch = wq * 48 + i * 16 + cp;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sY(buf) + tile_off<TQ>(rbase, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sY(buf) + tile_off<TQ>(rbase + 16, ch)));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ch = wc * 48 + j * 16 + cp;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sX(buf) + tile_off<TC>(rbase, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sX(buf) + tile_off<TC>(rbase + 16, ch)));
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    int mbase = mb;
    stage(0, mbase);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        __syncthreads();
        buf ^= 1;
        mbase = mnext;
    }

    // acc[i][j][r] = D[q = q0 + wq*48 + i*16 + 4*g16 + r][c = c0 + wc*48 + j*16 + li]
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

// wgrad3x3.hip
bool wgrad3x3_applicable(const aau_conv_desc* d);
int wgrad3x3_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, float* dw, hipStream_t s);

template <int TQ, int TC>
static int launch(WgradArgs& a, hipStream_t s) {
    constexpr int KWAVES = 4 / (TQ * TC);
    constexpr int KSUBW = (TQ * TC == 1) ? 1 : 2;
    constexpr int BKP = 32 * KWAVES * KSUBW;
    const aau_conv_desc& d = a.d;
    const int T = d.KH * d.KW;
    const int64_t tiles = (int64_t)((d.Cout + 48 * TQ - 1) / (48 * TQ)) * ((d.Cin + 48 * TC - 1) / (48 * TC)) * T;
    // Split-K: each split ends with a tile of fp32 atomics (memory side, ~1.3 TB/s chip-wide, slower still when
    // many workgroups hit the same few rows), so the split count is a trade against occupancy.  Measured per
    // shape on one device: 1x1 / 2x2 problems (K-steps are cheap, atomics dominate: -50 % time going from 2048
    // to 512 workgroups, 256 when the whole matrix is one or two tiles); dilated 3x3 keeps 2048.
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
    hipLaunchKernelGGL((wgrad_kernel<TQ, TC>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return check_launch("aau_conv_wgrad");
}

}  // namespace aau

extern "C" int aau_conv_wgrad(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz,
                              float* dw, void* stream) {
    using namespace aau;
    AAU_REQUIRE(d && src && dz && dw, "aau_conv_wgrad: null pointer");
    AAU_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0 && d->Cout % 8 == 0,
                "aau_conv_wgrad: Cin=%d / Cout=%d must be positive multiples of 8", d->Cin, d->Cout);
    AAU_REQUIRE(d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0, "aau_conv_wgrad: pitches must be multiples of 8");
    AAU_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= 16, "aau_conv_wgrad: taps %dx%d", d->KH, d->KW);
    AAU_REQUIRE((int64_t)d->N * d->H * d->W < 0x7fffffff && (int64_t)d->N * d->Ho * d->Wo < 0x7fffffff,
                "aau_conv_wgrad: pixel count overflows int32");
    AAU_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dz & 15) == 0, "aau_conv_wgrad: 16-byte alignment");
    WgradArgs a;
    a.d = *d;
    a.src = src; a.dz = dz; a.dw = dw;
    a.M = d->N * d->Ho * d->Wo;
    a.linear = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H == d->Ho && d->W == d->Wo);
    const double flops = 2.0 * a.M * (double)d->Cout * d->Cin * d->KH * d->KW;
    ProfScope prof(1, flops, (hipStream_t)stream);
    if (wgrad3x3_applicable(d)) return wgrad3x3_launch(d, src, dz, dw, (hipStream_t)stream);
    const bool q2 = d->Cout > 48, c2 = d->Cin > 48;
    if (q2 && c2) return launch<2, 2>(a, (hipStream_t)stream);
    if (q2) return launch<2, 1>(a, (hipStream_t)stream);
    if (c2) return launch<1, 2>(a, (hipStream_t)stream);
    return launch<1, 1>(a, (hipStream_t)stream);
}

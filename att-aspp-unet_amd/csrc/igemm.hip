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
#define AAU_IG_BLK ((int)blockIdx.x)
#define AAU_IG_NBLK ((int)gridDim.x)
#include "igemm_body.inc"
#undef AAU_IG_BLK
#undef AAU_IG_NBLK
}

// Several independent wide-tile problems in ONE grid (aau_conv_igemm_multi): the four spatial branches of the ASPP bridge
// (pipeline:80-83) read the same input but have their own weights, dilation, output slice and statistics; as four launches
// each is exactly one workgroup per CU with its own ramp, prologue and tail.  Problem i owns blocks [begin[i], begin[i+1]).
constexpr int IGEMM_MULTI_MAX = 4;
struct IgemmMulti {
    IgemmArgs p[IGEMM_MULTI_MAX];
    int begin[IGEMM_MULTI_MAX + 1];
    int n;
};
template <int BK, int BQ>
__global__ __launch_bounds__(512) void igemm_multi_kernel(const IgemmMulti g) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < IGEMM_MULTI_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.begin[i]) pi = i;
    const IgemmArgs a = g.p[pi];          // by value: scalar loads once, then SGPRs
    const int mblk = (int)blockIdx.x - g.begin[pi], mnblk = g.begin[pi + 1] - g.begin[pi];
    constexpr bool SMALL = false;
#define AAU_IG_BLK mblk
#define AAU_IG_NBLK mnblk
#include "igemm_body.inc"
#undef AAU_IG_BLK
#undef AAU_IG_NBLK
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
                        hipStream_t s, const float* in_scale = nullptr, const float* in_shift = nullptr);

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

// argument checks + the argument block of one problem (shared by aau_conv_igemm and aau_conv_igemm_multi)
static int conv_prepare(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                        const float* scale, const float* shift, float* stats, aau::IgemmArgs& a) {
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
#if !defined(AAU_IGEMM_STAMP) && !defined(AAU_PW_STAMP)      /* the diagnostic builds take their stamp buffer through `shift` */
    AAU_REQUIRE((scale == nullptr) == (shift == nullptr), "aau_conv_igemm: scale and shift come together");
#endif
#endif
    AAU_REQUIRE(!d->shuffle2x2 || (d->Cout % 32 == 0), "aau_conv_igemm: shuffle2x2 needs Cout %% 32 == 0");
    AAU_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)wpk & 15) == 0 && ((uintptr_t)dst & 7) == 0,
                "aau_conv_igemm: pointers must be 16-byte (src, wpk) / 8-byte (dst) aligned");
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
    a.nchunk = d->Cpad / (d->Cpad % 64 == 0 ? 64 : 32);
    a.rev = 0;
    return AAU_OK;
}

// the wide 128 x 192 tile serves this problem (long K, channel count in whole 192-wide tiles, a workgroup per CU)
static bool conv_wide_ok(const aau_conv_desc* d, const aau::IgemmArgs& a) {
    const bool bk64 = (d->Cpad % 64 == 0);
    const int64_t tiles192 = (int64_t)((a.M + 127) / 128) * (d->Cout / 192);
    bool wide = bk64 && d->Cout % 192 == 0 && tiles192 >= 224 && a.nchunk * d->KH * d->KW >= 6;
    if (const char* e = getenv("AAU_IGEMM_WIDE")) wide = bk64 && d->Cout % 192 == 0 && atoi(e) != 0;
    return wide;
}

static int conv_dispatch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst,
                         const float* bias, const float* scale, const float* shift, float* stats, void* stream) {
    using namespace aau;
    IgemmArgs a;
    if (const int rc = conv_prepare(d, src, wpk, dst, bias, scale, shift, stats, a)) return rc;
    const bool bk64 = (d->Cpad % 64 == 0);
    const double flops = 2.0 * a.M * (double)d->Cout * d->Cin * d->KH * d->KW;
    ProfScope prof(0, flops, (hipStream_t)stream);
    // algorithmic HBM bytes: every input / output element and every weight once (bf16)
    prof_tag(nullptr, 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)a.M * d->Cout * (d->accumulate ? 2 : 1) +
                             (double)d->Cout * d->KH * d->KW * d->Cin));
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
    if (conv_wide_ok(d, a)) {
        if (const char* e = getenv("AAU_IGEMM_ABL")) a.rev |= atoi(e) & 14;   // timing ablation, results are wrong
        return launch<64, 192>(a, (hipStream_t)stream);
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

// 1 when aau_conv_igemm_multi serves these n problems: each of them alone would take the wide tile
extern "C" int aau_conv_igemm_multi_ok(const aau_conv_desc* descs, int n) {
    using namespace aau;
    if (!descs || n < 2 || n > IGEMM_MULTI_MAX || getenv("AAU_NO_IGEMM_MULTI")) return 0;
    for (int i = 0; i < n; ++i) {
        const aau_conv_desc* d = &descs[i];
        if (d->Cin <= 0 || d->Cout <= 0 || d->Cpad % 32 || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
        if (conv3x3_applicable(d) || conv1x1_resw_applicable(d, true) || d->src_split_c > 0 || d->dst_split_c > 0) return 0;
        IgemmArgs a;
        a.M = d->N * d->Ho * d->Wo;
        a.nchunk = d->Cpad / (d->Cpad % 64 == 0 ? 64 : 32);
        if (!conv_wide_ok(d, a)) return 0;
    }
    return 1;
}

extern "C" int aau_conv_igemm_multi(const aau_conv_desc* descs, const aau_bf16* const* srcs, const aau_bf16* const* wpks,
                                    aau_bf16* const* dsts, aau_stat* const* stats, const int64_t* stats_bytes, int n, void* stream) {
    using namespace aau;
    AAU_REQUIRE(descs && srcs && wpks && dsts && aau_conv_igemm_multi_ok(descs, n),
                "aau_conv_igemm_multi: null pointer, or the problems are not served (aau_conv_igemm_multi_ok, n=%d)", n);
    IgemmMulti g;
    g.n = n;
    int blocks = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const aau_conv_desc* d = &descs[i];
        aau_stat* st = stats ? stats[i] : nullptr;
        AAU_CHECK_STAT("aau_conv_igemm_multi", st, stats_bytes ? stats_bytes[i] : 0, d->Cout);
        if (const int rc = conv_prepare(d, srcs[i], wpks[i], dsts[i], nullptr, nullptr, nullptr, (float*)st, g.p[i])) return rc;
        g.p[i].rev = next_traversal();
        g.begin[i] = blocks;
        blocks += ((g.p[i].M + 127) / 128) * (d->Cout / 192);
        flops += 2.0 * g.p[i].M * (double)d->Cout * d->Cin * d->KH * d->KW;
        bytes += 2.0 * ((double)d->N * d->H * d->W * d->Cin + (double)g.p[i].M * d->Cout + (double)d->Cout * d->KH * d->KW * d->Cin);
    }
    for (int i = n; i <= IGEMM_MULTI_MAX; ++i) g.begin[i] = blocks;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(0, flops, s);
    prof_tag("igemm<64,192,0> multi", bytes);
    constexpr size_t lds = (size_t)2 * (192 + 128) * 64 * 2;
    static bool attr = false;
    if (!attr) {
        AAU_REQUIRE(hipFuncSetAttribute((const void*)igemm_multi_kernel<64, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                    "aau_conv_igemm_multi: cannot reserve %zu bytes of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL((igemm_multi_kernel<64, 192>), dim3((unsigned)blocks), dim3(512), lds, s, g);
    return check_launch("aau_conv_igemm_multi");
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

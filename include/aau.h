/*
 * aau.h -- C ABI of the MI355X (gfx950) Attention-ASPP-UNet hot-path library.
 *
 * The reference (vivi-git188/ATT-ASPP-UNET) is pure Python on PyTorch: it has no FFI
 * of its own, its "operator interface" for this path is the nn.Module / function
 * surface of attention_aspp_unet_pipeline_stage.py (abbreviated "pipeline" below).
 * Every entry point here names the reference call site whose arithmetic it replaces.
 * The host side (the *.py files of att-aspp-unet_amd/) mirrors that Python surface and reaches this
 * library through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Device pointers unless stated.
 *   - activations: NHWC, bf16 ("u16" bit patterns), a pixel's channels contiguous;
 *     every tensor argument carries a pixel pitch (elements between two pixels) so a
 *     channel slice of a wider concat buffer is addressable without a copy.
 *   - parameters / gradients / statistics: fp32.  Conv weights are physically
 *     [Cout][KH][KW][Cin] (the channels_last image of PyTorch's OIHW tensor);
 *     ConvTranspose2d weights physically [Cin][KH][KW][Cout] (channels_last of IOHW).
 *   - ownership: the caller owns every buffer; the library never allocates, frees or
 *     keeps a pointer after the call returns.  No call synchronises the host: all
 *     work is enqueued on `stream` (a hipStream_t passed as void*), so every entry
 *     point is hipGraph-capturable.
 *   - return value: 0 on success, negative on error (AAU_E_*); aau_last_error()
 *     returns a thread-local message.  Nothing throws across the ABI.
 *   - threading: re-entrant; safe to call from the autograd worker thread.
 */
#ifndef AAU_H_
#define AAU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AAU_OK 0
#define AAU_E_INVALID (-1)     /* bad argument / unsupported shape */
#define AAU_E_HIP (-2)         /* HIP runtime error at launch */

#define AAU_STAT_REPLICAS 32   /* statistic accumulators are spread over this many replicas (same-address atomics) */

typedef uint16_t aau_bf16;
/* Per-channel batch statistics (sum, sum of squares) accumulated ORDER-INDEPENDENTLY: int64              */
/* [AAU_STAT_REPLICAS][2][C][2 limbs] followed by one poison word (+1 pad); value = hi*2^-8 + lo*2^-52, added  */
/* with 64-bit integer atomics, so the totals are the same bits for every arrival order (BatchNorm batch      */
/* statistics no longer differ from run to run).  The caller zeroes the buffer (AAU_STAT_WORDS(C) words).     */
/* Every entry point that takes an aau_stat buffer also takes its size in BYTES right behind the pointer and    */
/* refuses a buffer smaller than AAU_STAT_WORDS(C) * 8 for the channel count of the call: the epilogues' 64-bit */
/* atomics reach up to the poison word, so an undersized buffer would be an out-of-bounds device write.         */
typedef int64_t aau_stat;
#define AAU_STAT_WORDS(C) ((size_t)AAU_STAT_REPLICAS * 2 * (size_t)(C) * 2 + 2)

const char* aau_last_error(void);
int aau_version(void);

/* ---- launch profiling (used by bench.py for the live roofline figure) ------------- */
/* While enabled every kernel launch is bracketed by hipEvents on its own stream.      */
/* families: 0 igemm (fwd+dgrad), 1 wgrad, 2 elementwise/reduction, 3 optimizer+loss   */
#define AAU_PROF_FAMILIES 4
int aau_prof_enable(int on);
/* Synchronises the recorded events (host sync: never call inside graph capture),     */
/* returns per-family summed kernel time [ms], launch counts and algorithmic FLOPs,   */
/* then clears the records.                                                           */
int aau_prof_collect(double ms[AAU_PROF_FAMILIES], int64_t launches[AAU_PROF_FAMILIES],
                     double flops[AAU_PROF_FAMILIES]);

/* Per-launch form of the same records (clears them as well): up to cap launches in issue    */
/* order with the kernel variant the launcher picked (tags: cap x AAU_PROF_TAG_LEN chars),    */
/* event time [ms], algorithmic FLOPs and algorithmic HBM bytes (0 where not stated).         */
/* tags[i] = "<caller label>|<kernel variant>"; the label is whatever aau_prof_label() set on   */
/* the calling thread before the launch (the engine passes its layer names), consumed by it.  */
#define AAU_PROF_TAG_LEN 96
int aau_prof_label(const char* label);
int aau_prof_collect_launches(int cap, int* n_out, char* tags, double* ms, double* flops,
                              double* bytes, int* family);

/* ---- implicit-GEMM convolution on MFMA --------------------------------------------- */
/* One descriptor drives forward convolution (pipeline:63 Conv2d in ConvBNReLU, :71-78  */
/* ASPP convs incl. dilation, :88-90 gate 1x1, :101 ConvTranspose2d as a 1x1 GEMM with a */
/* pixel-shuffle store) and data-gradient convolution (the ATen convolution_backward    */
/* input-gradient of the same call sites).                                              */
typedef struct aau_conv_desc {
    int32_t N, H, W;        /* gather-source spatial dims                               */
    int32_t Cin;            /* channels read per tap (multiple of 8)                    */
    int32_t src_pitch;      /* elements between source pixels                           */
    int32_t Ho, Wo;         /* output grid; GEMM M = N*Ho*Wo                            */
    int32_t Cout;           /* GEMM N (multiple of 8); for shuffle2x2 this is 4*Co      */
    int32_t dst_pitch;      /* elements between destination pixels                      */
    int32_t KH, KW;         /* taps                                                     */
    int32_t stride, pad, dil;
    int32_t Cpad;           /* per-tap channel count of the packed weights (mult. of 32) */
    int32_t shuffle2x2;     /* 1: Cout = 4*Co ordered [dy][dx][co]; pixel (y,x) of the  */
                            /*    GEMM writes destination pixel (2y+dy, 2x+dx), co      */
    int32_t accumulate;     /* 1: dst += result (read-modify-write, bf16)               */
    int32_t relu;           /* 1: clamp at 0 after the affine epilogue                  */
    /* Two-plane ("planar concat") operands: channels [split_c, C) of a pixel live in a second */
    /* dense plane, element offset split_off from the base, same pixel pitch.  0 = one plane.   */
    /* torch.cat([skip, up], 1) of the decoder (pipeline:108) at level 1 is kept as two dense   */
    /* [M][Co] planes: 96-byte half rows at a 192-byte pitch cost 1.5-2x per byte on gfx950      */
    /* (round-2 micro-benchmark).  Only the kernels aau_conv_split_ok() names take them.          */
    int32_t src_split_c, src_split_off;
    int32_t dst_split_c, dst_split_off;
} aau_conv_desc;

/* dst[m][q] = epi( sum_{t,c} src[gather(m,t)][c] * wpk[q][t][c] )                      */
/* wpk: bf16 [Cout][KH*KW][Cpad]; bias/scale/shift: optional fp32 [Cout] (NULL = none;    */
/* [Co] and indexed by co when shuffle2x2);                                             */
/* epi(v) = relu?( (v + bias) * scale + shift ).  stats (optional):                     */
/* aau_stat buffer for Cout channels (see the typedef): accumulates sum and sum of        */
/* squares of v (pre-epilogue accumulator) per output channel, order-independently -- the */
/* batch statistics of the following BatchNorm2d in training mode (read them with         */
/* aau_bn_finalize / aau_fold_stats).  The caller zeroes stats beforehand.                */
int aau_conv_igemm(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk,
                   aau_bf16* dst, const float* bias, const float* scale, const float* shift,
                   aau_stat* stats, int64_t stats_bytes, void* stream);

/* A 3x3 data-gradient conv whose destination is the gradient dy of a [BatchNorm -> ReLU] layer (raw conv output z,  */
/* folded scale / shift, saved mean / invstd of THAT layer): besides dst it accumulates that layer's BatchNorm-backward */
/* sums  sum(g), sum(g * zhat),  g = bf16(dy) * [z*scale+shift > 0],  into `sums` (an aau_stat buffer for Cout channels,   */
/* zeroed by the caller) -- the separate reduce pass of aau_bn_bwd_reduce over (z, dy) disappears (pipeline:59-65 backward).*/
/* aau_stats_to_red turns the sums into the fp32 [2][C] `red` operand of the apply passes.  Served for the descriptors   */
/* aau_conv_bnred_ok() accepts (48 -> 48 channels on the strip kernel).                                                  */
int aau_conv_bnred_ok(const aau_conv_desc* d);
int aau_conv_igemm_bnred(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst,
                         const aau_bf16* z, int z_pitch, const float* scale, const float* shift,
                         const float* save_mean, const float* save_invstd, aau_stat* sums, int64_t sums_bytes,
                         void* stream);
int aau_stats_to_red(const aau_stat* stats, int64_t stats_bytes, int C, float* red, void* stream);

/* ConvBNReLU -> ConvBNReLU without the activation in between (pipeline:59-65 twice, e.g. d1 = Sequential(ConvBNReLU,  */
/* ConvBNReLU), pipeline:113): src is the RAW conv output z of the producing layer and in_scale / in_shift its folded   */
/* BatchNorm affine (aau_bn_finalize); the kernel applies y = relu(z * in_scale + in_shift) in LDS behind the landing     */
/* fill, so dst (+ stats) equals aau_bn_act followed by aau_conv_igemm BIT FOR BIT while y is never written or read.      */
/* Served for the descriptors aau_conv_bnin_ok() accepts: 3x3 with 48 / 96 channels in and out (the strip kernel), and      */
/* the 1x1 / ConvTranspose2d(2,2)-forward (shuffle2x2, pipeline:101 behind a ConvBNReLU pair, :116-120) problems of the      */
/* resident-weight kernel (H, W multiples of 16, at most 192 input channels); `bias` (1x1 form only, else NULL) as in        */
/* aau_conv_igemm.                                                                                                           */
int aau_conv_bnin_ok(const aau_conv_desc* d);
int aau_conv_igemm_bnin(const aau_conv_desc* d, const aau_bf16* src, const float* in_scale, const float* in_shift,
                        const aau_bf16* wpk, aau_bf16* dst, const float* bias, aau_stat* stats, int64_t stats_bytes,
                        void* stream);
/* ... and the same for the weight gradient of that convolution: x = relu(src * in_scale + in_shift) applied in LDS, */
/* dw += as aau_conv_wgrad (workspace: aau_conv_wgrad_ws_bytes of the same descriptor).  aau_conv_wgrad_bnin_ok():       */
/* descriptors that take the all-taps 3x3 kernel.                                                                       */
int aau_conv_wgrad_bnin_ok(const aau_conv_desc* d);
int aau_conv_wgrad_bnin(const aau_conv_desc* d, const aau_bf16* src, const float* in_scale, const float* in_shift,
                        const aau_bf16* dz, float* dw, float* ws, int64_t ws_bytes, void* stream);
/* The weight gradient of a ConvTranspose2d(2,2) (pipeline:101) is stated as aau_conv_wgrad of the 2x2 / stride-2 problem    */
/* with the FINE gradient as `src` and the coarse input activation as `dz`; when that activation was never stored, `dz` is   */
/* the producing layer's raw conv output and dz' = relu(dz * dz_scale + dz_shift) is applied on the operand in LDS.  Served   */
/* for the descriptors aau_conv_wgrad_bnin_dz_ok() accepts (1x1, or 2x2 / stride 2 with Wo a multiple of 128; M a multiple   */
/* of 128).                                                                                                                  */
int aau_conv_wgrad_bnin_dz_ok(const aau_conv_desc* d);
int aau_conv_wgrad_bnin_dz(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz, const float* dz_scale,
                           const float* dz_shift, float* dw, float* ws, int64_t ws_bytes, void* stream);

/* 1 when the launch this descriptor selects supports its two-plane operands (src_split_c /  */
/* dst_split_c): mode 0 = aau_conv_igemm (the resident-weight 3x3 kernels), 1 = aau_conv_wgrad */
/* (wgrad3x3).  A descriptor with split operands that the selected kernel cannot serve fails   */
/* loudly in the call itself.                                                                  */
int aau_conv_split_ok(const aau_conv_desc* d, int mode);

/* 1 when the halo-tiled 3x3 kernels (csrc/conv3x3.hip) serve this descriptor: 3x3, pad 1,   */
/* stride 1, H and W multiples of 16, no accumulate.                                         */
int aau_conv_is_halo3x3(const aau_conv_desc* d);

/* Weight-gradient of the same convolutions (ATen convolution_backward, weight part):   */
/* dw[q][t][c] += sum_m dz[m][q] * src[gather(m,t)][c]   (the caller zeroes dw)         */
/* d->Cout = channels of dz (pitch d->dst_pitch), d->Cin = channels of src.             */
/* dw is [Cout][KH*KW][Cin] fp32 dense.  ConvTranspose2d(2,2) uses the same entry with    */
/* the roles swapped by the caller: dz := the layer input g (q = its Cin), src := the      */
/* output gradient gathered with stride 2 / 2x2 taps (c = Cout), giving [Cin][4][Cout].    */
/* The pixel reduction is split over workgroups.  With a workspace `ws` (16-byte aligned, */
/* at least aau_conv_wgrad_ws_bytes(d) bytes, contents don't care) every workgroup stores */
/* its partial tile and a second launch sums them in a fixed order: bitwise reproducible  */
/* and faster (no atomic tail).  ws == NULL: partial tiles are added with fp32 atomics.   */
int aau_conv_wgrad(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* dz,
                   float* dw, float* ws, int64_t ws_bytes, void* stream);
int aau_conv_wgrad_ws_bytes(const aau_conv_desc* d, int64_t* bytes);

/* Grouped weight gradient: n (<= 8) problems tiled into ONE grid of 192 x 192 (q x c) tiles, every     */
/* tile over the full pixel range (no split-K workspace, bitwise reproducible), dw_i += result.        */
/* Built for the ASPP bridge (pipeline:67-83: blocks[0..3] and project share their input / gradient     */
/* tensors): three dilated 3x3 + 1x1 + the 5C -> C projection in one launch.  Each problem must be a    */
/* 1x1/stride-1 conv or have Wo % 32 == 0 (K-steps are 32-pixel row segments; steps whose source row is  */
/* outside the image are skipped); aau_conv_wgrad_group_ok says whether the group is in range AND large   */
/* enough to fill the chip (otherwise use aau_conv_wgrad per problem).                                  */
int aau_conv_wgrad_group_member_ok(const aau_conv_desc* d);   /* one problem's eligibility (the group needs >= 160 tiles in all) */
int aau_conv_wgrad_group_ok(const aau_conv_desc* descs, int n);
/* The launch is persistent: 512 workgroups pull tiles, longest first, from eight work queues (one per XCD); `queue`   */
/* holds their heads: aau_conv_wgrad_group_queue_bytes() bytes that the CALLER has zeroed on the stream before the   */
/* call (every call: the kernel leaves them advanced).                                                               */
int64_t aau_conv_wgrad_group_queue_bytes(void);
int aau_conv_wgrad_group(const aau_conv_desc* descs, const aau_bf16* const* srcs,
                         const aau_bf16* const* dzs, float* const* dws, int n, int32_t* queue, int64_t queue_bytes,
                         void* stream);

/* Several independent convolutions in ONE launch: problem i is exactly aau_conv_igemm(&descs[i], srcs[i], wpks[i],      */
/* dsts[i], no bias / affine, stats[i]) -- its own weights, dilation, destination and statistics.  Built for the forward   */
/* of the four spatial ASPP branches (pipeline:80-83), which are one workgroup per CU each when launched separately.        */
/* n = 2..4 problems that would each take the 128 x 192 tile (aau_conv_igemm_multi_ok); bitwise the results of the         */
/* separate launches.  `stats` / `stats_bytes` may be NULL (no statistics) or hold NULL entries.                           */
int aau_conv_igemm_multi_ok(const aau_conv_desc* descs, int n);
int aau_conv_igemm_multi(const aau_conv_desc* descs, const aau_bf16* const* srcs, const aau_bf16* const* wpks,
                         aau_bf16* const* dsts, aau_stat* const* stats, const int64_t* stats_bytes, int n, void* stream);

/* Grouped data gradient: dst = (descs[0].accumulate ? dst : 0) + sum_i conv_i(srcs[i], wpks[i]) -- n (2..8) stride-1      */
/* same-size convolutions (1x1 or dilated 3x3, pad = dil * (k/2)) of DIFFERENT sources with the same N, H, W, Cin, Cout   */
/* into ONE destination.  Built for the input gradient of the ASPP bridge (pipeline:80-83: the 1x1 and the three dilated  */
/* branches all read x, so dL/dx is the sum of four data gradients with only Cout = 8c output channels at 1/16          */
/* resolution).  The K dimension is the concatenation of every segment's (tap, channel) list, accumulated in registers     */
/* and cut into contiguous ranges so that tiles x ranges fills the chip; the ranges leave fp32 slabs in `ws`              */
/* (aau_conv_igemm_group_ws_bytes, 16-byte aligned, may be null when that returns 0) added in a fixed order:            */
/* deterministic.  Cin % 64 == 0, Cpad == Cin, Cout % 192 == 0, dst_pitch % 8 == 0; descs[i > 0].accumulate must be 1.   */
/* aau_conv_igemm_group_ok: in range AND too few 128 x 192 tiles for separate launches to fill the chip.                */
int aau_conv_igemm_group_ok(const aau_conv_desc* descs, int n);
int64_t aau_conv_igemm_group_ws_bytes(const aau_conv_desc* descs, int n);
int aau_conv_igemm_group(const aau_conv_desc* descs, const aau_bf16* const* srcs, const aau_bf16* const* wpks, int n,
                         aau_bf16* dst, float* ws, void* stream);

/* Traversal hint (per calling thread).  alternate = 1: from now on every launch of the large  */
/* streaming kernels (BN / pool / first and last layer / 3x3 and 1x1 convs / weight grads)    */
/* walks its tensors in the direction OPPOSITE to the previous such launch, starting with     */
/* "forward"; alternate = 0 (default): always forward.  Results do not depend on it.  Why:    */
/* a consumer that starts where its producer just finished finds that end of a > 128 MB       */
/* tensor still in the 256-MiB Infinity Cache (BN backward reduce -> apply at 201 MB tensors: */
/* -12 % for the pair).  The engine calls it at the start of every forward / backward replay  */
/* so that the directions are the same in every step.                                         */
int aau_traverse(int alternate);

/* ---- first layer: Conv2d(1, C, 3, pad 1) on fp32 input (pipeline:113 d1[0]) --------- */
int aau_conv1_fwd(const float* x, const float* w /*[C][9]*/, aau_bf16* z, aau_stat* stats, int64_t stats_bytes,
                  int N, int H, int W, int C, void* stream);
int aau_conv1_wgrad(const float* x, const aau_bf16* dz, float* dw /*[C][9]*/,
                    int N, int H, int W, int C, void* stream);

/* ---- weight packing (fp32 master -> bf16 GEMM operands), table driven ---------------- */
typedef struct aau_pack_entry {
    int64_t src_off;        /* element offset into the flat fp32 parameter buffer        */
    int64_t dst_off;        /* element offset into the packed bf16 buffer (multiple of 8) */
    int32_t R;              /* rows of the packed matrix                                 */
    int32_t T;              /* taps                                                      */
    int32_t C;              /* channels per tap                                          */
    int32_t Cpad;           /* padded channels per tap in the packed matrix              */
    int32_t s_r, s_t, s_c;  /* source strides (elements) of row / tap / channel          */
    int32_t t_flip;         /* 1: packed tap t reads source tap T-1-t                    */
    int32_t R2;             /* 0, or: row r = r1*R2 + r2 with strides (s_r, s_r2)        */
    int32_t s_r2;
    int64_t blk_begin;      /* first thread block of this entry: prefix sum of           */
                            /* ceil(R*T*Cpad/8 / 256) (a thread packs 8 channels)        */
} aau_pack_entry;
int aau_pack_weights(const float* flat, aau_bf16* packed, const aau_pack_entry* table_dev,
                     int n_entries, int64_t total_blocks, void* stream);

/* Clears up to 8 device buffers (16-byte aligned, sizes multiples of 16) in one launch and, when `counter` is not */
/* NULL, adds counter_inc (mod 2^64) to the 64-bit word it points to: the per-pass housekeeping of a training step */
/* (statistic / reduction workspaces, the flat gradient, the dropout seed chain of the reference's nn.Dropout,     */
/* att_aspp_unet_pipeline.py:78) that would otherwise be one fill per buffer.                                       */
int aau_zero_multi(void* const* bufs, const int64_t* bytes, int n, uint64_t* counter, uint64_t counter_inc, void* stream);

/* ---- BatchNorm2d (pipeline:64 and every BN of :71-90) -------------------------------- */
/* training: stats replicas -> mean / biased var -> scale = g*invstd, shift = b-mean*scale; */
/* saves mean, invstd; running stats: momentum 0.1, unbiased var; nbt += 1.              */
int aau_bn_finalize(const aau_stat* stats, int64_t stats_bytes, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* scale, float* shift, float* save_mean, float* save_invstd,
                    int C, int64_t count, float eps, float momentum, void* stream);
/* eval: scale/shift from running statistics                                             */
int aau_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float* scale, float* shift, int C, float eps,
                     void* stream);

/* Several BatchNorm layers of the SAME width in one launch each (blockIdx.y = layer; n <= 8): the four spatial branches  */
/* of the ASPP (pipeline:80-83) run their statistics / activation / backward passes as 2 + 3 launches instead of 8 + 12.   */
/* `tab` is a HOST array of n rows of pointers, in the order of the single-layer call's pointer arguments:                 */
/*   finalize_multi   [stats, gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, invstd]     */
/*   act_multi        [z, y, scale, shift]                         (ReLU optional, no dropout / broadcast)                 */
/*   bwd_reduce_multi [z, dy, scale, shift, mean, invstd, red, ws] (ws: 2 C x 1024 floats of scratch per layer)            */
/*   bwd_apply_multi  [z, dz, gamma, mean, invstd, red, dgamma, dbeta, dy, scale, shift]                                   */
/* Every layer's arithmetic and order of additions are those of aau_bn_finalize / aau_bn_act / aau_bn_bwd_reduce /          */
/* aau_bn_bwd_apply with the same grid: bitwise the same results as n single launches whose reduce grid is capped alike.  */
int aau_bn_finalize_multi(int n, const void* const* tab, int64_t stats_bytes, int C, int64_t count, float eps,
                          float momentum, void* stream);
int aau_bn_act_multi(int n, const void* const* tab, int z_pitch, int y_pitch, int64_t M, int C, int relu, void* stream);
int aau_bn_bwd_reduce_multi(int n, const void* const* tab, int z_pitch, int dy_pitch, int N, int H, int W, int C, int relu,
                            void* stream);
int aau_bn_bwd_apply_multi(int n, const void* const* tab, int z_pitch, int dz_pitch, int dy_pitch, int64_t M, int C,
                           int relu, void* stream);
/* Dropout masks are counter based: keep(m, c) = hash(*drop_seed, m*C + c) >= p.  drop_seed is */
/* a DEVICE pointer (read only when drop_p > 0, may be NULL otherwise): the caller advances it  */
/* on the stream once per step, so a step captured as a hipGraph draws a new mask per replay.  */
/* y = relu?(z*scale+shift) [* dropout keep mask / (1-p)], optional broadcast of one     */
/* source row per image (ASPP image-pool branch, pipeline:82).                           */
int aau_bn_act(const aau_bf16* z, int z_pitch, aau_bf16* y, int y_pitch, const float* scale,
               const float* shift, int64_t M, int C, int relu, int64_t bcast_hw,
               float drop_p, const uint64_t* drop_seed, void* stream);
/* y = relu(z*scale+shift) and p = MaxPool2d(2)(y) in one pass (encoder stages, :115-118)   */
int aau_bn_act_pool(const aau_bf16* z, int z_pitch, aau_bf16* y, int y_pitch, aau_bf16* p, int p_pitch,
                    const float* scale, const float* shift, int N, int H, int W, int C, void* stream);
/* MaxPool2d(2) (pipeline:115-118)                                                        */
int aau_maxpool2(const aau_bf16* y, int y_pitch, aau_bf16* p, int p_pitch, int N, int H, int W,
                 int C, void* stream);
/* backward of [BN -> ReLU (-> dropout)] with up to two gradient sources:                 */
/*   dy (same resolution, optional) and dpool (gradient of MaxPool2d(2) output, optional, */
/*   routed to the first maximum of each window).  Pass 1 writes the masked gradient      */
/*   g = relu'(y) * (dy + pool-routed) into dz and accumulates sum(g), sum(g*zhat) into    */
/*   red [2][C] (overwritten: totals over all pixels); pass 2 turns it into dz in place    */
/*   and adds dgamma / dbeta.                                                            */
int aau_bn_bwd_reduce(const aau_bf16* z, int z_pitch, const aau_bf16* dy, int dy_pitch,
                      const aau_bf16* dpool, int dpool_pitch, aau_bf16* dz, int dz_pitch,
                      const float* scale, const float* shift, const float* save_mean,
                      const float* save_invstd, float* red, int N, int H, int W, int C,
                      int relu, float drop_p, const uint64_t* drop_seed, float* ws, void* stream);
/* Workspace of the reduce passes (aau_bn_bwd_reduce, aau_conv1_bn_bwd_reduce,                */
/* aau_bn_bwd_reduce_outconv) for a BatchNorm of up to C channels: every workgroup stores    */
/* its row of partial sums there and a second launch adds the rows in an order that depends  */
/* on the grid only (no float atomics: bitwise reproducible).  16-byte aligned, contents     */
/* irrelevant on entry; one buffer serves any number of calls issued one after another on a  */
/* stream (never two at the same time).                                                      */
int64_t aau_bn_red_ws_bytes(int C);
/* Non-pooled layers may skip the intermediate: pass dz = NULL to the reduce pass and give */
/* the apply pass dy (+ scale, shift, relu, dropout parameters); it recomputes the mask.   */
int aau_bn_bwd_apply(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch,
                     const float* gamma, const float* save_mean, const float* save_invstd,
                     const float* red, float* dgamma, float* dbeta, int64_t M, int C,
                     const aau_bf16* dy, int dy_pitch, const float* scale, const float* shift,
                     int relu, float drop_p, const uint64_t* drop_seed, void* stream);
/* Pooled layers (the encoder's second ConvBNReLU, pipeline:113-116 d1..d4 + MaxPool2d): the  */
/* apply pass redoes the max-pool routing from dy (skip path, may be NULL), z and dpool, so  */
/* the reduce pass may be called with dz = NULL there as well (no routed gradient stored).   */
int aau_bn_bwd_apply_pool(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch,
                          const float* gamma, const float* save_mean, const float* save_invstd,
                          const float* red, float* dgamma, float* dbeta, int N, int H, int W, int C,
                          const aau_bf16* dy, int dy_pitch, const aau_bf16* dpool, int dpool_pitch,
                          const float* scale, const float* shift, int relu, void* stream);

/* First layer (pipeline:113 d1[0]): the apply pass fused with the weight gradient of its   */
/* Conv2d(1, C, 3, pad 1) -- the layer has no input gradient, so dz is never written.     */
/* x: the fp32 frame [N][H][W]; dw: [C][9] fp32 (+=); ws: aau_bn_red_ws_bytes(C) (rows of   */
/* partial sums, added in a fixed order).                                                   */
/* z == NULL: z is recomputed from x and the conv weights w [C][9] (see aau_conv1_bn_act). */
int aau_bn_bwd_apply_conv1(const aau_bf16* z, int z_pitch, const float* gamma, const float* save_mean,
                           const float* save_invstd, const float* red, float* dgamma, float* dbeta,
                           int N, int H, int W, int C, const aau_bf16* dy, int dy_pitch,
                           const float* scale, const float* shift, const float* x, const float* w,
                           float* dw, float* ws, void* stream);
/* The first layer's raw output z costs 9 FMAs per value to recompute and 2 bytes to move,   */
/* so it is never stored in training: aau_conv1_fwd(z = NULL) accumulates the statistics     */
/* only, and these two recompute z (same fma chain, same bf16 rounding) from the frame:      */
/* y = relu(bf16(conv1(x))*scale+shift), and the BatchNorm-backward reduce of that layer.     */
int aau_conv1_bn_act(const float* x, const float* w, aau_bf16* y, int y_pitch, const float* scale,
                     const float* shift, int N, int H, int W, int C, void* stream);
int aau_conv1_bn_bwd_reduce(const float* x, const float* w, const aau_bf16* dy, int dy_pitch,
                            const float* scale, const float* shift, const float* save_mean,
                            const float* save_invstd, float* red, int N, int H, int W, int C,
                            float* ws, void* stream);

/* ---- ASPP image-pool branch (pipeline:75-77,82) -------------------------------------- */
/* ws: caller-provided fp32 [AAU_GAP_WS_ROWS][N][C] workspace: one row of sums per pixel   */
/* slab, added in slab order (no float atomics)                                          */
#define AAU_GAP_WS_ROWS 64
int aau_gap_fwd(const aau_bf16* x, int x_pitch, aau_bf16* pooled, float* ws, int N, int HW, int C, void* stream);
/* dsrc[n][p][c] += dpooled[n][c] / HW  (accumulate into bf16)                           */
int aau_gap_bwd(const aau_bf16* dpooled, aau_bf16* dx, int dx_pitch, int N, int HW, int C, void* stream);
/* out[n][c] = sum_p src[n][p][c]   (bf16 in, bf16 out, fp32 accumulate)                  */
int aau_spatial_sum(const aau_bf16* src, int src_pitch, aau_bf16* out, float* ws, int N, int HW, int C, void* stream);

/* The image-pool branch between its spatial ends (pipeline:75-77): pooled [B][Cin] -> Conv2d(Cin, Cout, 1, bias=False) */
/* -> BatchNorm2d over the B samples (training) -> ReLU, as three latency-sized launches (csrc/poolbranch.hip) instead of   */
/* the generic conv / finalize / reduce / apply / wgrad / dgrad sequence.  B <= 16, Cin and Cout multiples of 8.          */
/* fwd: z[b][q] = bf16(sum_c x[b][c] wpk[q][c]); batch statistics of the fp32 sums with aau_bn_finalize's arithmetic.     */
/* bwd: g = dy * [z*scale+shift > 0]; dbeta += sum g; dgamma += sum g zhat; dz = bf16(gamma invstd (g - mean g -       */
/*      zhat mean(g zhat))); dw[q][c] += sum_b dz[b][q] x[b][c].   dx: dx[b][c] = bf16(sum_q dz[b][q] wpd[c][q]).        */
int aau_poolbranch_fwd(const aau_bf16* x, int x_pitch, const aau_bf16* wpk, int Cpad, aau_bf16* z, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                       float* scale, float* shift, float* save_mean, float* save_invstd, int B, int Cin, int Cout,
                       float eps, float momentum, void* stream);
int aau_poolbranch_bwd(const aau_bf16* dy, int dy_pitch, const aau_bf16* z, const aau_bf16* x, int x_pitch,
                       const float* gamma, const float* scale, const float* shift, const float* save_mean,
                       const float* save_invstd, aau_bf16* dz, float* dgamma, float* dbeta, float* dw, int B, int Cin,
                       int Cout, void* stream);
int aau_poolbranch_dx(const aau_bf16* dz, const aau_bf16* wpd, int Cpad_d, aau_bf16* dx, int dx_pitch, int B, int Cin,
                      int Cout, void* stream);

/* ---- attention gate (pipeline:85-92) -------------------------------------------------- */
/* psi_pre[m] = sum_f wpsi[f]*relu(zg[m,f]*sg[f]+hg[f] + zx[m,f]*sx[f]+hx[f]); also sum /   */
/* sumsq of psi_pre into stats [REPLICAS][2][1].                                           */
int aau_gate_psi(const aau_bf16* zg, const aau_bf16* zx, const float* sg, const float* hg,
                 const float* sx, const float* hx, const float* wpsi, float* psi_pre,
                 aau_stat* stats, int64_t stats_bytes, int64_t M, int F, void* stream);
/* alpha[m] = sigmoid(psi_pre*scale1+shift1); out[m,c] = x[m,c]*alpha[m]                   */
int aau_gate_apply(const aau_bf16* x, int x_pitch, const float* psi_pre, const float* scale1,
                   const float* shift1, float* alpha, aau_bf16* out, int out_pitch, int64_t M,
                   int C, void* stream);
/* The per-channel sums of the three backward steps are added in a fixed order (workgroup  */
/* rows in `ws` = aau_bn_red_ws_bytes(C) + one fold launch): no float atomics.              */
/* backward step 1: dx[m,c] = dout*alpha; dq[m] = (sum_c dout*x) * alpha*(1-alpha);          */
/* red1 fp32 [4] = (sum dq, sum dq*psihat, 0, 0), overwritten                                */
int aau_gate_bwd1(const aau_bf16* dout, int dout_pitch, const aau_bf16* x, int x_pitch,
                  const float* alpha, const float* psi_pre, const float* mean1,
                  const float* invstd1, aau_bf16* dx, int dx_pitch, float* dq, float* red1,
                  int64_t M, int C, float* ws, void* stream);
/* backward step 2: dpsi_pre = BN(1) backward of dq; ds[m,f] = dpsi_pre*wpsi[f]*[s>0];      */
/* writes ds (bf16, masked gradient shared by both branches) and tot fp32 [4][F] =          */
/* (sum dpsi_pre*s = the psi weight gradient, sum ds, sum ds*zhat_g, sum ds*zhat_x),        */
/* overwritten; dgamma1/dbeta1 +=.                                                          */
int aau_gate_bwd2(const float* dq, const float* psi_pre, const float* red1, const float* gamma1,
                  const float* mean1, const float* invstd1, const aau_bf16* zg,
                  const aau_bf16* zx, const float* sg, const float* hg, const float* sx,
                  const float* hx, const float* mean_g, const float* invstd_g,
                  const float* mean_x, const float* invstd_x, const float* wpsi,
                  aau_bf16* ds, float* tot, float* dgamma1, float* dbeta1, int64_t M, int F,
                  float* ws, void* stream);
/* backward step 3: dzg/dzx from ds (BN backward without ReLU for both branches);           */
/* dwpsi[f] += tot[0][f] (dwpsi may be NULL); dgamma / dbeta of both branch BNs +=          */
int aau_gate_bwd3(const aau_bf16* ds, const aau_bf16* zg, const aau_bf16* zx,
                  const float* gamma_g, const float* mean_g, const float* invstd_g,
                  const float* gamma_x, const float* mean_x, const float* invstd_x,
                  const float* tot, aau_bf16* dzg, aau_bf16* dzx,
                  float* dgamma_g, float* dbeta_g, float* dgamma_x, float* dbeta_x,
                  float* dwpsi, int64_t M, int F, void* stream);

/* ---- out_conv: Conv2d(C, 1, 1) with bias (pipeline:122) -------------------------------- */
int aau_outconv_fwd(const aau_bf16* y, int y_pitch, const float* w, const float* b,
                    float* logits, int64_t M, int C, void* stream);
/* ws: fp32 [AAU_STAT_REPLICAS][C+8] workspace (zeroed by the call)                          */
int aau_outconv_bwd(const aau_bf16* y, int y_pitch, const float* dlogits, const float* w,
                    aau_bf16* dy, int dy_pitch, float* dw, float* db, float* ws, int64_t M, int C,
                    void* stream);
/* Network head in training (pipeline:121-122,126: the last ConvBNReLU feeds out_conv only).    */
/* forward: logits[m] = b + sum_c w[c] * bf16(relu(z[m][c]*scale[c]+shift[c])) -- the activated   */
/* tensor is never written (the logits of aau_bn_act + aau_outconv_fwd up to fp32 summation order). */
int aau_bn_act_outconv(const aau_bf16* z, int z_pitch, const float* scale, const float* shift,
                       const float* w, const float* b, float* logits, int64_t M, int C, void* stream);
/* backward: the gradient w.r.t. that activation is rank one, dy[m][c] = bf16(dlogits[m]*w[c]),   */
/* and is never written either.  reduce_outconv = aau_outconv_bwd's parameter gradients           */
/* (dw[c] += sum dl*y, db += sum dl; y recomputed from z) + aau_bn_bwd_reduce of the last         */
/* BatchNorm; ws: aau_bn_red_ws_bytes(C).  apply_rank1 = aau_bn_bwd_apply with that dy.            */
int aau_bn_bwd_reduce_outconv(const aau_bf16* z, int z_pitch, const float* dlogits, const float* w,
                              const float* scale, const float* shift, const float* save_mean,
                              const float* save_invstd, float* red, float* dw, float* db, float* ws,
                              int64_t M, int C, void* stream);
int aau_bn_bwd_apply_rank1(const aau_bf16* z, int z_pitch, aau_bf16* dz, int dz_pitch,
                           const float* gamma, const float* save_mean, const float* save_invstd,
                           const float* red, float* dgamma, float* dbeta, int64_t M, int C,
                           const float* dlogits, const float* w_out, const float* scale,
                           const float* shift, void* stream);
/* out[c] += per-channel sum over pixels of a bf16 tensor (ConvTranspose2d bias gradient);   */
/* ws as above                                                                              */
int aau_colsum(const aau_bf16* src, int src_pitch, float* out, float* ws, int64_t M, int C, void* stream);
/* out[i] += sum_r ws[r*stride + i] (i < n, r < AAU_STAT_REPLICAS): reads a per-channel sum out  */
/* of the `stats` a conv epilogue accumulated -- the ConvTranspose2d bias gradient is the        */
/* channel sum of the gradient that the preceding data-gradient convs produced.                 */
int aau_fold_replicas(const float* ws, int stride, float* out, int n, void* stream);
/* out[i] += total of statistic `which` (0 sum, 1 sum of squares) of channel c_begin + i of an aau_stat buffer   */
/* for C channels (e.g. the ConvTranspose2d bias gradient = channel sums that a data-gradient conv accumulated) */
int aau_fold_stats(const aau_stat* stats, int64_t stats_bytes, int C, int which, int c_begin, int n, float* out, void* stream);
/* out[i] += sum-statistic of channel ca_begin + i of buffer a (+ of channel cb_begin + i of buffer b, b may be NULL): the   */
/* ConvTranspose2d bias gradient of a gated decoder level has two contributions (the conv and the gate data gradients)     */
int aau_fold_stats_pair(const aau_stat* a, int64_t a_bytes, int CA, int ca_begin, const aau_stat* b, int64_t b_bytes,
                        int CB, int cb_begin, int n, float* out, void* stream);
/* out fp64 [2][C] = the totals of an aau_stat buffer (NaN if poisoned by a non-finite partial)                   */
int aau_stats_to_f64(const aau_stat* stats, int64_t stats_bytes, int C, double* out, void* stream);

/* Residual gate of the ablation variant (test_ablation.py:128-143; no BatchNorm, bias on psi):     */
/*   alpha[m] = sigmoid(sum_f wpsi[f]*relu(zg[m,f]+zx[m,f]) + bpsi);  out[m,c] = x[m,c]*alpha[m] + x[m,c] */
int aau_gate2_fwd(const aau_bf16* zg, const aau_bf16* zx, const float* wpsi, const float* bpsi,
                  const aau_bf16* x, int x_pitch, float* alpha, aau_bf16* out, int out_pitch, int64_t M,
                  int F, int C, void* stream);
/* backward: dx = dout*(1+alpha) (written); ds[m,f] = dpre*wpsi[f]*[s>0] with dpre = <dout,x>*alpha*(1-alpha)  */
/* (the gradient of BOTH 1x1 outputs); dwpsi += sum dpre*relu(s), dbpsi += sum dpre; rep_ws:                     */
/* aau_bn_red_ws_bytes(F) (one row of partial sums per wave, added in a fixed order)                           */
int aau_gate2_bwd(const aau_bf16* dout, int dout_pitch, const aau_bf16* x, int x_pitch, const float* alpha,
                  const aau_bf16* zg, const aau_bf16* zx, const float* wpsi, aau_bf16* dx, int dx_pitch,
                  aau_bf16* ds, float* rep_ws, float* dwpsi, float* dbpsi, int64_t M, int F, int C, void* stream);

/* ---- criterion (pipeline:219-232 build_criterion with ComboLoss :187-189, DiceLoss        */
/* :173-178, EdgeLoss :196-216) and metrics (:191-194 iou_score, :240 eval Dice) -------- */
/* sums: fp32 [AAU_STAT_REPLICAS][B][8] workspace (zeroed by the call; [0] holds the per-   */
/* sample sums afterwards; behind it the tile sums are accumulated as 64-bit fixed point:   */
/* order-independent); loss_out: fp32 [4] = total, dice,                                    */
/* bce, edge.  dlogits (optional, fp32 [B*H*W]) receives d(loss*loss_scale)/d(logits).       */
int aau_criterion(const float* logits, const float* targets, float* sums, float* loss_out,
                  float* dlogits, int B, int H, int W, int finetune, float neg_bce_w,
                  float edge_w, float loss_scale, void* stream);
/* The loss classes on their own (pipeline:173-189 DiceLoss / TverskyLoss / ComboLoss, :196-216  */
/* EdgeLoss; every sample counts):                                                               */
/*   total = w_ratio*mean_b(1 - (nu*tp_b+s_n)/(d_tp*tp_b + d_p*sum p_b + d_t*sum t_b + s_d))      */
/*         + w_bce*mean(bce) + w_edge*mean|grad p - grad t|                                       */
/* coef9 (host) = {w_ratio, nu, s_n, d_tp, d_p, d_t, s_d, w_bce, w_edge}; loss_out fp32 [4] =     */
/* total, ratio term, bce term, edge term; dlogits optional.                                     */
int aau_loss_terms(const float* logits, const float* targets, float* sums, float* loss_out,
                   float* dlogits, int B, int H, int W, const float* coef9, void* stream);
/* metrics_out: fp32 [2] = mean soft Dice (1 - DiceLoss), mean hard IoU at thr            */
int aau_seg_metrics(const float* logits, const float* targets, float* sums, float* metrics_out,
                    int B, int H, int W, float thr, void* stream);

/* Integer counts behind eval_segmentation_batch.py:41-49 (`_bin`: a > 0; `dice`, `iou`):     */
/* out3 (device, uint64 [3], zeroed by the call) = |a|, |b|, |a & b| over n elements; a / b are  */
/* uint8 masks or fp32 arrays (x_is_f32).  Exact: the quotients are formed on the host in fp64. */
int aau_seg_counts(const void* a, int a_is_f32, const void* b, int b_is_f32, int64_t n,
                   uint64_t* out3, void* stream);

/* ---- optimiser (pipeline:302 AdamW, :323 clip_grad_norm_) ------------------------------ */
/* norm_ws: fp32 [AAU_SQNORM_WS]; [0] = sum of squares of grad*inv_scale (workgroup partials  */
/* in the rest, added in a fixed order: bitwise reproducible, no float atomics)              */
#define AAU_SQNORM_WS 1028
int aau_grad_sqnorm(const float* grad, int64_t n, float inv_scale, float* norm_ws, void* stream);
/* p,m,v,g: flat fp32 [n].  clip coefficient = min(1, max_norm/(sqrt(norm_ws)+1e-6));       */
/* step_dev: int64 step counter on the device, incremented by the call (bias correction).  */
/* If sqrt(norm_ws) is inf/nan the update is skipped (GradScaler semantics, :324).          */
int aau_adamw_step(float* p, float* m, float* v, const float* g, int64_t n,
                   const float* norm_ws, int64_t* step_dev, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float max_norm, float inv_scale, void* stream);

/* The same update with lr / weight_decay read from DEVICE memory at run time, so a captured    */
/* hipGraph follows the LR schedule (pipeline:303-306,325), and one (lr, weight_decay) pair per   */
/* parameter group: hyp fp32 [n_groups][2]; group_of_block: group id of every 64-element block   */
/* of the flat buffers (null when n_groups == 1).  Differential LR: test_ablation.py:576-586.    */
int aau_adamw_step_dev(float* p, float* m, float* v, const float* g, int64_t n, const float* norm_ws,
                       int64_t* step_dev, const float* hyp, const unsigned char* group_of_block,
                       int n_groups, float beta1, float beta2, float eps, float max_norm,
                       float inv_scale, void* stream);

/* ---- small utilities -------------------------------------------------------------------- */
int aau_f32_to_bf16(const float* src, aau_bf16* dst, int64_t n, void* stream);
int aau_bf16_to_f32(const aau_bf16* src, float* dst, int64_t n, void* stream);
/* NCHW fp32 -> NHWC bf16 with pitch, and back (module-boundary plumbing)                   */
int aau_nchw_to_nhwc(const float* src, aau_bf16* dst, int dst_pitch, int N, int C, int H, int W, void* stream);
int aau_nhwc_to_nchw(const aau_bf16* src, int src_pitch, float* dst, int N, int C, int H, int W, void* stream);
/* horizontal flip of [N][H][W] fp32 frames (TTA, pipeline:336-338) and the TTA merge        */
int aau_hflip_f32(const float* src, float* dst, int N, int H, int W, void* stream);
int aau_tta_merge(const float* l, const float* l_flipped, float* prob, int N, int H, int W, void* stream);

/* Sliding-window inference (build-side extension, BASELINE config 5; the reference's Att-ASPP   */
/* path has no tiling): Gaussian-weighted blend of ny*nx window logit maps            */
/* [ny*nx][win][win] (window (iy,ix) at offset (iy*stride, ix*stride)) into out [H][W].           */
int aau_window_blend(const float* win_logits, float* out, int H, int W, int win, int stride, int ny, int nx,
                     float sigma, void* stream);

/* ---- GPU-resident inference tail and input pipeline ------------ */
/* Batches of N frames, [N][H][W] row-major; uint8 masks hold 0 / 1.  Restated from the published algorithms   */
/* of OpenCV / scikit-image / SciPy that the reference calls (cv2 and skimage are not importable in the build  */
/* container, so those outputs are "parity unpinned"; the checker is oracle/imgproc_ref.py).                   */
/* cv2.resize(INTER_LINEAR): fp32 (pipeline:455 resize-back of the probability map) / uint8 (11-bit fixed point, */
/* albumentations Resize, pipeline:147,451)                                                                    */
int aau_resize_bilinear_f32(const float* src, int Hs, int Ws, float* dst, int Hd, int Wd, int N, void* stream);
int aau_resize_bilinear_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd, int N, void* stream);
/* cv2.GaussianBlur(x, (5,5), 0): [1 4 6 4 1]/16 separable, BORDER_REFLECT_101 (pipeline:456)                    */
int aau_gauss5_f32(const float* src, float* dst, int N, int H, int W, void* stream);
/* (prob > thr).astype(uint8) (pipeline:457)                                                                   */
int aau_threshold_u8(const float* src, float thr, uint8_t* dst, int64_t n, void* stream);
/* connected components (skimage.measure.label / scipy.ndimage.label): labels[i] = smallest linear index of the  */
/* pixel's component, -1 for background; conn8 = 8-connectivity                                                */
int aau_cc_label(const uint8_t* mask, int32_t* labels, int N, int H, int W, int conn8, void* stream);
/* refine_mask :341-345 / model_attention_aspp.py:76-80: out = the largest component if it has >= min_area        */
/* pixels (the earliest in raster order on ties), else zeros.  Workspaces: labels, sizes int32 [N*H*W], best u64 [N] */
int aau_cc_keep_largest(const uint8_t* mask, uint8_t* out, int32_t* labels_ws, int32_t* sizes_ws, uint64_t* best_ws,
                        int N, int H, int W, int conn8, int min_area, void* stream);
/* scipy.ndimage.binary_fill_holes (pipeline:348).  Workspaces int32 [N*H*W] each                                 */
int aau_fill_holes(const uint8_t* mask, uint8_t* out, int32_t* labels_ws, int32_t* flag_ws, int N, int H, int W,
                   void* stream);
/* binary dilation (erode = 0) / erosion (erode = 1), pixels outside the frame ignored: shape 7 = cv2 7x7 ellipse */
/* (MORPH_CLOSE of pipeline:347 = dilate then erode), shape 3 = full 3x3 (model_attention_aspp.py:75)             */
int aau_morph(const uint8_t* src, uint8_t* dst, int N, int H, int W, int shape, int erode, void* stream);
/* cv2.normalize(x, None, 0, 255, NORM_MINMAX) per frame (pipeline:449); mm_ws int32 [2N]                         */
int aau_normalize_minmax_u8(const uint8_t* src, uint8_t* dst, int32_t* mm_ws, int N, int H, int W, void* stream);
/* cv2.createCLAHE(clip_limit, (tiles, tiles)).apply (pipeline:450); lut_ws uint8 [N][tiles*tiles][256]           */
int aau_clahe_u8(const uint8_t* src, uint8_t* dst, uint8_t* lut_ws, int N, int H, int W, float clip_limit, int tiles,
                 void* stream);
/* cv2.medianBlur(x, 3) (pipeline:450)                                                                           */
int aau_median3_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, void* stream);
/* ToFloat(max_value): dst = float(src) / max_value (pipeline:154,451)                                           */
int aau_u8_to_f32(const uint8_t* src, float* dst, float max_value, int64_t n, void* stream);
/* model_attention_aspp.py:20-31 crop_roi_224: origin[f] = (x0, y0) of the R x R window centred on the mean position */
/* of the pixels brighter than 1.2 x the frame mean (frame centre if none), clamped to the frame; sums_ws fp64 [4N]  */
int aau_roi_origin(const float* img, double* sums_ws, int32_t* origin, int N, int H, int W, int R, void* stream);
int aau_roi_crop(const float* src, const int32_t* origin, float* dst, int N, int H, int W, int R, void* stream);
/* :54-59: full-frame probability = sigmoid(logits) pasted at the window, 0 elsewhere                              */
int aau_roi_paste_sigmoid(const float* logits, const int32_t* origin, float* full, int N, int H, int W, int R, void* stream);
/* :66-69: areas[f] = #(prob > thr)                                                                                */
int aau_frame_areas(const float* prob, float thr, int32_t* areas, int N, int H, int W, void* stream);

/* ---- random training augmentations of FetalACDataset (pipeline:149-153), batched over resident uint8 frames ------ */
/* Every entry takes PER-FRAME parameters; a frame whose transform was not drawn gets the identity (matrix / table /  */
/* alpha 0 / flag 0), so one launch serves a batch.  albumentations / cv2 published algorithms with exact bilinear    */
/* weights (parity unpinned: neither library is importable in the build container).                                 */
/* HorizontalFlip(0.5) (pipeline:149): frames whose flag is set are mirrored along x                                 */
int aau_hflip_frames_u8(const uint8_t* src, uint8_t* dst, const uint8_t* flags, int N, int H, int W, void* stream);
/* Affine(scale, rotate, translate_percent, p=0.7) (pipeline:150) = cv2.warpAffine: dst(x,y) = src(M (x,y,1)); inv_mats */
/* fp64 [N][6] = the dst->src maps; bilinear (image) or nearest (mask), constant `border` outside the frame          */
int aau_warp_affine_u8(const uint8_t* src, uint8_t* dst, const double* inv_mats, int N, int H, int W, int nearest,
                       int border, void* stream);
/* RandomGamma / RandomBrightnessContrast (pipeline:151-152) = cv2.LUT with one 256-entry table per frame            */
int aau_lut_u8(const uint8_t* src, uint8_t* dst, const uint8_t* luts /*[N][256]*/, int N, int64_t HW, void* stream);
/* ElasticTransform(alpha, sigma) (pipeline:153): out fp32 [N][2][H][W] uniform noise in [-1, 1), counter based        */
/* (keyed by seeds[n], plane, pixel); then aau_gauss_sep_f32 (GaussianBlur, BORDER_REFLECT_101, `ksize` taps, tmp of  */
/* the same size) and aau_remap_u8: dst(x,y) = src(x + alpha[n] dx, y + alpha[n] dy), BORDER_REFLECT_101              */
int aau_elastic_noise(const uint64_t* seeds, float* out, int N, int H, int W, void* stream);
int aau_gauss_sep_f32(const float* src, float* dst, float* tmp, const float* taps, int ksize, int64_t planes, int H,
                      int W, void* stream);
int aau_remap_u8(const uint8_t* src, uint8_t* dst, const float* disp, const float* alpha, int N, int H, int W,
                 int nearest, void* stream);
/* CLAHE / MedianBlur are drawn per frame (albumentations' default p = 0.5, pipeline:153,155): out[n] = flags[n] ? a : b */
int aau_select_frames_u8(const uint8_t* a, const uint8_t* b, const uint8_t* flags, uint8_t* out, int N, int64_t HW,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AAU_H_ */

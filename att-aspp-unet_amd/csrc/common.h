// Shared device/host helpers for the gfx950 kernels (internal; the ABI is include/aau.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "aau.h"

// 16-bit storage type of activations and packed weights.  The library is built twice from the same sources:
// libaau.so (bfloat16: training and inference, the metric's dtype) and libaau_f16.so (-DAAU_F16: IEEE half, for the
// reference's fp16 inference configuration -- no loss scaling exists here, so training stays bf16).  Kernels only
// touch the type through these names: `bf16x8` (an MFMA operand), AAU_MFMA16, bf2f / f2bf / pair_lo / pair_hi.
#ifdef AAU_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
#define AAU_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define AAU_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- transposed LDS reads as inline asm --------------------------------------------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr BUILTIN whenever an LDS-DMA
// (buffer_load ... lds / global_load_lds) is in flight: it treats the builtin as a read of memory the DMA may still be
// writing and cannot tell the ring slots apart.  That drains the prefetch of the NEXT K-step before the MFMAs of the
// current one start, i.e. no load / compute overlap inside a workgroup (plain ds_read_b128 loads do not trigger it).
// The compiler does not look inside an asm statement, so the kernels' own counted vmcnt + barrier stay the only
// wait.  What the asm form costs: the reads are invisible to hipcc's lgkmcnt bookkeeping as well, so every use sits
// behind an explicit AAU_LGKM_WAIT that names the destinations ("+v"; guide 5.7 item 1, form ii).
#define AAU_TR16(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
#define AAU_FRAG8(lo, hi) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))
// 32-bit LDS byte address of a __shared__ object (generic -> LDS address space)
#define AAU_LDS_ADDR(p) ((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)(p))

namespace aau {

// 16-B loads of out-of-image taps / rows past the tensor are redirected to this page
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

// Reduction workspaces are cleared by a kernel, never hipMemsetAsync: a memset node captured into a hipGraph
// stopped taking effect once the same range had been cleared by an eager hipMemsetAsync (ROCm 7.2, gfx950;
// round-1 reproduction script), which fed garbage partial sums to the replayed forward.
static __global__ void zero_f32_kernel(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}
static inline void zero_f32(float* p, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// ---- 16-bit storage <-> f32 (round to nearest even; plain casts keep NaN a NaN, see guide) ----
#ifdef AAU_F16
__device__ __forceinline__ float bf2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    _Float16 b = (_Float16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// the two values of a packed pair
__device__ __forceinline__ float pair_lo(unsigned u) { return bf2f((unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float pair_hi(unsigned u) { return bf2f((unsigned short)(u >> 16)); }
#else
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float pair_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pair_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const u32x4& v, float f[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = pair_lo(v[i]);
        f[2 * i + 1] = pair_hi(v[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float f[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
    return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15); every lane of the row ends with the total.
// Four VALU ops with DPP modifiers (quad swaps, half-mirror, mirror) -- no LDS crossbar traffic.
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
    return x;
}

// One output value of the first layer, Conv2d(1, C, 3, pad 1): a FIXED fma chain over the 9 taps, so that every kernel
// that recomputes z from the frame instead of reading it (the first layer's z is never stored) gets the same bits.
__device__ __forceinline__ float conv1_dot(const float v[9], const float* w9) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc = __builtin_fmaf(v[k], w9[k], acc);
    return acc;
}
// the 9 taps of pixel (yy, xx) of one fp32 image (zero padding)
__device__ __forceinline__ void conv1_taps(const float* img, int yy, int xx, int H, int W, float v[9]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y2 = yy + ky - 1, x2 = xx + kx - 1;
            v[ky * 3 + kx] = ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) ? img[y2 * W + x2] : 0.f;
        }
}

// ---- order-independent per-channel statistics -----------------------------------------------------------------------
// BatchNorm batch statistics are sums over thousands of workgroups.  fp32 atomics make them depend on the arrival
// order; through bf16 rounding and ReLU / max-pool routing that noise is amplified into visibly different logits and
// gradients from one run to the next (round 1: gradient cosine 0.99 between identical steps).  Integer adds are
// associative, so every partial sum is added as a TWO-LIMB FIXED-POINT number:
//     value = hi * 2^-8 + lo * 2^-52       (hi, lo: int64, 64-bit atomic adds)
// hi holds the partial rounded to a multiple of 2^-8 (|value| < 3.6e16), lo the remainder (|lo| <= 2^43 per add, so
// 2^19 adds per slot cannot overflow); a partial is represented exactly down to 2^-52 absolute.  The result is the
// same bit pattern for every arrival order.  A non-finite partial raises the buffer's poison word instead (the reader
// then returns NaN, which keeps the "skip the step on inf / nan" behaviour of pipeline:322-324).
// Layout of a statistics buffer for C channels: int64 [AAU_STAT_REPLICAS][2][C][2 limbs], then the poison word.
#define AAU_STAT_WORDS(C) ((size_t)AAU_STAT_REPLICAS * 2 * (size_t)(C) * 2 + 2)   /* int64 words incl. poison + pad */

__device__ __forceinline__ void stat_add(long long* base, int C, int replica, int which, int c, float v) {
    unsigned long long* slot = (unsigned long long*)base + (((size_t)replica * 2 + which) * C + c) * 2;
    if (!(fabsf(v) < 3.0e16f)) {          // inf, nan or beyond the fixed-point range
        atomicOr((unsigned long long*)base + (size_t)AAU_STAT_REPLICAS * 2 * C * 2, 1ull);
        return;
    }
    const double d = (double)v;
    const double h = rint(d * 256.0);
    const double r = d - h * (1.0 / 256.0);                              // exact
    const long long hi = (long long)h, lo = (long long)rint(r * 4503599627370496.0);
    if (hi) atomicAdd(slot, (unsigned long long)hi);
    if (lo) atomicAdd(slot + 1, (unsigned long long)lo);
}
// sum over the replicas of statistic `which` of channel c (NaN if the buffer is poisoned)
__device__ __forceinline__ double stat_total(const long long* base, int C, int which, int c) {
    if (base[(size_t)AAU_STAT_REPLICAS * 2 * C * 2] != 0) return __longlong_as_double(0x7ff8000000000000ll);
    long long hi = 0, lo = 0;
    for (int r = 0; r < AAU_STAT_REPLICAS; ++r) {
        const long long* s = base + (((size_t)r * 2 + which) * C + c) * 2;
        hi += s[0];
        lo += s[1];
    }
    return (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 4503599627370496.0);
}

// Epilogue store widening for the 16x16 MFMA accumulator layout (lane = 16 * fk + fr holds channels [4 fk, 4 fk + 4) of
// pixel fr: an 8-byte store per lane, sixteen 32-byte pieces per wave instruction).  Given the quads of TWO pixels A and
// B of the same 16-channel group, v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of
// the other (probe: scripts/probes/permlane16.hip), after which a lane with even fk owns channels [8 (fk >> 1), +8) of
// pixel A and a lane with odd fk the same channels of pixel B: ONE 16-byte store (or read-modify-write) per lane for the
// pair.  Same bytes, same addresses, half the vector-memory instructions (guide T21: such tails are issue-bound).
__device__ __forceinline__ void swap_pair8(const float a[4], const float b[4], float out[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        out[r] = __uint_as_float(t[0]);
        out[4 + r] = __uint_as_float(t[1]);
    }
}

// 16-byte buffer store with the row part of the address in the SCALAR offset.  gfx950 reads the four data registers of a
// buffer_store_dwordx4 AFTER the instruction has issued: a vector instruction that rewrites them in the next cycle reaches
// the store (seen here: dword 0 of lanes 12-15 of every row of 16 carried the NEXT accumulator pair's raw fp32 bits).
// hipcc pads that hazard itself only when the scalar offset is NOT a register (LLVM GCNHazardRecognizer::
// createsVALUHazard: "this hazard only exists if the instruction is not using a register in the soffset field"), so with
// the builtin + register soffset the emitted code had a v_mov into the data register directly behind the store -- wrong
// outputs whenever instruction fetch lets the two issue back to back (it came and went with code placement; round 3 took
// it for a miscounted vmcnt).  The store is therefore issued from an asm statement that carries its own wait states.
// (An asm store is not in hipcc's vmcnt bookkeeping: every wait that depends on these stores is hand-counted, NST below.)
__device__ __forceinline__ void store_b128_soff(u32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// 16-byte buffer load with a scalar offset, issued from asm so that hipcc does not count it: beside LDS-DMA builtins the
// compiler waits vmcnt(0) for every ordinary load (guide 5, trap b) and would drain the fill ring.  The CALLER waits, with
// a counted vmcnt and the destination as a "+v" operand of the wait statement (guide 5.7, form ii).
__device__ __forceinline__ u32x4 load_b128_soff(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}

// Workgroup part of a conv epilogue's statistics: every wave has stored its per-channel row sums in ITS OWN block of
// `sst` ([nwaves][2][BQ] floats, zero where a wave has no share); thread t < 2*BQ adds the blocks in wave order (a fixed
// order, unlike LDS atomics) and publishes one fixed-point add per channel and workgroup.
__device__ __forceinline__ void stats_publish(const float* sst, int nwaves, int BQ, int tid, int q0, int Cout,
                                              long long* stats, int replica) {
    if (tid < 2 * BQ) {
        const int which = tid / BQ, ql = tid - which * BQ;
        if (q0 + ql < Cout) {
            float v = 0.f;
            for (int w = 0; w < nwaves; ++w) v += sst[w * 2 * BQ + tid];
            stat_add(stats, Cout, replica, which, q0 + ql, v);
        }
    }
}

// ---- deterministic sums over the workgroups of a launch (BatchNorm backward, the head's parameter gradients) -------
// Every workgroup stores its row of n partial sums (n % 4 == 0) with plain stores into ws [nblk][n]; red_fold_launch
// (the kernel boundary is the only synchronisation) then adds the rows in row order and hands the totals out.  The
// order of the additions depends on the grid only: bitwise reproducible, no float atomics (the fp32-atomic form cost
// ~1.5 M atomics per deep-layer launch = its whole 30-us floor; an in-kernel last-arriver tree was tried and cost
// 20-120 us per launch: every workgroup's agent-scope release writes back the whole L2, dz stream included).
constexpr int AAU_BN_RED_MAX_BLOCKS = 2048;       // grid cap of every launch that writes rows (sizes aau_bn_red_ws_bytes)
__host__ __device__ inline size_t red_ws_floats(int n, int nblk) { return (size_t)nblk * n; }
__device__ __forceinline__ float* red_row(float* ws, int n, int blk) { return ws + (size_t)blk * n; }
// out[0 .. n_out) = totals (overwritten); acc[i] += total[n_out + i] for i < n_acc; acc2[0] += total[n_out + n_acc]
int red_fold_launch(const float* ws, int n, int nblk, float* out, int n_out, float* acc, int n_acc, float* acc2, hipStream_t s);

// Flat index -> coordinates with 32-bit unsigned division (indices stay below 2^31; the int64 form of % and / costs
// on the order of a hundred VALU instructions per pixel and showed up as 30-40 % of the pooled BN kernels).
__device__ __forceinline__ void decode3(int64_t i, int W, int H, int& x, int& y, int& n) {
    const unsigned u = (unsigned)i;
    const unsigned t = u / (unsigned)W;
    x = (int)(u - t * (unsigned)W);
    const unsigned nn = t / (unsigned)H;
    y = (int)(t - nn * (unsigned)H);
    n = (int)nn;
}

// First row of this workgroup's slice of a row-split launch.  A NEGATIVE rows-per-block encodes "walk the tensor
// from its end" (aau_traverse): workgroup b then owns slice gridDim.x-1-b.  Makes rpb positive.
template <typename T>
__device__ __forceinline__ int64_t slice_begin(T& rpb) {
    const bool rev = rpb < 0;
    if (rev) rpb = -rpb;
    return (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rpb;
}

// counter-based uniform in [0,1) for dropout: keyed by (seed, element index)
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// ---- host side: errors + launch profiling ----
int next_traversal();   // 1: this launch walks its tensors from the end (aau_traverse), see aau.h
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// names the kernel variant a launcher picked (and its algorithmic HBM bytes) on the live ProfScope of this thread
void prof_tag(const char* tag, double bytes = 0.0);

struct ProfScope {  // brackets one launch with events when profiling is on
    ProfScope(int family, double flops, hipStream_t s);
    ~ProfScope();
    int idx;
    hipStream_t stream;
};

// split-K reduction shared by wgrad.hip / wgrad3x3.hip (kernel in wgrad.hip)
struct WRedArgs {
    const float* ws;
    float* dw;
    int64_t nslots;
    int nsplit, P, NV;      // NV = accumulator vectors per thread of the producer (9 / 21 / 42)
    int sub, kwaves;        // MODE 0: TQ*TC sub-tiles and K-waves of the producer workgroup
    int TQ, TC, ntc, T;     // tile decode
    int Cout, Cin;
};
int wg_reduce_launch(int mode, WRedArgs& r, hipStream_t s);

}  // namespace aau

// size / alignment check of an aau_stat buffer handed in through the C ABI (NULL = no statistics wanted)
#define AAU_CHECK_STAT(fn, stats, bytes, C)                                                                            \
    AAU_REQUIRE((stats) == nullptr || ((int64_t)(bytes) >= (int64_t)(AAU_STAT_WORDS(C) * 8) && ((uintptr_t)(stats) & 7) == 0), \
                fn ": statistics buffer of %lld bytes, %lld needed for %d channels (8-byte aligned)", (long long)(bytes),    \
                (long long)(AAU_STAT_WORDS(C) * 8), (int)(C))

#define AAU_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            aau::set_error(__VA_ARGS__);            \
            return AAU_E_INVALID;                   \
        }                                           \
    } while (0)

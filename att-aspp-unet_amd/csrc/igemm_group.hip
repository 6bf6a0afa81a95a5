// Grouped implicit-GEMM: ONE output accumulates several convolutions of DIFFERENT sources,
//
//   dst[m][q] = sum_g sum_{tap,c} src_g[gather_g(m,tap)][c] * wpk_g[q][tap][c]
//
// -- the input gradient of the ASPP bridge (pipeline:80-83: the 1x1 and the three dilated 3x3 branches all read the same
// x, so dL/dx is the sum of their four data gradients).  Run one after the other (aau_conv_igemm with accumulate) each
// branch is a GEMM with M = 8192 pixels and only N = 384 output channels: 128 of the 128 x 192 tiles, or 512 quarter
// tiles at 55 FLOP per byte of LDS fill, plus a bf16 read-modify-write of dst per branch.  Grouped, the K-dimension is
// the concatenation of every branch's (tap, channel) list, the accumulators stay in registers across branches, and the
// list is cut into `nsplit` contiguous ranges so that tiles x nsplit covers the chip; the ranges leave fp32 slabs that
// one small kernel adds in a fixed order (deterministic, no atomics).
//
// The main loop is the wide tile of igemm.hip (128 pixels x 192 channels, 8 waves, register-staged fill, fragments read
// one sub-step ahead); what is new is the cursor (segment, tap, chunk).
#include "common.h"
#ifdef AAU_NO_MFMA_PIN         /* A/B build (build.py -DAAU_NO_MFMA_PIN --tag=nopin): the scheduler's own order */
#define AAU_PIN_SB()
#else
#define AAU_PIN_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#include <type_traits>

namespace aau {

constexpr int GMAX = 8;

struct GroupArgs {
    const unsigned short* src[GMAX];
    const unsigned short* wpk[GMAX];
    unsigned src_bytes[GMAX], wpk_bytes[GMAX];
    int src_pitch[GMAX], k[GMAX], dil[GMAX];
    int nseg, N, H, W, Cin, Cout, M, nchunk, nsplit, ntiles;
    float* ws;              // [nsplit][M][Cout] when nsplit > 1
    unsigned short* dst;
    int dst_pitch, accumulate;
};

__device__ __forceinline__ int gswz(int row, int lc) { return lc ^ ((row >> 1) & 7); }

__global__ __launch_bounds__(512) void igemm_group_kernel(const GroupArgs a) {
    constexpr int BK = 64, BQ = 192, BP = 128, NWV = 8;
    constexpr int SLOTS = BK / 8, RPI = 64 / SLOTS;
    constexpr int NA = BP / (NWV * RPI), NW = BQ / (NWV * RPI), NL = NA + NW;
    constexpr int MI = 4, NI = 3, WPX = MI * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    __shared__ unsigned s_mask[GMAX];
    auto sW = [&](int buf) -> unsigned short* { return smem + buf * ((BQ + BP) * BK); };
    auto sA = [&](int buf) -> unsigned short* { return smem + buf * ((BQ + BP) * BK) + BQ * BK; };

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wp = wave >> 2, wq = wave & 3;

    // split index slowest: the workgroups of one K-range (same taps, same weights) are neighbours in an XCD's L2
    const int ntq = a.Cout / BQ;
    const int nwg = gridDim.x;
    int bid = (int)blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int z = bid / a.ntiles;
    const int tile = bid - z * a.ntiles;
    const int tq = tile % ntq, tp = tile / ntq;
    const int q0 = tq * BQ, m0 = tp * BP;
    const int HW = a.H * a.W;
    constexpr unsigned OOB = 0x80000000u;

    // ---- rows of the activation gather (stride 1, same-size output) ----
    int pix0[NA], yx0[NA], lcA[NA];
    const int slot = lane % SLOTS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (i * NWV + wave) * RPI + lane / SLOTS;
        const int m = m0 + row;
        if (m < a.M) {
            const int n = m / HW;
            const int rem = m - n * HW;
            const int y = rem / a.W;
            pix0[i] = n * HW;
            yx0[i] = (y << 16) | (rem - y * a.W);
        } else {
            pix0[i] = 0;
            yx0[i] = (int)0x80000000;
        }
        lcA[i] = gswz(row, slot);
    }
    int wrow[NW], lcW[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int row = (j * NWV + wave) * RPI + lane / SLOTS;
        wrow[j] = q0 + row;
        lcW[j] = gswz(row, slot);
    }

    // ---- active taps of every segment for this tile (a tap none of whose rows lands in the image is skipped) ----
    if (tid < GMAX) s_mask[tid] = 0;
    __syncthreads();
    for (int g = 0; g < a.nseg; ++g) {
        const int k = a.k[g], dil = a.dil[g], half = k >> 1;
        unsigned mine = 0;
        for (int t = 0; t < k * k; ++t) {
            const int dy = (t / k - half) * dil, dx = (t % k - half) * dil;
            bool any = false;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int y = (yx0[i] >> 16) + dy, x = (yx0[i] & 0xffff) + dx;
                any |= (yx0[i] != (int)0x80000000) && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            }
            if (any) mine |= 1u << t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor(mine, o, 64);      // one LDS atomic per wave (igemm.hip)
        if (lane == 0 && mine) atomicOr(&s_mask[g], mine);
    }
    __syncthreads();

    // ---- this workgroup's range of the concatenated (segment, tap, chunk) list ----
    int total = 0;
    for (int g = 0; g < a.nseg; ++g) total += __builtin_popcount(s_mask[g]) * a.nchunk;
    total = __builtin_amdgcn_readfirstlane(total);
    int first = (int)((int64_t)total * z / a.nsplit);
    const int nsteps = (int)((int64_t)total * (z + 1) / a.nsplit) - first;

    // cursor
    int seg = 0, tap = 0, chunk = 0, kk_ = 1, dil_ = 1, pitch_ = 0;
    unsigned mask = 0;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0], 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk[0], 0, 0, 0x00020000);
    unsigned wbase[NW], abase[NA];
    auto enter_seg = [&](int g) {
        seg = g;
        kk_ = a.k[g];
        dil_ = a.dil[g];
        pitch_ = a.src_pitch[g];
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[g], 0, a.src_bytes[g], 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk[g], 0, a.wpk_bytes[g], 0x00020000);
        mask = __builtin_amdgcn_readfirstlane(s_mask[g]);
#pragma unroll
        for (int j = 0; j < NW; ++j) wbase[j] = (unsigned)((wrow[j] * kk_ * kk_ * a.Cin + lcW[j] * 8) * 2);
    };
    auto set_tap = [&](int t) {
        const int half = kk_ >> 1;
        const int dy = (t / kk_ - half) * dil_, dx = (t % kk_ - half) * dil_;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int y = (yx0[i] >> 16) + dy, x = (yx0[i] & 0xffff) + dx;
            const bool ok = (yx0[i] != (int)0x80000000) && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            abase[i] = ok ? (unsigned)(((pix0[i] + y * a.W + x) * pitch_ + lcA[i] * 8) * 2) : OOB;
        }
    };
    auto next_tap = [&]() {
        tap = __builtin_ctz(mask);
        mask &= mask - 1;
        set_tap(tap);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;

    if (nsteps > 0) {
        {   // position the cursor on step `first`
            int g = 0;
            while (true) {
                const int cnt = __builtin_popcount(s_mask[g]) * a.nchunk;
                if (first < cnt) break;
                first -= cnt;
                ++g;
            }
            enter_seg(g);
            const int ord = first / a.nchunk;
            chunk = first - ord * a.nchunk;
            for (int i = 0; i < ord; ++i) mask &= mask - 1;
            next_tap();
        }
        u32x4 R0[NL], R1[NL];
        auto gload = [&](u32x4 (&R)[NL]) {       // fetch the step at the cursor, then advance the cursor
            const int soffA = chunk * BK * 2, soffW = (tap * a.Cin + chunk * BK) * 2;
#pragma unroll
            for (int i = 0; i < NA; ++i) R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)abase[i], soffA, 0);
#pragma unroll
            for (int j = 0; j < NW; ++j) R[NA + j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)wbase[j], soffW, 0);
            if (++chunk == a.nchunk) {
                chunk = 0;
                if (mask) {
                    next_tap();
                } else if (seg + 1 < a.nseg) {
                    enter_seg(seg + 1);
                    next_tap();
                }
            }
        };
        auto lwrite = [&](int buf, const u32x4 (&R)[NL]) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *(u32x4*)(sA(buf) + (i * NWV + wave) * RPI * BK + lane * 8) = R[i];
#pragma unroll
            for (int j = 0; j < NW; ++j) *(u32x4*)(sW(buf) + (j * NWV + wave) * RPI * BK + lane * 8) = R[NA + j];
        };
        struct Frag { bf16x8 w[NI], a[MI]; };
        auto read_frags = [&](int buf, int kk, Frag& f) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wq * 48 + ni * 16 + fr;
                f.w[ni] = *(const bf16x8*)(sW(buf) + row * BK + gswz(row, kk * 4 + fk) * 8);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wp * WPX + mi * 16 + fr;
                f.a[mi] = *(const bf16x8*)(sA(buf) + row * BK + gswz(row, kk * 4 + fk) * 8);
            }
        };
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = AAU_MFMA16(f.w[ni], f.a[mi], acc[ni][mi], 0, 0, 0);
        };
        // same pipeline as igemm.hip's wide tile: see the comments there
        Frag F0, F1;
        auto iter = [&](int t, u32x4 (&Rnext)[NL], u32x4 (&Rfree)[NL], auto fetch, auto write) {
            read_frags(t & 1, 1, F1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(fetch)::value) gload(Rfree);
            AAU_PIN_SB();
            mma(F0);
            if constexpr (decltype(write)::value) {
                lwrite((t + 1) & 1, Rnext);
                // keep this group's MFMAs ABOVE the barrier, two in front of every LDS write (igemm.hip: left alone the
                // scheduler sinks them below it and the write / wait / barrier run with the matrix pipe idle)
#ifndef AAU_NO_MFMA_PIN      /* A/B build: the scheduler's own order */
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NI * MI - NL * 2, 0);
#endif
                AAU_PIN_SB();
                __syncthreads();
                read_frags((t + 1) & 1, 0, F0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(F1);
        };
        using Y = std::integral_constant<bool, true>;
        using N = std::integral_constant<bool, false>;
        gload(R0);
        if (nsteps > 1) gload(R1);
        lwrite(0, R0);
        __syncthreads();
        read_frags(0, 0, F0);
        int t = 0;
        for (; t + 3 < nsteps; t += 2) {
            iter(t, R1, R0, Y{}, Y{});
            iter(t + 1, R0, R1, Y{}, Y{});
        }
        if (t + 2 < nsteps) {
            iter(t, R1, R0, Y{}, Y{});
            iter(t + 1, R0, R1, N{}, Y{});
            iter(t + 2, R1, R0, N{}, N{});
        } else if (t + 1 < nsteps) {
            iter(t, R1, R0, N{}, Y{});
            iter(t + 1, R0, R1, N{}, N{});
        } else {
            iter(t, R1, R0, N{}, N{});
        }
    }

    // ---- epilogue: lane holds D[channel q0 + wq*48 + ni*16 + 4*fk + r][pixel m0 + wp*64 + mi*16 + fr] ----
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wp * WPX + mi * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = q0 + wq * 48 + ni * 16 + 4 * fk;
            if (a.nsplit > 1) {
                *(f32x4*)(a.ws + ((size_t)z * a.M + m) * a.Cout + q) = acc[ni][mi];
            } else {
                unsigned short* out = a.dst + (size_t)m * a.dst_pitch + q;
                float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
                if (a.accumulate) {
                    const u32x2 old = *(const u32x2*)out;
                    v[0] += pair_lo(old[0]);
                    v[1] += pair_hi(old[0]);
                    v[2] += pair_lo(old[1]);
                    v[3] += pair_hi(old[1]);
                }
                u32x2 pk;
                pk[0] = pack2(v[0], v[1]);
                pk[1] = pack2(v[2], v[3]);
                *(u32x2*)out = pk;
            }
        }
    }
}

// dst[m][c .. c+8) = bf16( (accumulate ? dst : 0) + slab_0 + slab_1 + ... ), slabs added in index order
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* ws, int nsplit, int M, int C, unsigned short* dst,
                                                       int dst_pitch, int accumulate) {
    const int cg = C >> 3;
    const int64_t n = (int64_t)M * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / cg), c = (int)(i - (int64_t)m * cg) * 8;
        float v[8];
        unsigned short* out = dst + (size_t)m * dst_pitch + c;
        if (accumulate) unpack8(*(const u32x4*)out, v);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        for (int s = 0; s < nsplit; ++s) {
            const float* p = ws + ((size_t)s * M + m) * C + c;
            const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += lo[j]; v[4 + j] += hi[j]; }
        }
        *(u32x4*)out = pack8(v);
    }
}

static bool group_seg_ok(const aau_conv_desc* d, const aau_conv_desc* d0) {
    return d->N == d0->N && d->H == d0->H && d->W == d0->W && d->Ho == d->H && d->Wo == d->W && d->Cin == d0->Cin &&
           d->Cout == d0->Cout && d->Cpad == d->Cin && d->dst_pitch == d0->dst_pitch && d->stride == 1 && d->KH == d->KW &&
           (d->KH == 1 || d->KH == 3) && d->dil >= 1 && d->pad == d->dil * (d->KH / 2) && d->src_pitch >= d->Cin &&
           d->src_pitch % 8 == 0 && !d->relu && !d->shuffle2x2 && d->src_split_c <= 0 && d->dst_split_c <= 0;
}

static int group_nsplit(const aau_conv_desc* d0) {
    const int64_t M = (int64_t)d0->N * d0->H * d0->W;
    const int64_t tiles = ((M + 127) / 128) * (d0->Cout / 192);
    int ns = (int)(256 / tiles);
    if (const char* e = getenv("AAU_GROUP_NSPLIT")) ns = atoi(e);       // experiment
    return ns < 1 ? 1 : (ns > 4 ? 4 : ns);
}

}  // namespace aau

static bool group_in_range(const aau_conv_desc* descs, int n) {
    using namespace aau;
    if (!descs || n < 2 || n > GMAX) return false;
    const aau_conv_desc* d0 = &descs[0];
    if (d0->Cout % 192 != 0 || d0->Cin % 64 != 0 || d0->dst_pitch % 8 != 0) return 0;
    const int64_t M = (int64_t)d0->N * d0->H * d0->W;
    if (M <= 0 || M * d0->Cout >= 0x7fffffff || d0->H >= 32768 || d0->W >= 32768) return 0;
    for (int i = 0; i < n; ++i) {
        if (!group_seg_ok(&descs[i], d0)) return 0;
        if ((((int64_t)M - 1) * descs[i].src_pitch + descs[i].Cin) * 2 >= 0x7fffffff) return 0;
        if ((int64_t)d0->Cout * descs[i].KH * descs[i].KW * d0->Cin * 2 >= 0x7fffffff) return 0;
        if (i > 0 && !descs[i].accumulate) return 0;        // every later segment adds to the first one's result
    }
    return true;
}

extern "C" int aau_conv_igemm_group_ok(const aau_conv_desc* descs, int n) {
    if (getenv("AAU_NO_IGEMM_GROUP") || !group_in_range(descs, n)) return 0;
    // worth it when the separate launches cannot fill the chip with wide tiles
    const int64_t M = (int64_t)descs[0].N * descs[0].H * descs[0].W;
    return ((M + 127) / 128) * (descs[0].Cout / 192) <= 192 ? 1 : 0;
}

extern "C" int64_t aau_conv_igemm_group_ws_bytes(const aau_conv_desc* descs, int n) {
    using namespace aau;
    if (!descs || n < 1) return 0;
    const int ns = group_nsplit(&descs[0]);
    return ns > 1 ? (int64_t)ns * descs[0].N * descs[0].H * descs[0].W * descs[0].Cout * 4 : 0;
}

extern "C" int aau_conv_igemm_group(const aau_conv_desc* descs, const aau_bf16* const* srcs, const aau_bf16* const* wpks,
                                    int n, aau_bf16* dst, float* ws, void* stream) {
    using namespace aau;
    AAU_REQUIRE(descs && srcs && wpks && dst && n >= 2 && n <= GMAX, "aau_conv_igemm_group: bad args (n=%d, at most %d segments)", n, GMAX);
    AAU_REQUIRE(group_in_range(descs, n), "aau_conv_igemm_group: the group is outside the kernel's range (see include/aau.h)");
    const aau_conv_desc* d0 = &descs[0];
    GroupArgs a;
    a.nseg = n;
    a.N = d0->N; a.H = d0->H; a.W = d0->W; a.Cin = d0->Cin; a.Cout = d0->Cout;
    a.M = d0->N * d0->H * d0->W;
    a.nchunk = d0->Cin / 64;
    a.nsplit = group_nsplit(d0);
    a.ntiles = ((a.M + 127) / 128) * (a.Cout / 192);
    a.dst = (unsigned short*)dst; a.dst_pitch = d0->dst_pitch; a.accumulate = d0->accumulate;
    a.ws = ws;
    AAU_REQUIRE(a.nsplit == 1 || (ws && ((uintptr_t)ws & 15) == 0), "aau_conv_igemm_group: %d K-ranges need the workspace (aau_conv_igemm_group_ws_bytes)", a.nsplit);
    AAU_REQUIRE(((uintptr_t)dst & 15) == 0, "aau_conv_igemm_group: dst must be 16-byte aligned");
    double flops = 0.0, bytes = 2.0 * a.M * (double)a.Cout * (a.accumulate ? 2 : 1);
    for (int i = 0; i < n; ++i) {
        const aau_conv_desc* d = &descs[i];
        AAU_REQUIRE(srcs[i] && wpks[i] && ((uintptr_t)srcs[i] & 15) == 0 && ((uintptr_t)wpks[i] & 15) == 0,
                    "aau_conv_igemm_group: segment %d: null or unaligned pointer", i);
        a.src[i] = (const unsigned short*)srcs[i];
        a.wpk[i] = (const unsigned short*)wpks[i];
        a.src_pitch[i] = d->src_pitch; a.k[i] = d->KH; a.dil[i] = d->dil;
        a.src_bytes[i] = (unsigned)((((int64_t)a.M - 1) * d->src_pitch + d->Cin) * 2);
        a.wpk_bytes[i] = (unsigned)((int64_t)a.Cout * d->KH * d->KW * a.Cin * 2);
        flops += 2.0 * a.M * (double)a.Cout * a.Cin * d->KH * d->KW;
        bytes += 2.0 * ((double)a.M * a.Cin + (double)a.Cout * d->KH * d->KW * a.Cin);
    }
    for (int i = n; i < GMAX; ++i) { a.src[i] = nullptr; a.wpk[i] = nullptr; a.src_bytes[i] = a.wpk_bytes[i] = 0; a.src_pitch[i] = 0; a.k[i] = 1; a.dil[i] = 1; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(0, flops, s);
    prof_tag("igemm_group<128,192>", bytes);
    constexpr size_t lds = (size_t)2 * (192 + 128) * 64 * 2;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)igemm_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("aau_conv_igemm_group: cannot reserve %zu bytes of LDS", lds);
            return AAU_E_INVALID;
        }
        attr = true;
    }
    hipLaunchKernelGGL(igemm_group_kernel, dim3((unsigned)(a.ntiles * a.nsplit)), dim3(512), lds, s, a);
    if (a.nsplit > 1) {
        const int64_t items = (int64_t)a.M * (a.Cout / 8);
        int64_t blocks = (items + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)ws, a.nsplit, a.M, a.Cout,
                           (unsigned short*)dst, a.dst_pitch, a.accumulate);
    }
    return check_launch("aau_conv_igemm_group");
}

// Strip-streaming 3x3 (stride 1, pad 1) convolution with REGISTER-RESIDENT weights for the high-resolution,
// few-channel layers (48 / 96 channels in and out: d1.1, d2.*, u1.conv.*, u2.conv.1 of pipeline:113-121, forward and
// data-gradient).
//
// Those layers are streaming problems (arithmetic intensity 216-430 FLOP/B): what bounds them is how many bytes a CU
// keeps in flight, not the MFMA.  The resident-weight kernels of conv3x3.hip keep the packed weights in LDS (83-110 KB),
// which leaves room for ONE 24-KB halo tile of look-ahead per CU: 2.7 TB/s chip-wide (MFMA 0.35, wait
// 0.45, HBM a third of peak).  Here the weights live in REGISTERS instead: a wave owns one 16-channel output group and
// holds all of its 9-tap x Cin fragments (60 / 108 VGPRs) for the life of the kernel, so
//   * LDS holds nothing but pixels: a ring of R image rows of an 18-pixel-wide column STRIP; the workgroup marches down
//     the strip, 16 new rows per 16x16 patch (no vertical halo re-fetch), with 16-32 rows = 55 KB of LDS-DMA in flight;
//   * no weight fragment is ever read from LDS: 6 pixel-fragment reads per 12 MFMAs.
// Pixels are stored channel-contiguous ([row][18 px][Cin], pixel stride 96 B / 224 B = 6 / 14 sixteen-byte units: any
// stride = 2 mod 4 units is bank-conflict free for the ds_read_b128 lane groups of gfx950), so for one vertical tap the
// three horizontal taps of an output pixel are ONE contiguous K run (Cin = 48: 144 elements = 4.5 K-blocks, the half
// block multiplies zero weights) and every LDS address is a row base + an immediate.
// 12 waves (3 per SIMD, <= 168 VGPRs): wave = (16-channel group g, row-quad slot); one barrier per patch.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "c3args.h"

namespace aau {

template <int N>
__device__ __forceinline__ void wait_vm_s() {
    static_assert(N >= 0 && N <= 10, "count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// Epilogue of one pair of accumulator quads (c3args.h: epi_pair_wide) with the store as a buffer store: statistics, bias,
// folded-BN affine (staged in LDS as [3][96] floats bias | scale | shift; par = this lane's channel quad in it) in the MFMA layout, the cross-lane swap, ReLU, one
// 16-byte store at rsD[voff + soff].
// BNRED: (zA, zB) are the 4 raw conv outputs of the CONSUMING BatchNorm layer at this lane's two pixels (MFMA layout); the
// statistics are that layer's backward sums, par then holds [4][96] floats scale | shift | mean | invstd of that layer.
template <bool BNRED>
__device__ __forceinline__ void epi_pair_store(const C3Args& a, const aau_conv_desc& d, int q, const f32x4& accA, const f32x4& accB,
                                               bool want_stats, float s1[4], float s2[4], const unsigned char* par,
                                               __amdgpu_buffer_rsrc_t rsD, unsigned voff, unsigned soff, u32x2 zA = u32x2{0, 0},
                                               u32x2 zB = u32x2{0, 0}) {
    float va[4], vb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { va[r] = accA[r]; vb[r] = accB[r]; }
    if constexpr (BNRED) {
        const f32x4 sc = *(const f32x4*)(par), sh = *(const f32x4*)(par + 384), mu = *(const f32x4*)(par + 768),
                    is = *(const f32x4*)(par + 1152);
        const float za[4] = {pair_lo(zA[0]), pair_hi(zA[0]), pair_lo(zA[1]), pair_hi(zA[1])};
        const float zb[4] = {pair_lo(zB[0]), pair_hi(zB[0]), pair_lo(zB[1]), pair_hi(zB[1])};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // the gradient as the separate reduce pass would read it back: rounded to the 16-bit storage type
            const float ga = (za[r] * sc[r] + sh[r] > 0.f) ? bf2f(f2bf(va[r])) : 0.f;
            const float gb = (zb[r] * sc[r] + sh[r] > 0.f) ? bf2f(f2bf(vb[r])) : 0.f;
            s1[r] += ga + gb;
            s2[r] += ga * ((za[r] - mu[r]) * is[r]) + gb * ((zb[r] - mu[r]) * is[r]);
        }
    } else if (want_stats) { epi_stats(a, 0, q, va, s1, s2); epi_stats(a, 0, q, vb, s1, s2); }
    if (!BNRED && a.bias) {
        const f32x4 b = *(const f32x4*)(par);
#pragma unroll
        for (int r = 0; r < 4; ++r) { va[r] += b[r]; vb[r] += b[r]; }
    }
    if (!BNRED && a.scale) {
        const f32x4 sc = *(const f32x4*)(par + 384);
        const f32x4 sh = *(const f32x4*)(par + 768);
#pragma unroll
        for (int r = 0; r < 4; ++r) { va[r] = va[r] * sc[r] + sh[r]; vb[r] = vb[r] * sc[r] + sh[r]; }
    }
    float w[8];
    swap_pair8(va, vb, w);
    if (d.relu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = fmaxf(w[r], 0.f);
    }
    store_b128_soff(pack8(w), rsD, voff, soff);
}

// A work unit is a vertical segment of one 16-pixel-wide strip of one image: `segh` patches (the last segment of a strip
// may be shorter).  The fill stream of a unit is blocks b = -1 .. K-1 of 16 image rows: block b holds rows
// ys + 16 b + 1 .. ys + 16 b + 16, patch b (rows ys + 16 b ...) needs the last two rows of block b-1 and all of block b.
// NWV = 12 waves per workgroup (one workgroup per CU), PR = 16 patch rows.  (Two independent 6-wave workgroups per CU on
// 8-row patches measured 20-30 % slower -- twice the barriers, 10 halo rows per 8 -- and an epilogue staggered between the
// two halves of the waves 0-10 % slower: profiles/NOTES.md.)
// BNIN (aau_conv_igemm_bnin): the source tensor is the raw output z of the previous ConvBNReLU layer and
// y = relu(z * scale + shift) is applied IN LDS: every lane transforms exactly the 16-byte pieces its own LDS-DMA
// instructions fetched (it knows their addresses, and its own counted vmcnt says they have landed: no extra barrier),
// one block ahead of the block the MFMAs read -- at the end of a step, behind the epilogue, when the accumulators are
// dead.  Pieces outside the image stay the zeros the range check delivered (the padding of y is zero, not relu(shift)).
// The arithmetic is aau_bn_act's (fp32 fma, max, round to the 16-bit type), so the result equals the two-kernel path bit
// for bit while the 2 x 201 MB round trip of y at level 1 (2 x 50 MB at level 2) is gone.
template <int CIN, int G, int NWV, int PR, int ABL, bool BNRED, bool BNIN = false>
__global__ __launch_bounds__(64 * NWV, 3) void conv3x3s_kernel(const C3Args a, int nunits, int strips, int nseg, int segh) {
    constexpr int abl = ABL;                           // timing ablations (AAU_C3S_ABL; builds with -DAAU_C3S_ABLATE only)
    constexpr int PXB = CIN == 48 ? 96 : 224;          // bytes per pixel in LDS
    constexpr int SU = PXB / 16;                       // ... in 16-byte units
    constexpr int NB = CIN == 48 ? 5 : 9;              // K-blocks per vertical tap
    constexpr int ROWB = CIN == 48 ? 2048 : 4096;      // bytes per ring row (18 px, padded to whole 1-KiB DMA pieces)
    constexpr int IPR = ROWB / 1024;                   // LDS-DMA instructions per row
    constexpr int D = CIN == 48 ? 2 : 1;               // blocks in flight
    static_assert(PR == 16, "ring sizes below");
    constexpr int R = CIN == 48 ? 64 : 36;             // ring rows
    constexpr int FW = NWV / IPR * IPR;                // waves that issue fills (a lane's piece of a row must not depend on i)
    constexpr int NF = (PR * IPR + FW - 1) / FW;       // fill instructions per wave and block
    constexpr int SLOTS = NWV / G;                     // row-quad slots
    constexpr int NQ = (PR / 4) / SLOTS;               // row quads per wave and patch
    static_assert(SLOTS >= 1 && NQ >= 1 && NQ * SLOTS * 4 == PR, "waves x quads must tile the patch");
    constexpr int NST = NQ * 2;                        // 16-byte stores per lane and patch
    constexpr int BQ = 16 * G;
    constexpr int PF = CIN == 48 ? 6 : 3;              // pixel fragments in flight per wave
    static_assert(NF <= NB, "the fills of a block are issued between the K-blocks of one row quad");
    constexpr unsigned OOB = 0x80000000u;
    static_assert(R >= PR + 2 + PR * D, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // [R][ROWB] ring | 1 KiB scratch | [4][96] floats | BNIN: [2][96] floats
    static_assert(!(BNIN && BNRED), "one fused form at a time");

    const aau_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    const int g = wave % G, slot = wave / G;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const int tiles_y = d.H / PR;

    // ---- weights of this wave's 16-channel group: registers, once ----
    bf16x8 wr[3][NB];
    {
        const int q = g * 16 + fr;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int tx, c0;
                if constexpr (CIN == 48) {
                    const int t = 4 * j + fk;           // 16-byte unit inside the 3-pixel K run
                    tx = t / SU; c0 = (t - tx * SU) * 8;
                } else {
                    tx = j / 3; c0 = (j - tx * 3) * 32 + fk * 8;
                }
                bf16x8 v = {};
                if (tx < 3 && q < d.Cout) v = *(const bf16x8*)(a.wpk + ((size_t)(q * 9 + ty * 3 + tx) * d.Cpad + c0));
                wr[ty][j] = v;
            }
    }

    // ---- fill roles: this lane always fetches the same (pixel, 16-byte part) of a row; the row is wave-uniform ----
    const int sub = wave % IPR;
    const int piece = sub * 64 + lane;
    const int fpx = piece / SU, fpart = piece - fpx * SU;
    const bool lane_ok = fpx < 18 && fpart * 8 < CIN;
    const int split_c = d.src_split_c > 0 ? d.src_split_c : 0x7fffffff;
    const int sadj = (fpart * 8 >= split_c) ? d.src_split_off - d.src_split_c : 0;
    const int dsplit_c = d.dst_split_c > 0 ? d.dst_split_c : 0x7fffffff;
    const int dsplit_adj = d.dst_split_off - d.dst_split_c;

    auto unit_decode = [&](int u, int& n, int& x0, int& ys, int& K) {
        if (a.rev) u = nunits - 1 - u;
        const int seg = u % nseg;
        const int t2 = u / nseg;
        const int strip = t2 % strips;
        n = t2 / strips; x0 = strip * 16; ys = seg * segh * PR;
        K = tiles_y - seg * segh; if (K > segh) K = segh;
    };

    // issue cursor
    int iu = blockIdx.x, ib = -1, in_ = 0, ix0 = 0, iys = 0, iK = 0;
    unsigned ivec = OOB;                       // per-lane byte offset of (pixel, part) inside a row, OOB outside the image
    auto issue_unit_setup = [&]() {
        if (iu < nunits) {
            unit_decode(iu, in_, ix0, iys, iK);
            const int x = ix0 - 1 + fpx;
            ivec = (lane_ok && (unsigned)x < (unsigned)d.W) ? (unsigned)((x * d.src_pitch + fpart * 8 + sadj) * 2) : OOB;
        }
    };
    issue_unit_setup();
    int irb = 0;                               // ring row of the next block to issue
    unsigned vq = 0;                           // BNIN: one bit per issued piece of this lane, oldest in bit 0: inside the image?
    int vqn = 0;
    // one LDS-DMA instruction (1 KiB: a quarter / half of a ring row) of the block under the issue cursor
    auto issue_one = [&](int i) {
        const bool live = iu < nunits;
        const int idx = i * FW + wave;
        const bool real = wave < FW && idx < PR * IPR;                 // spare slots write zeros into the scratch KiB
        const int r = idx / IPR;                                       // row of the block (wave-uniform)
        const int y = iys + PR * ib + 1 + r;
        const bool row_ok = live && real && (unsigned)y < (unsigned)d.H && (ib >= 0 || r >= PR - 2);
        int rr = irb + r; if (rr >= R) rr -= R;
        unsigned char* dstp = real ? dsm + rr * ROWB + sub * 1024 : dsm + R * ROWB;
        const unsigned soff = row_ok ? (unsigned)(((in_ * d.H + y) * d.W) * d.src_pitch * 2) : 0u;
        const unsigned v = (row_ok && !(abl & 2)) ? ivec : OOB;       // abl: timing ablations (AAU_C3S_ABL), never set in production
        if (!(abl & 32)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dstp), 16, (int)v, (int)soff, 0, 0);
        if constexpr (BNIN) { vq |= (real && v != OOB ? 1u : 0u) << vqn; ++vqn; }
    };
    auto issue_end = [&]() {
        irb += PR; if (irb >= R) irb -= R;
        if (iu < nunits) {
            if (++ib == iK) { iu += gridDim.x; ib = -1; issue_unit_setup(); }
        }
    };
    auto issue_block = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) issue_one(i);
        issue_end();
    };

    // per-channel epilogue constants of the inference form, staged once (no global loads between the fills and the stores)
    float* par = (float*)(dsm + R * ROWB + 1024);
    if constexpr (BNRED) {
        for (int i = tid; i < BQ; i += 64 * NWV) {
            par[i] = a.bn_scale[i]; par[96 + i] = a.bn_shift[i]; par[192 + i] = a.bn_mean[i]; par[288 + i] = a.bn_invstd[i];
        }
        __syncthreads();
    } else if (a.bias || a.scale) {
        for (int i = tid; i < BQ; i += 64 * NWV) {
            par[i] = a.bias ? a.bias[i] : 0.f;
            par[96 + i] = a.scale ? a.scale[i] : 1.f;
            par[192 + i] = a.scale ? a.shift[i] : 0.f;
        }
        __syncthreads();
    }
    float* inp = par + 4 * 96;                 // BNIN: [2][96] floats scale | shift of the INPUT channels
    if constexpr (BNIN) {
        for (int i = tid; i < CIN; i += 64 * NWV) { inp[i] = a.in_scale[i]; inp[96 + i] = a.in_shift[i]; }
        __syncthreads();
    }
    int trb = 0;                               // BNIN: ring row of the next block to transform (stream order)
    // y = relu(z * scale + shift) on this lane's own pieces of the oldest untransformed block (they have landed: caller)
    auto xform_block = [&]() {
        if constexpr (BNIN) {
            float sc[8], sh[8];
            const int c8 = fpart * 8 < CIN ? fpart * 8 : 0;
            *(f32x4*)(sc) = *(const f32x4*)(inp + c8); *(f32x4*)(sc + 4) = *(const f32x4*)(inp + c8 + 4);
            *(f32x4*)(sh) = *(const f32x4*)(inp + 96 + c8); *(f32x4*)(sh + 4) = *(const f32x4*)(inp + 96 + c8 + 4);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int idx = i * FW + wave;
                const int r = idx / IPR;
                int rr = trb + r; if (rr >= R) rr -= R;
                if ((vq >> i) & 1u) {
                    u32x4* pz = (u32x4*)(dsm + rr * ROWB + sub * 1024 + lane * 16);
                    float f[8];
                    unpack8(*pz, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
                    *pz = pack8(f);
                }
            }
            vq >>= NF; vqn -= NF;
            trb += PR; if (trb >= R) trb -= R;
        }
    };
    const unsigned par_off = (unsigned)(R * ROWB + 1024 + (g * 16 + 4 * fk) * 4);
    const bool want_stats = a.stats != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int lane_off = fr * PXB + fk * 16;
    // after the cross-lane swap of an accumulator pair this lane owns channels qw .. qw+7 of the pixel (row + (fk & 1), fr)
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, 0x7fffffff, 0x00020000);
    const int qw_ = g * 16 + 8 * (fk >> 1);
    const unsigned st_voff = (abl & 4) ? 0x80000000u : (unsigned)((((fk & 1) * d.W + fr) * d.dst_pitch + qw_ + (qw_ >= dsplit_c ? dsplit_adj : 0)) * 2);

#pragma unroll
    for (int i = 0; i < D; ++i) issue_block();
    if constexpr (BNIN) {                      // the first block of the stream: every later one is transformed one step ahead
        wait_vm_s<NF * (D - 1)>();
        xform_block();
    }

    int crb = 0;                               // ring row of block t
    bool prev_patch = false;
    f32x4 acc[4];
    // BNRED: the consuming layer's raw conv outputs at this lane's (pixel, 4 channels) of the four rows of the quad, loaded
    // BEFORE the quad's MFMA stream (their latency hides behind it; they are older than the fills issued during the
    // stream, so the compiler's wait for them leaves the fills in flight)
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)(BNRED ? a.bn_z : a.dst), 0, 0x7fffffff, 0x00020000);
    const unsigned z_voff = (unsigned)((fr * a.bn_zp + g * 16 + 4 * fk) * 2);
    u32x2 zq[4];
    auto load_z = [&](int n_, int y_, int x0_) {
        if constexpr (BNRED) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                zq[mi] = __builtin_amdgcn_raw_buffer_load_b64(rsZ, (int)z_voff, (int)((((n_ * d.H + y_ + mi) * d.W) + x0_) * a.bn_zp * 2), 0);
        }
    };
    auto epilogue = [&](unsigned soff0) {
        if constexpr (BNRED) {
            // the z loads are followed by exactly the NF fills of this step (one row quad per wave in this form): wait for
            // them by hand and make every later use of zq depend on the wait.  (hipcc's own count in front of the first use
            // was too large by one with LDS-DMA builtins and asm waits in the stream: z landed late, into registers the
            // allocator had already handed to the store data.)
            static_assert(NQ == 1 && NF == 3, "the hand-counted wait below");
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(zq[0]), "+v"(zq[1]), "+v"(zq[2]), "+v"(zq[3]) : : "memory");
        }
#pragma unroll
        for (int mp = 0; mp < 4; mp += 2)
            epi_pair_store<BNRED>(a, d, g * 16 + 4 * fk, acc[mp], acc[mp + 1], want_stats, s1, s2, dsm + par_off, rsD, st_voff,
                                  soff0 + (unsigned)(mp * d.W * d.dst_pitch * 2), zq[mp], zq[mp + 1]);
    };
    for (int cu = blockIdx.x; cu < nunits; cu += gridDim.x) {
        int n, x0, ys, K;
        unit_decode(cu, n, x0, ys, K);
        for (int cb = -1; cb < K; ++cb) {
            // block t has landed: the only younger operations are the fills of the blocks behind it and the previous
            // patch's stores (vmcnt retires in issue order)
            if (!(abl & 16)) {
                if (!prev_patch) wait_vm_s<NF * (D - 1)>();
                else wait_vm_s<NF * (D - 1) + NST>();
                if constexpr (BNIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's transformed pieces are written
                __builtin_amdgcn_s_barrier();
            }
            // block t + D goes into rows nobody reads any more; on a patch step its NF instructions are issued between
            // the K-blocks of the first row quad instead of in one burst behind the barrier
            prev_patch = cb >= 0;
            if (cb < 0) issue_block();
            if (cb >= 0) {
                const int y0 = ys + PR * cb;
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const int q = slot + qi * SLOTS;
                    unsigned va[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        int rr = crb + R - 2 + 4 * q + r;
                        rr -= (rr >= 2 * R) ? 2 * R : (rr >= R ? R : 0);
                        va[r] = (unsigned)(rr * ROWB + lane_off);
                    }
                    load_z(n, y0 + 4 * q, x0);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // pixel fragments in program order s = 6 j + r, read PF ahead of their MFMAs
                    constexpr int NS = 6 * NB;
                    auto rd = [&](int s_) -> bf16x8 {
                        const int j = s_ / 6, r = s_ - 6 * j;
                        const int off = CIN == 48 ? 64 * j : (j / 3) * PXB + (j % 3) * 64;
                        return *(const bf16x8*)(dsm + va[r] + off);
                    };
                    bf16x8 af[PF];
#pragma unroll
                    for (int s_ = 0; s_ < PF; ++s_) af[s_] = rd(s_);
#pragma unroll
                    for (int s_ = 0; s_ < NS; ++s_) {
                        const int j = s_ / 6, r = s_ - 6 * j;
                        const bf16x8 cur = (abl & 64) ? wr[0][0] : af[s_ % PF];      // 64: MFMAs without LDS reads
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) {
                            const int mi = r - ty;
                            if (mi >= 0 && mi < 4 && !(abl & 1)) acc[mi] = AAU_MFMA16(wr[ty][j], cur, acc[mi], 0, 0, 0);
                        }
                        if (s_ + PF < NS) af[s_ % PF] = rd(s_ + PF);
                        if (qi == 0 && r == 5 && j < NF) issue_one(j);
                    }
                    if (qi == 0) issue_end();
                    // epilogue of the quad: rows in pairs, one 16-byte buffer store per lane and pair -- the per-lane part
                    // of the address is a kernel constant, the rest a scalar offset: no 64-bit address registers
                    if (abl & 8) {                 // ablation: no epilogue (keep the accumulators alive)
                        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) s1[0] += 1.f;
                        continue;
                    }
                    const unsigned soff = (unsigned)((((n * d.H + y0 + 4 * q) * d.W) + x0) * d.dst_pitch * 2);
                    epilogue(soff);
                }
            }
            if constexpr (BNIN) {
                // the NEXT block of the stream: its fills are older than the NF * (D - 1) fills issued since and this step's
                // stores, so the same counts as at the top of the next step say that they have landed
                if (cb >= 0) wait_vm_s<NF * (D - 1) + NST>();
                else wait_vm_s<NF * (D - 1)>();
                xform_block();
            }
            crb += PR; if (crb >= R) crb -= R;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (want_stats) {
        // per-wave row sums (DPP) into the wave's OWN block of LDS, combined in wave order, then one order-independent
        // fixed-point add per channel and workgroup (common.h: stat_add)
        float* sst = (float*)dsm;                      // [NWV][2][BQ]
        __syncthreads();
        for (int i = tid; i < NWV * 2 * BQ; i += (int)blockDim.x) sst[i] = 0.f;
        __syncthreads();
        float* mine = sst + wave * 2 * BQ;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = row16_sum(s1[r]), x2 = row16_sum(s2[r]);
            if (fr == 0) {
                mine[g * 16 + 4 * fk + r] = x1;
                mine[BQ + g * 16 + 4 * fk + r] = x2;
            }
        }
        __syncthreads();
        stats_publish(sst, NWV, BQ, tid, 0, d.Cout, (long long*)a.stats, (int)(blockIdx.x % AAU_STAT_REPLICAS));
    }
}

// true when the strip-streaming kernel serves this descriptor (conv3x3_applicable has already passed)
bool conv3x3s_applicable(const aau_conv_desc* d, const void* src, const void* dst) {
    if (getenv("AAU_NO_C3S")) return false;
    if (!((d->Cin == 48 || d->Cin == 96) && (d->Cout == 48 || d->Cout == 96))) return false;
    if (d->Cpad != (d->Cin == 48 ? 64 : 96) || d->accumulate) return false;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || d->src_pitch % 8 || d->dst_pitch % 8) return false;
    if (d->src_split_c > 0 && (d->src_split_c % 8 || d->src_split_off % 8)) return false;
    if (d->dst_split_c > 0 && (d->dst_split_c % 8 || d->dst_split_off % 8)) return false;
    int minp = 1;
    if (const char* e = getenv("AAU_C3S_MINPATCH")) minp = atoi(e);
    return (int64_t)d->N * (d->H / 16) * (d->W / 16) >= minp;
}

int conv3x3s_launch(C3Args& a, hipStream_t s) {
    const aau_conv_desc& d = a.d;
    const bool c48 = d.Cin == 48, g3 = d.Cout == 48;
    constexpr int PR = 16, per_cu = 1;          // one 12-wave workgroup per CU on 16-row patches
    const int strips = d.W / 16, tiles_y = d.H / PR;
    // cut the strips into vertical segments until every workgroup slot has a unit (each segment restarts the row stream:
    // at least two patches per segment where the image allows it)
    int nseg = 1;
    while ((int64_t)d.N * strips * nseg < 256 * per_cu && tiles_y / (nseg * 2) >= 2) nseg *= 2;
    const int segh = (tiles_y + nseg - 1) / nseg;
    nseg = (tiles_y + segh - 1) / segh;
    const int64_t nunits = (int64_t)d.N * strips * nseg;
    if (nunits > 0x7fffffff) { set_error("conv3x3s: too many units"); return AAU_E_INVALID; }
    const int grid = nunits < 256 * per_cu ? (int)nunits : 256 * per_cu;
    const size_t ring = c48 ? (size_t)64 * 2048 : (size_t)36 * 4096;
    const size_t lds = ring + 1024 + 4 * 96 * 4 + 2 * 96 * 4;
    int abl = 0;
    if (const char* e = getenv("AAU_C3S_ABL")) abl = atoi(e);       // timing ablations: 1 no MFMA, 2 fills out of range, 4 no stores, 8 no epilogue, 16 no barrier / wait, 32 no fill instructions
    prof_tag(c48 ? (g3 ? "conv3x3s<48,48>" : "conv3x3s<48,96>") : (g3 ? "conv3x3s<96,48>" : "conv3x3s<96,96>"));
    auto go = [&](auto kern, int threads) {
        static bool attr = false;               // one flag per instantiation of this generic lambda
        if (!attr) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, a, (int)nunits, strips, nseg, segh);
    };
    if (a.bn_z) {          // fused BatchNorm-backward sums (aau_conv_igemm_bnred): the default form only, 48 input channels
        go(conv3x3s_kernel<48, 3, 12, 16, 0, true>, 768);      // 48 -> 48 (the 48 -> 96 form spills at 168 VGPRs)
        return check_launch("aau_conv_igemm_bnred(3x3 strips)");
    }
    if (a.in_scale) {      // BatchNorm + ReLU of the producing layer applied on the input in LDS (aau_conv_igemm_bnin)
        if (c48 && g3) go(conv3x3s_kernel<48, 3, 12, 16, 0, false, true>, 768);
        else if (c48) go(conv3x3s_kernel<48, 6, 12, 16, 0, false, true>, 768);
        else if (g3) go(conv3x3s_kernel<96, 3, 12, 16, 0, false, true>, 768);
        else go(conv3x3s_kernel<96, 6, 12, 16, 0, false, true>, 768);
        return check_launch("aau_conv_igemm_bnin(3x3 strips)");
    }
    auto pick = [&](auto ablc) {
        constexpr int A = decltype(ablc)::value;
        if (c48 && g3) go(conv3x3s_kernel<48, 3, 12, 16, A, false>, 768);
        else if (c48) go(conv3x3s_kernel<48, 6, 12, 16, A, false>, 768);
        else if (g3) go(conv3x3s_kernel<96, 3, 12, 16, A, false>, 768);
        else go(conv3x3s_kernel<96, 6, 12, 16, A, false>, 768);
    };
    switch (abl) {
#ifdef AAU_C3S_ABLATE
#define AAU_ABL_CASE(v) case v: pick(std::integral_constant<int, v>{}); break;
        AAU_ABL_CASE(1) AAU_ABL_CASE(2) AAU_ABL_CASE(4) AAU_ABL_CASE(6) AAU_ABL_CASE(8) AAU_ABL_CASE(14) AAU_ABL_CASE(16)
        AAU_ABL_CASE(24) AAU_ABL_CASE(32) AAU_ABL_CASE(46) AAU_ABL_CASE(47) AAU_ABL_CASE(63) AAU_ABL_CASE(64) AAU_ABL_CASE(70)
        AAU_ABL_CASE(78) AAU_ABL_CASE(110) AAU_ABL_CASE(126)
#undef AAU_ABL_CASE
#endif
        default: pick(std::integral_constant<int, 0>{}); break;
    }
    return check_launch("aau_conv_igemm(3x3 strips, register-resident weights)");
}

}  // namespace aau

using namespace aau;

namespace aau { bool conv3x3_applicable(const aau_conv_desc* d); }

// 1 when aau_conv_igemm_bnred serves this descriptor: a 48 -> 48 channel 3x3 data-gradient conv that the strip kernel takes
extern "C" int aau_conv_bnred_ok(const aau_conv_desc* d) {
    if (!d || getenv("AAU_NO_BNRED")) return 0;
    return conv3x3_applicable(d) && d->Cin == 48 && d->Cout == 48 && d->Cpad == 64 && !d->accumulate && !d->relu &&
           d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0 && d->src_split_c <= 0 && d->dst_split_c <= 0 && !getenv("AAU_NO_C3S");
}

extern "C" int aau_conv_igemm_bnred(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst,
                                    const aau_bf16* z, int z_pitch, const float* scale, const float* shift, const float* save_mean,
                                    const float* save_invstd, aau_stat* sums, int64_t sums_bytes, void* stream) {
    AAU_REQUIRE(d && src && wpk && dst && z && scale && shift && save_mean && save_invstd && sums, "aau_conv_igemm_bnred: null pointer");
    AAU_REQUIRE(aau_conv_bnred_ok(d) && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && ((uintptr_t)z & 7) == 0 && z_pitch >= d->Cout &&
                    z_pitch % 4 == 0,
                "aau_conv_igemm_bnred: descriptor not served (aau_conv_bnred_ok) or misaligned operands");
    AAU_CHECK_STAT("aau_conv_igemm_bnred", sums, sums_bytes, d->Cout);
    const int64_t M = (int64_t)d->N * d->H * d->W;
    const int64_t src_bytes = ((M - 1) * d->src_pitch + d->Cin) * 2, z_bytes = ((M - 1) * z_pitch + d->Cout) * 2;
    AAU_REQUIRE(src_bytes < 0x7fffffff && z_bytes < 0x7fffffff && M * d->dst_pitch * 2 < 0x7fffffff,
                "aau_conv_igemm_bnred: tensors must stay below 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(0, 2.0 * M * (double)d->Cout * d->Cin * 9, s);
    prof_tag(nullptr, 2.0 * ((double)M * d->Cin + 2.0 * (double)M * d->Cout + (double)d->Cout * 9 * d->Cin));
    C3Args a;
    a.d = *d;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.stats = (float*)sums;
    a.rev = next_traversal();
    a.nchunk = d->Cpad / 32;
    a.src_bytes = (unsigned)src_bytes;
    a.wpk_bytes = (unsigned)((int64_t)d->Cout * 9 * d->Cpad * 2);
    a.tiles_x = d->W / 16; a.tiles_y = d->H / 16;
    a.nowide = 0; a.nopair = 0;
    a.bn_z = z; a.bn_zp = z_pitch; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = save_mean; a.bn_invstd = save_invstd;
    return conv3x3s_launch(a, s);
}


namespace aau {
bool conv1x1_resw_applicable(const aau_conv_desc* d, bool want_stats);
int conv1x1_resw_launch(const aau_conv_desc* d, const aau_bf16* src, const aau_bf16* wpk, aau_bf16* dst, const float* bias,
                        const float* scale, const float* shift, float* stats, unsigned src_bytes, unsigned wpk_bytes,
                        hipStream_t s, const float* in_scale, const float* in_shift);
}
// the 1x1 / ConvTranspose-forward form: what conv1x1_rs_kernel takes (conv3x3.hip: 16-byte stores, at most six 32-channel chunks)
static bool bnin_1x1_ok(const aau_conv_desc* d) {
    return conv1x1_resw_applicable(d, false) && d->Cpad / 32 <= 6 && !d->accumulate && !d->relu && d->src_pitch % 8 == 0 &&
           d->dst_pitch % 8 == 0 && (!d->shuffle2x2 || (d->Cout >> 2) % 8 == 0) && d->src_split_c <= 0 && d->dst_split_c <= 0 &&
           !getenv("AAU_PW_OLD") && !getenv("AAU_PW_NBUF") && !getenv("AAU_NO_WIDE_STORE");
}

// 1 when aau_conv_igemm_bnin serves this descriptor: a 3x3 conv that the strip kernel takes, one dense source plane; or a
// 1x1 conv / ConvTranspose2d(2,2) forward (shuffle2x2) that the scalar-offset resident-weight kernel takes
extern "C" int aau_conv_bnin_ok(const aau_conv_desc* d) {
    if (!d || getenv("AAU_NO_BNIN")) return 0;
    if (bnin_1x1_ok(d)) return 1;
    if (getenv("AAU_NO_C3S")) return 0;
    return conv3x3_applicable(d) && (d->Cin == 48 || d->Cin == 96) && (d->Cout == 48 || d->Cout == 96) &&
           d->Cpad == (d->Cin == 48 ? 64 : 96) && !d->accumulate && !d->relu && d->src_pitch % 8 == 0 && d->dst_pitch % 8 == 0 &&
           d->src_split_c <= 0 && d->dst_split_c <= 0;
}

// dst = conv3x3(relu(src * in_scale + in_shift), w) (+ statistics of dst): aau_bn_act followed by aau_conv_igemm, bit for
// bit, without the activation ever leaving the chip (pipeline:59-65: the ReLU(BatchNorm(.)) of the PRODUCING block fused
// into the consuming convolution's operand path)
extern "C" int aau_conv_igemm_bnin(const aau_conv_desc* d, const aau_bf16* src, const float* in_scale, const float* in_shift,
                                   const aau_bf16* wpk, aau_bf16* dst, const float* bias, aau_stat* stats, int64_t stats_bytes,
                                   void* stream) {
    AAU_REQUIRE(d && src && in_scale && in_shift && wpk && dst, "aau_conv_igemm_bnin: null pointer");
    AAU_REQUIRE(aau_conv_bnin_ok(d) && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                "aau_conv_igemm_bnin: descriptor not served (aau_conv_bnin_ok) or misaligned operands");
    if (stats) AAU_CHECK_STAT("aau_conv_igemm_bnin", stats, stats_bytes, d->Cout);
    const int64_t M = (int64_t)d->N * d->H * d->W;
    const int64_t src_bytes = ((M - 1) * d->src_pitch + d->Cin) * 2;
    AAU_REQUIRE(src_bytes < 0x7fffffff && M * (d->shuffle2x2 ? 4 : 1) * d->dst_pitch * 2 < 0x7fffffff,
                "aau_conv_igemm_bnin: tensors must stay below 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    if (bnin_1x1_ok(d)) {      // 1x1 / ConvTranspose forward: the resident-weight kernel with the transform on its pixel tiles
        ProfScope prof1(0, 2.0 * M * (double)d->Cout * d->Cin, s);
        prof_tag(nullptr, 2.0 * ((double)M * d->Cin + (double)M * d->Cout + (double)d->Cout * d->Cin));
        return conv1x1_resw_launch(d, src, wpk, dst, bias, nullptr, nullptr, (float*)stats, (unsigned)src_bytes,
                                   (unsigned)((int64_t)d->Cout * d->Cpad * 2), s, in_scale, in_shift);
    }
    AAU_REQUIRE(bias == nullptr, "aau_conv_igemm_bnin: the 3x3 form takes no bias");
    ProfScope prof(0, 2.0 * M * (double)d->Cout * d->Cin * 9, s);
    prof_tag(nullptr, 2.0 * ((double)M * d->Cin + (double)M * d->Cout + (double)d->Cout * 9 * d->Cin));
    C3Args a;
    a.d = *d;
    a.src = src; a.wpk = wpk; a.dst = dst; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.stats = (float*)stats;
    a.rev = next_traversal();
    a.nchunk = d->Cpad / 32;
    a.src_bytes = (unsigned)src_bytes;
    a.wpk_bytes = (unsigned)((int64_t)d->Cout * 9 * d->Cpad * 2);
    a.tiles_x = d->W / 16; a.tiles_y = d->H / 16;
    a.nowide = 0; a.nopair = 0;
    a.in_scale = in_scale; a.in_shift = in_shift;
    return conv3x3s_launch(a, s);
}

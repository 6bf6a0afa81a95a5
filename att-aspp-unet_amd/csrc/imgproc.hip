// Byte / integer image work around the network on gfx950 (HBM-bound; no MFMA here):
//   * aau_seg_counts: the integer counts behind eval_segmentation_batch.py:41-49 Dice / IoU.
// (The GPU-resident inference tail and input pipeline of SURVEY.md section 8 rows f1 / f2 / f4 live here too.)
#include "common.h"

// The fp32 / fp64 filters below restate published algorithms operation by operation (separately rounded multiplies and
// adds): no fused multiply-add contraction anywhere in this file (HIP's __fmul_rn / __fadd_rn are plain operators that
// hipcc would still contract under its default -ffp-contract=fast): build.py compiles this file with -ffp-contract=off.

namespace aau {

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void seg_counts_kernel(const TA* __restrict__ a, const TB* __restrict__ b, int64_t n,
                                                         unsigned long long* __restrict__ out) {
    unsigned na = 0, nb = 0, ni = 0;   // per-thread counts stay below 2^32 (grid-stride over < 2^40 elements)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool x = a[i] > (TA)0, y = b[i] > (TB)0;
        na += x; nb += y; ni += (x && y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        na += __shfl_xor(na, o, 64); nb += __shfl_xor(nb, o, 64); ni += __shfl_xor(ni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (na) atomicAdd(out + 0, (unsigned long long)na);
        if (nb) atomicAdd(out + 1, (unsigned long long)nb);
        if (ni) atomicAdd(out + 2, (unsigned long long)ni);
    }
}

__global__ void zero_u64_kernel(unsigned long long* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0ull;
}

}  // namespace aau

using namespace aau;

extern "C" int aau_seg_counts(const void* a, int a_is_f32, const void* b, int b_is_f32, int64_t n, uint64_t* out3,
                              void* stream) {
    AAU_REQUIRE(a && b && out3 && n > 0, "aau_seg_counts: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(3, 0, s);
    unsigned long long* o = (unsigned long long*)out3;
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(64), 0, s, o, 3);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 g((unsigned)blocks), t(256);
    if (a_is_f32 && b_is_f32) hipLaunchKernelGGL((seg_counts_kernel<float, float>), g, t, 0, s, (const float*)a, (const float*)b, n, o);
    else if (a_is_f32) hipLaunchKernelGGL((seg_counts_kernel<float, unsigned char>), g, t, 0, s, (const float*)a, (const unsigned char*)b, n, o);
    else if (b_is_f32) hipLaunchKernelGGL((seg_counts_kernel<unsigned char, float>), g, t, 0, s, (const unsigned char*)a, (const float*)b, n, o);
    else hipLaunchKernelGGL((seg_counts_kernel<unsigned char, unsigned char>), g, t, 0, s, (const unsigned char*)a, (const unsigned char*)b, n, o);
    return check_launch("aau_seg_counts");
}

// =====================================================================================================================
// GPU-resident inference tail and input pipeline: byte and index work around
// the network, restated from the published algorithms of the libraries the reference calls (cv2 / skimage / scipy are
// not importable here: their outputs are "parity unpinned", the CPU restatement in oracle/imgproc_ref.py is the checker).
//   pipeline:449-457 (predict): normalize -> CLAHE(1.0, 8x8) -> medianBlur 3 -> Resize 512 -> ToFloat | forward + TTA |
//                               resize back -> GaussianBlur 5x5 -> threshold -> refine_mask (:340-348)
//   model_attention_aspp.py:20-89: ROI-224 crop around the bright centroid, paste back, 3x3 dilation + largest component
// One thread per pixel everywhere; images are small (0.2-0.5 Mpixel), the point is that nothing leaves HBM between the
// decoder and the final mask.
// =====================================================================================================================
namespace aau {

__device__ __forceinline__ int reflect101(int i, int n) {      // cv2 BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }      // cvRound: nearest, ties to even
__device__ __forceinline__ unsigned char sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- cv2.resize(..., INTER_LINEAR) on fp32 (resize-back of the probability map, pipeline:455) ----
// source coordinate of destination index d: f = (d + 0.5) * scale - 0.5, s = floor(f), clamped as resize.cpp does
// (resize.cpp: scale = 1 / ((double)ndst / nsrc), fx = (float)((dx + 0.5) * scale - 0.5), in double, no contraction)
__device__ __forceinline__ void lin_coord(int d, double scale, int nsrc, int& s, float& f) {
    f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= nsrc - 1) { f = 0.f; s = nsrc - 1; }
}
__global__ void resize_lin_f32_kernel(const float* src, float* dst, int Hs, int Ws, int Hd, int Wd, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Hd * Wd) return;
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), n = (int)(i / ((int64_t)Wd * Hd));
    int sx, sy; float fx, fy;
    lin_coord(x, 1.0 / ((double)Wd / (double)Ws), Ws, sx, fx);
    lin_coord(y, 1.0 / ((double)Hd / (double)Hs), Hs, sy, fy);
    const int sx1 = min(sx + 1, Ws - 1), sy1 = min(sy + 1, Hs - 1);
    const float* S = src + (int64_t)n * Hs * Ws;
    // horizontal pass of both rows, then the vertical blend -- the order (and the separate roundings) of resize.cpp
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float r0 = __fadd_rn(__fmul_rn(S[(int64_t)sy * Ws + sx], a0), __fmul_rn(S[(int64_t)sy * Ws + sx1], a1));
    const float r1 = __fadd_rn(__fmul_rn(S[(int64_t)sy1 * Ws + sx], a0), __fmul_rn(S[(int64_t)sy1 * Ws + sx1], a1));
    dst[i] = __fadd_rn(__fmul_rn(r0, b0), __fmul_rn(r1, b1));
}

// ---- cv2.resize on uint8: 11-bit fixed-point coefficients, the 8-bit vertical pass of resize.cpp ----
__global__ void resize_lin_u8_kernel(const unsigned char* src, unsigned char* dst, int Hs, int Ws, int Hd, int Wd, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Hd * Wd) return;
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), n = (int)(i / ((int64_t)Wd * Hd));
    int sx, sy; float fx, fy;
    lin_coord(x, 1.0 / ((double)Wd / (double)Ws), Ws, sx, fx);
    lin_coord(y, 1.0 / ((double)Hd / (double)Hs), Hs, sy, fy);
    const int sx1 = min(sx + 1, Ws - 1), sy1 = min(sy + 1, Hs - 1);
    const int a0 = cv_round((1.f - fx) * 2048.f), a1 = cv_round(fx * 2048.f);
    const int b0 = cv_round((1.f - fy) * 2048.f), b1 = cv_round(fy * 2048.f);
    const unsigned char* S = src + (int64_t)n * Hs * Ws;
    const int r0 = S[(int64_t)sy * Ws + sx] * a0 + S[(int64_t)sy * Ws + sx1] * a1;
    const int r1 = S[(int64_t)sy1 * Ws + sx] * a0 + S[(int64_t)sy1 * Ws + sx1] * a1;
    dst[i] = (unsigned char)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// ---- cv2.GaussianBlur(prob, (5,5), 0): fixed kernel [1 4 6 4 1]/16, separable, BORDER_REFLECT_101 ----
__global__ void gauss5_f32_kernel(const float* src, float* dst, int H, int W, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const float* S = src + (int64_t)n * H * W;
    const float k0 = 0.375f, k1 = 0.25f, k2 = 0.0625f;
    int xs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) xs[t] = reflect101(x + t - 2, W);
    float row[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const float* R = S + (int64_t)reflect101(y + r - 2, H) * W;
        // symmetric row filter: centre, then the mirrored pairs
        float s = __fmul_rn(R[xs[2]], k0);
        s = __fadd_rn(s, __fmul_rn(__fadd_rn(R[xs[1]], R[xs[3]]), k1));
        s = __fadd_rn(s, __fmul_rn(__fadd_rn(R[xs[0]], R[xs[4]]), k2));
        row[r] = s;
    }
    float s = __fmul_rn(row[2], k0);
    s = __fadd_rn(s, __fmul_rn(__fadd_rn(row[1], row[3]), k1));
    s = __fadd_rn(s, __fmul_rn(__fadd_rn(row[0], row[4]), k2));
    dst[i] = s;
}

__global__ void threshold_u8_kernel(const float* src, float thr, unsigned char* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] > thr ? 1 : 0;
}

// ---- connected components: lock-free union-find on the pixel grid (labels = smallest linear index of the component) ----
__device__ __forceinline__ int uf_find(int* L, int i) {
    int r = i;
    while (true) {
        const int p = __hip_atomic_load(&L[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == r) break;
        r = p;
    }
    return r;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }       // a < b: hang b's root under a
        const int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;                                              // someone re-rooted b meanwhile: merge that root too
    }
}
// fg = 1: label the non-zero pixels, fg = 0: label the zero pixels (background components, for hole filling)
__global__ void cc_init_kernel(const unsigned char* mask, int* L, int64_t n, int HW, int fg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool on = (mask[i] != 0) == (fg != 0);
    L[i] = on ? (int)(i % HW) : -1;
}
__global__ void cc_merge_kernel(int* Lall, int H, int W, int N, int conn8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int HW = H * W;
    const int n = (int)(i / HW), p = (int)(i % HW);
    int* L = Lall + (int64_t)n * HW;
    if (L[p] < 0) return;
    const int x = p % W, y = p / W;
    // backward neighbours only: every adjacency is seen from its later pixel
    if (x > 0 && L[p - 1] >= 0) uf_union(L, p, p - 1);
    if (y > 0) {
        if (L[p - W] >= 0) uf_union(L, p, p - W);
        if (conn8) {
            if (x > 0 && L[p - W - 1] >= 0) uf_union(L, p, p - W - 1);
            if (x < W - 1 && L[p - W + 1] >= 0) uf_union(L, p, p - W + 1);
        }
    }
}
__global__ void cc_flatten_kernel(int* Lall, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int* L = Lall + (i / HW) * HW;
    const int p = (int)(i % HW);
    if (L[p] >= 0) L[p] = uf_find(L, p);
}
__global__ void cc_count_kernel(const int* L, int* sizes, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    if (r >= 0) atomicAdd(sizes + (i / HW) * HW + r, 1);
}
// best[n] = max over roots of (size << 32 | ~root): largest component, the earliest one in raster order on ties
// (skimage / scipy number components in raster order of their first pixel, and argmax keeps the first maximum)
__global__ void cc_best_kernel(const int* L, const int* sizes, unsigned long long* best, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = (int)(i % HW);
    if (L[i] == p) {
        const unsigned long long key = ((unsigned long long)(unsigned)sizes[i] << 32) | (0xffffffffu - (unsigned)p);
        atomicMax(best + i / HW, key);
    }
}
__global__ void cc_keep_best_kernel(const int* L, const unsigned long long* best, unsigned char* out, int HW, int64_t n,
                                    int min_area) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = best[i / HW];
    const int size = (int)(b >> 32), root = (int)(0xffffffffu - (unsigned)(b & 0xffffffffu));
    out[i] = (size > 0 && size >= min_area && L[i] == root) ? 1 : 0;
}
// hole filling (scipy.ndimage.binary_fill_holes, 4-connected background): a background component that touches the
// frame border is outside; every other zero pixel becomes foreground
__global__ void holes_border_kernel(const int* L, int* flag, int H, int W, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 2 * (H + W);
    if (i >= (int64_t)N * per) return;
    const int n = (int)(i / per), k = (int)(i % per);
    int x, y;
    if (k < W) { x = k; y = 0; }
    else if (k < 2 * W) { x = k - W; y = H - 1; }
    else if (k < 2 * W + H) { x = 0; y = k - 2 * W; }
    else { x = W - 1; y = k - 2 * W - H; }
    const int64_t base = (int64_t)n * H * W;
    const int r = L[base + (int64_t)y * W + x];
    if (r >= 0) flag[base + r] = 1;
}
__global__ void holes_fill_kernel(const unsigned char* mask, const int* L, const int* flag, unsigned char* out, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    out[i] = (mask[i] != 0 || (r >= 0 && flag[(i / HW) * HW + r] == 0)) ? 1 : 0;
}

// ---- binary morphology, pixels outside the frame ignored (cv2 default border of dilate / erode; scipy border 0) ----
// shape 7: cv2.getStructuringElement(MORPH_ELLIPSE, (7,7)) -- row half-widths 0 2 3 3 3 2 0;  shape 3: full 3x3
__global__ void morph_kernel(const unsigned char* src, unsigned char* dst, int H, int W, int N, int shape, int erode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const unsigned char* S = src + (i / ((int64_t)W * H)) * ((int64_t)W * H);
    const int R = shape == 7 ? 3 : 1;
    bool any = false, all = true;
    for (int dy = -R; dy <= R; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        int hw = R;
        if (shape == 7) { const int a = dy < 0 ? -dy : dy; hw = a == 3 ? 0 : (a == 2 ? 2 : 3); }
        for (int dx = -hw; dx <= hw; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const bool v = S[(int64_t)yy * W + xx] != 0;
            any |= v;
            all &= v;
        }
    }
    dst[i] = (erode ? all : any) ? 1 : 0;
}

// ---- input pipeline: cv2.normalize(NORM_MINMAX) -> CLAHE(1.0, 8x8) -> medianBlur(3) -> resize -> ToFloat ----
__global__ void minmax_u8_kernel(const unsigned char* src, int* mm, int HW, int64_t n) {
    int lo = 255, hi = 0;
    const int frame = blockIdx.y;
    const unsigned char* S = src + (int64_t)frame * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const int v = S[i];
        lo = min(lo, v); hi = max(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(mm + 2 * frame, lo); atomicMax(mm + 2 * frame + 1, hi); }
}
__global__ void mm_init_kernel(int* mm, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { mm[2 * i] = 255; mm[2 * i + 1] = 0; }
}
// dst = saturate(round(src * scale + shift)), scale = 255 / (max - min), shift = -min * scale (convertScaleAbs-free form)
__global__ void normalize_u8_kernel(const unsigned char* src, unsigned char* dst, const int* mm, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i / HW);
    const int lo = mm[2 * f], hi = mm[2 * f + 1];
    const double scale = hi > lo ? 255.0 / (double)(hi - lo) : 0.0;
    const double shift = -(double)lo * scale;
    dst[i] = sat_u8(__double2int_rn(__dadd_rn(__dmul_rn((double)src[i], scale), shift)));
}

// CLAHE LUTs: one workgroup per (tile, frame).  When either axis does not divide by the grid, clahe.cpp extends BOTH
// on the right / bottom by tiles - size % tiles with BORDER_REFLECT_101 (tw, th come from the extended size).
__global__ __launch_bounds__(256) void clahe_lut_kernel(const unsigned char* src, unsigned char* lut, int H, int W, int tw,
                                                        int th, int tiles, float clip_limit) {
    __shared__ int hist[256];
    __shared__ int s_clipped;
    const int tx = blockIdx.x % tiles, ty = blockIdx.x / tiles, frame = blockIdx.y;
    const unsigned char* S = src + (int64_t)frame * H * W;
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_clipped = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int y = reflect101(ty * th + i / tw, H), x = reflect101(tx * tw + i % tw, W);
        atomicAdd(&hist[S[(int64_t)y * W + x]], 1);
    }
    __syncthreads();
    const int area = tw * th;
    int climit = 0;
    if (clip_limit > 0.f) {
        climit = (int)(clip_limit * (float)area / 256.f);
        if (climit < 1) climit = 1;
        const int h = hist[threadIdx.x];
        if (h > climit) { atomicAdd(&s_clipped, h - climit); hist[threadIdx.x] = climit; }
        __syncthreads();
        const int clipped = s_clipped;
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        hist[threadIdx.x] += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            // for (i = 0; i < 256 && residual > 0; i += step, --residual) ++hist[i]
            const int i = threadIdx.x;
            if (i % step == 0 && i / step < residual) hist[i] += 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float scale = 255.f / (float)area;
        int sum = 0;
        unsigned char* out = lut + ((int64_t)frame * tiles * tiles + ty * tiles + tx) * 256;
        for (int i = 0; i < 256; ++i) {
            sum += hist[i];
            out[i] = sat_u8(cv_round((float)sum * scale));
        }
    }
}
__global__ void clahe_apply_kernel(const unsigned char* src, const unsigned char* lut, unsigned char* dst, int H, int W, int tw,
                                   int th, int tiles, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const float inv_tw = 1.f / (float)tw, inv_th = 1.f / (float)th;
    const float tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f), txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f);
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    int ty2 = ty1 + 1, tx2 = tx1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.f - ya, xa = txf - (float)tx1, xa1 = 1.f - xa;
    ty1 = max(ty1, 0); ty2 = min(ty2, tiles - 1); tx1 = max(tx1, 0); tx2 = min(tx2, tiles - 1);
    const int v = src[i];
    const unsigned char* L = lut + (int64_t)n * tiles * tiles * 256;
    const float l11 = L[(ty1 * tiles + tx1) * 256 + v], l12 = L[(ty1 * tiles + tx2) * 256 + v];
    const float l21 = L[(ty2 * tiles + tx1) * 256 + v], l22 = L[(ty2 * tiles + tx2) * 256 + v];
    const float res = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), ya1),
                                __fmul_rn(__fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa)), ya));
    dst[i] = sat_u8(cv_round(res));
}
// cv2.medianBlur(img, 3): BORDER_REPLICATE
__global__ void median3_u8_kernel(const unsigned char* src, unsigned char* dst, int H, int W, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const unsigned char* S = src + (i / ((int64_t)W * H)) * ((int64_t)W * H);
    int v[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
            v[(dy + 1) * 3 + dx + 1] = S[(int64_t)min(max(y + dy, 0), H - 1) * W + min(max(x + dx, 0), W - 1)];
#define AAU_SW(a, b) { const int lo_ = min(v[a], v[b]), hi_ = max(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
    AAU_SW(1, 2) AAU_SW(4, 5) AAU_SW(7, 8) AAU_SW(0, 1) AAU_SW(3, 4) AAU_SW(6, 7) AAU_SW(1, 2) AAU_SW(4, 5) AAU_SW(7, 8)
    AAU_SW(0, 3) AAU_SW(5, 8) AAU_SW(4, 7) AAU_SW(3, 6) AAU_SW(1, 4) AAU_SW(2, 5) AAU_SW(4, 7) AAU_SW(4, 2) AAU_SW(6, 4)
    AAU_SW(4, 2)
#undef AAU_SW
    dst[i] = (unsigned char)v[4];
}
// albumentations ToFloat: img.astype(float32) / max_value (a division, not a multiplication by the reciprocal)
__global__ void u8_to_f32_kernel(const unsigned char* src, float* dst, float max_value, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i] / max_value;
}

// ---- ROI-224 crop of model_attention_aspp.py:20-31: centre = mean position of the pixels brighter than 1.2 x mean ----
// sums per frame: [0] sum of values (fp64), then (after the threshold is known) [1] count, [2] sum x, [3] sum y
__global__ void roi_sum_kernel(const float* img, double* sums, int HW) {
    const int frame = blockIdx.y;
    const float* S = img + (int64_t)frame * HW;
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) acc += (double)S[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sums + 4 * frame, acc);
}
__global__ void roi_centroid_kernel(const float* img, double* sums, int H, int W) {
    const int frame = blockIdx.y, HW = H * W;
    const float* S = img + (int64_t)frame * HW;
    // numpy: thr = img.mean() * 1.2 on a float32 array -> float32 mean (pairwise sum; the fp64 sum rounded to fp32 here)
    const float thr = (float)(sums[4 * frame] / (double)HW) * 1.2f;
    unsigned long long cnt = 0, sx = 0, sy = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x)
        if (S[i] > thr) { ++cnt; sx += i % W; sy += i / W; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
    if ((threadIdx.x & 63) == 0 && cnt) {
        unsigned long long* u = (unsigned long long*)sums;
        atomicAdd(u + 4 * frame + 1, cnt); atomicAdd(u + 4 * frame + 2, sx); atomicAdd(u + 4 * frame + 3, sy);
    }
}
// origin (x0, y0) of the 224 x 224 window per frame (int(xs.mean()), int(ys.mean()); the frame centre when empty)
__global__ void roi_origin_kernel(const double* sums, int* org, int H, int W, int R, int N) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= N) return;
    const unsigned long long* u = (const unsigned long long*)sums;
    const unsigned long long cnt = u[4 * f + 1];
    int cx = W / 2, cy = H / 2;
    if (cnt) { cx = (int)((double)u[4 * f + 2] / (double)cnt); cy = (int)((double)u[4 * f + 3] / (double)cnt); }
    int x0 = max(0, cx - R / 2), y0 = max(0, cy - R / 2);
    x0 = min(x0, W - R); y0 = min(y0, H - R);
    org[2 * f] = x0; org[2 * f + 1] = y0;
}
// dst[f] = src[f][y0 : y0+R, x0 : x0+R] (zero where the window leaves the frame: frames smaller than R)
__global__ void roi_crop_kernel(const float* src, const int* org, float* dst, int H, int W, int R, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * R * R) return;
    const int x = (int)(i % R), y = (int)((i / R) % R), f = (int)(i / ((int64_t)R * R));
    const int xs = org[2 * f] + x, ys = org[2 * f + 1] + y;
    dst[i] = ((unsigned)xs < (unsigned)W && (unsigned)ys < (unsigned)H && xs >= 0 && ys >= 0) ? src[((int64_t)f * H + ys) * W + xs] : 0.f;
}
// full[f][y0 + y][x0 + x] = sigmoid(logits[f][y][x]) inside the window, 0 elsewhere (full is written completely)
__global__ void roi_paste_sigmoid_kernel(const float* logits, const int* org, float* full, int H, int W, int R, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), f = (int)(i / ((int64_t)W * H));
    const int rx = x - org[2 * f], ry = y - org[2 * f + 1];
    float v = 0.f;
    if ((unsigned)rx < (unsigned)R && (unsigned)ry < (unsigned)R) v = 1.f / (1.f + expf(-logits[((int64_t)f * R + ry) * R + rx]));
    full[i] = v;
}
// areas[f] = number of pixels above thr
__global__ void frame_area_kernel(const float* prob, float thr, int* areas, int HW) {
    const int frame = blockIdx.y;
    const float* S = prob + (int64_t)frame * HW;
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) c += S[i] > thr ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(areas + frame, c);
}
__global__ void zero_i32_kernel(int* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace aau

#define IMG_GRID(n) dim3((unsigned)(((int64_t)(n) + 255) / 256)), dim3(256)
#define IMG_CHECK_DIMS(name, N, H, W) \
    AAU_REQUIRE((N) > 0 && (H) > 0 && (W) > 0 && (int64_t)(N) * (H) * (W) < 0x7fffffff, name ": bad shape %d x %d x %d", N, H, W)

extern "C" int aau_resize_bilinear_f32(const float* src, int Hs, int Ws, float* dst, int Hd, int Wd, int N, void* stream) {
    AAU_REQUIRE(src && dst, "aau_resize_bilinear_f32: null pointer");
    IMG_CHECK_DIMS("aau_resize_bilinear_f32", N, Hs, Ws);
    IMG_CHECK_DIMS("aau_resize_bilinear_f32", N, Hd, Wd);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(resize_lin_f32_kernel, IMG_GRID((int64_t)N * Hd * Wd), 0, (hipStream_t)stream, src, dst, Hs, Ws, Hd, Wd, N);
    return check_launch("aau_resize_bilinear_f32");
}
extern "C" int aau_resize_bilinear_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd, int N, void* stream) {
    AAU_REQUIRE(src && dst, "aau_resize_bilinear_u8: null pointer");
    IMG_CHECK_DIMS("aau_resize_bilinear_u8", N, Hs, Ws);
    IMG_CHECK_DIMS("aau_resize_bilinear_u8", N, Hd, Wd);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(resize_lin_u8_kernel, IMG_GRID((int64_t)N * Hd * Wd), 0, (hipStream_t)stream, src, dst, Hs, Ws, Hd, Wd, N);
    return check_launch("aau_resize_bilinear_u8");
}
extern "C" int aau_gauss5_f32(const float* src, float* dst, int N, int H, int W, void* stream) {
    AAU_REQUIRE(src && dst && src != dst, "aau_gauss5_f32: null / aliased pointers");
    IMG_CHECK_DIMS("aau_gauss5_f32", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(gauss5_f32_kernel, IMG_GRID((int64_t)N * H * W), 0, (hipStream_t)stream, src, dst, H, W, N);
    return check_launch("aau_gauss5_f32");
}
extern "C" int aau_threshold_u8(const float* src, float thr, uint8_t* dst, int64_t n, void* stream) {
    AAU_REQUIRE(src && dst && n > 0, "aau_threshold_u8: bad args");
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(threshold_u8_kernel, IMG_GRID(n), 0, (hipStream_t)stream, src, thr, dst, n);
    return check_launch("aau_threshold_u8");
}

static void cc_label(const uint8_t* mask, int* labels, int N, int H, int W, int conn8, int fg, hipStream_t s) {
    const int64_t n = (int64_t)N * H * W;
    hipLaunchKernelGGL(cc_init_kernel, IMG_GRID(n), 0, s, mask, labels, n, H * W, fg);
    hipLaunchKernelGGL(cc_merge_kernel, IMG_GRID(n), 0, s, labels, H, W, N, conn8);
    hipLaunchKernelGGL(cc_flatten_kernel, IMG_GRID(n), 0, s, labels, H * W, n);
}

extern "C" int aau_cc_label(const uint8_t* mask, int32_t* labels, int N, int H, int W, int conn8, void* stream) {
    AAU_REQUIRE(mask && labels, "aau_cc_label: null pointer");
    IMG_CHECK_DIMS("aau_cc_label", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    cc_label(mask, labels, N, H, W, conn8, 1, (hipStream_t)stream);
    return check_launch("aau_cc_label");
}
extern "C" int aau_cc_keep_largest(const uint8_t* mask, uint8_t* out, int32_t* labels_ws, int32_t* sizes_ws, uint64_t* best_ws,
                                   int N, int H, int W, int conn8, int min_area, void* stream) {
    AAU_REQUIRE(mask && out && labels_ws && sizes_ws && best_ws, "aau_cc_keep_largest: null pointer");
    IMG_CHECK_DIMS("aau_cc_keep_largest", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    const int64_t n = (int64_t)N * H * W;
    cc_label(mask, labels_ws, N, H, W, conn8, 1, s);
    hipLaunchKernelGGL(zero_i32_kernel, IMG_GRID(n), 0, s, sizes_ws, n);
    hipLaunchKernelGGL(zero_i32_kernel, IMG_GRID(2 * N), 0, s, (int*)best_ws, (int64_t)2 * N);
    hipLaunchKernelGGL(cc_count_kernel, IMG_GRID(n), 0, s, labels_ws, sizes_ws, H * W, n);
    hipLaunchKernelGGL(cc_best_kernel, IMG_GRID(n), 0, s, labels_ws, sizes_ws, (unsigned long long*)best_ws, H * W, n);
    hipLaunchKernelGGL(cc_keep_best_kernel, IMG_GRID(n), 0, s, labels_ws, (const unsigned long long*)best_ws, out, H * W, n, min_area);
    return check_launch("aau_cc_keep_largest");
}
extern "C" int aau_fill_holes(const uint8_t* mask, uint8_t* out, int32_t* labels_ws, int32_t* flag_ws, int N, int H, int W,
                              void* stream) {
    AAU_REQUIRE(mask && out && labels_ws && flag_ws, "aau_fill_holes: null pointer");
    IMG_CHECK_DIMS("aau_fill_holes", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    const int64_t n = (int64_t)N * H * W;
    cc_label(mask, labels_ws, N, H, W, 0, 0, s);
    hipLaunchKernelGGL(zero_i32_kernel, IMG_GRID(n), 0, s, flag_ws, n);
    hipLaunchKernelGGL(holes_border_kernel, IMG_GRID((int64_t)N * 2 * (H + W)), 0, s, labels_ws, flag_ws, H, W, N);
    hipLaunchKernelGGL(holes_fill_kernel, IMG_GRID(n), 0, s, mask, labels_ws, flag_ws, out, H * W, n);
    return check_launch("aau_fill_holes");
}
extern "C" int aau_morph(const uint8_t* src, uint8_t* dst, int N, int H, int W, int shape, int erode, void* stream) {
    AAU_REQUIRE(src && dst && src != dst, "aau_morph: null / aliased pointers");
    AAU_REQUIRE(shape == 3 || shape == 7, "aau_morph: shape %d (3 = full 3x3, 7 = 7x7 ellipse)", shape);
    IMG_CHECK_DIMS("aau_morph", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(morph_kernel, IMG_GRID((int64_t)N * H * W), 0, (hipStream_t)stream, src, dst, H, W, N, shape, erode);
    return check_launch("aau_morph");
}
extern "C" int aau_normalize_minmax_u8(const uint8_t* src, uint8_t* dst, int32_t* mm_ws, int N, int H, int W, void* stream) {
    AAU_REQUIRE(src && dst && mm_ws, "aau_normalize_minmax_u8: null pointer");
    IMG_CHECK_DIMS("aau_normalize_minmax_u8", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(mm_init_kernel, IMG_GRID(N), 0, s, mm_ws, N);
    hipLaunchKernelGGL(minmax_u8_kernel, dim3(64, N), dim3(256), 0, s, src, mm_ws, H * W, (int64_t)N * H * W);
    hipLaunchKernelGGL(normalize_u8_kernel, IMG_GRID((int64_t)N * H * W), 0, s, src, dst, mm_ws, H * W, (int64_t)N * H * W);
    return check_launch("aau_normalize_minmax_u8");
}
extern "C" int aau_clahe_u8(const uint8_t* src, uint8_t* dst, uint8_t* lut_ws, int N, int H, int W, float clip_limit, int tiles,
                            void* stream) {
    AAU_REQUIRE(src && dst && lut_ws && tiles >= 1 && tiles <= 64, "aau_clahe_u8: bad args");
    IMG_CHECK_DIMS("aau_clahe_u8", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    // clahe.cpp: if EITHER axis does not divide, BOTH grow by tiles - size % tiles (a divisible axis by a whole `tiles`)
    const bool ext = (W % tiles) || (H % tiles);
    AAU_REQUIRE(!ext || (W > tiles && H > tiles), "aau_clahe_u8: frame smaller than the tile grid");
    const int Wp = ext ? W + tiles - W % tiles : W, Hp = ext ? H + tiles - H % tiles : H;
    const int tw = Wp / tiles, th = Hp / tiles;
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles * tiles, N), dim3(256), 0, s, src, lut_ws, H, W, tw, th, tiles, clip_limit);
    hipLaunchKernelGGL(clahe_apply_kernel, IMG_GRID((int64_t)N * H * W), 0, s, src, lut_ws, dst, H, W, tw, th, tiles, N);
    return check_launch("aau_clahe_u8");
}
extern "C" int aau_median3_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, void* stream) {
    AAU_REQUIRE(src && dst && src != dst, "aau_median3_u8: null / aliased pointers");
    IMG_CHECK_DIMS("aau_median3_u8", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(median3_u8_kernel, IMG_GRID((int64_t)N * H * W), 0, (hipStream_t)stream, src, dst, H, W, N);
    return check_launch("aau_median3_u8");
}
extern "C" int aau_u8_to_f32(const uint8_t* src, float* dst, float max_value, int64_t n, void* stream) {
    AAU_REQUIRE(src && dst && n > 0 && max_value > 0.f, "aau_u8_to_f32: bad args");
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(u8_to_f32_kernel, IMG_GRID(n), 0, (hipStream_t)stream, src, dst, max_value, n);
    return check_launch("aau_u8_to_f32");
}
extern "C" int aau_roi_origin(const float* img, double* sums_ws, int32_t* origin, int N, int H, int W, int R, void* stream) {
    AAU_REQUIRE(img && sums_ws && origin && R > 0, "aau_roi_origin: bad args");
    IMG_CHECK_DIMS("aau_roi_origin", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(zero_i32_kernel, IMG_GRID(8 * N), 0, s, (int*)sums_ws, (int64_t)8 * N);
    hipLaunchKernelGGL(roi_sum_kernel, dim3(64, N), dim3(256), 0, s, img, sums_ws, H * W);
    hipLaunchKernelGGL(roi_centroid_kernel, dim3(64, N), dim3(256), 0, s, img, sums_ws, H, W);
    hipLaunchKernelGGL(roi_origin_kernel, IMG_GRID(N), 0, s, sums_ws, origin, H, W, R, N);
    return check_launch("aau_roi_origin");
}
extern "C" int aau_roi_crop(const float* src, const int32_t* origin, float* dst, int N, int H, int W, int R, void* stream) {
    AAU_REQUIRE(src && origin && dst && R > 0, "aau_roi_crop: bad args");
    IMG_CHECK_DIMS("aau_roi_crop", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(roi_crop_kernel, IMG_GRID((int64_t)N * R * R), 0, (hipStream_t)stream, src, origin, dst, H, W, R, N);
    return check_launch("aau_roi_crop");
}
extern "C" int aau_roi_paste_sigmoid(const float* logits, const int32_t* origin, float* full, int N, int H, int W, int R,
                                     void* stream) {
    AAU_REQUIRE(logits && origin && full && R > 0, "aau_roi_paste_sigmoid: bad args");
    IMG_CHECK_DIMS("aau_roi_paste_sigmoid", N, H, W);
    ProfScope prof(2, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(roi_paste_sigmoid_kernel, IMG_GRID((int64_t)N * H * W), 0, (hipStream_t)stream, logits, origin, full, H, W, R, N);
    return check_launch("aau_roi_paste_sigmoid");
}
extern "C" int aau_frame_areas(const float* prob, float thr, int32_t* areas, int N, int H, int W, void* stream) {
    AAU_REQUIRE(prob && areas, "aau_frame_areas: null pointer");
    IMG_CHECK_DIMS("aau_frame_areas", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(zero_i32_kernel, IMG_GRID(N), 0, s, areas, (int64_t)N);
    hipLaunchKernelGGL(frame_area_kernel, dim3(64, N), dim3(256), 0, s, prob, thr, areas, H * W);
    return check_launch("aau_frame_areas");
}

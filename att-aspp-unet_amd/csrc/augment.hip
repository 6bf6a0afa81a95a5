// Random training augmentations of FetalACDataset (attention_aspp_unet_pipeline_stage.py:149-153) on gfx950, batched
// over frames that are already resident in HBM as uint8 [N][H][W]:
//   Affine(scale, rotate, translate_percent)         -> aau_warp_affine_u8   (cv2.warpAffine, BORDER_CONSTANT 0)
//   RandomGamma + RandomBrightnessContrast           -> aau_lut_u8           (cv2.LUT; the per-frame tables come from the host)
//   ElasticTransform(alpha, sigma)                   -> aau_elastic_noise / aau_gauss_sep_f32 / aau_remap_u8
//                                                       (noise -> GaussianBlur -> cv2.remap, BORDER_REFLECT_101)
// Every kernel takes PER-FRAME parameters (a frame whose transform was not drawn gets the identity), so one launch
// serves a whole batch.  HBM-bound byte work: one thread per pixel, coalesced along x.
// albumentations / cv2 are not importable in the build container: these follow the published algorithms with exact
// (not cv2's 1/32-pixel fixed-point) bilinear weights; the checker is oracle/augment_ref.py -- PARITY UNPINNED against
// the libraries themselves.  Compiled with -ffp-contract=off like imgproc.hip (separately rounded multiplies and adds).
#include "common.h"

namespace aau {

__device__ __forceinline__ int aug_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
__device__ __forceinline__ unsigned char aug_sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// dst(x, y) = src(m0 x + m1 y + m2, m3 x + m4 y + m5): `mats` holds the INVERSE map of each frame (dst -> src), fp64.
// nearest = 0: bilinear, pixels outside the frame count as `border` (constant);  nearest = 1: floor(c + 0.5).
__global__ void warp_affine_u8_kernel(const unsigned char* src, unsigned char* dst, const double* mats, int N, int H, int W,
                                      int nearest, int border) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const double* m = mats + 6 * n;
    const double fx = __dadd_rn(__dadd_rn(__dmul_rn(m[0], (double)x), __dmul_rn(m[1], (double)y)), m[2]);
    const double fy = __dadd_rn(__dadd_rn(__dmul_rn(m[3], (double)x), __dmul_rn(m[4], (double)y)), m[5]);
    const unsigned char* S = src + (int64_t)n * H * W;
    auto at = [&](int xx, int yy) -> float {
        return ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? (float)S[(int64_t)yy * W + xx] : (float)border;
    };
    if (nearest) {
        const int xs = (int)floor(fx + 0.5), ys = (int)floor(fy + 0.5);
        dst[i] = (unsigned char)at(xs, ys);
        return;
    }
    const double x0d = floor(fx), y0d = floor(fy);
    // far outside: every tap is border (also keeps the int conversion in range)
    if (x0d < -2.0 || y0d < -2.0 || x0d > (double)W + 1.0 || y0d > (double)H + 1.0) { dst[i] = (unsigned char)border; return; }
    const int x0 = (int)x0d, y0 = (int)y0d;
    const float a = (float)(fx - x0d), b = (float)(fy - y0d);
    const float r0 = __fadd_rn(__fmul_rn(at(x0, y0), 1.f - a), __fmul_rn(at(x0 + 1, y0), a));
    const float r1 = __fadd_rn(__fmul_rn(at(x0, y0 + 1), 1.f - a), __fmul_rn(at(x0 + 1, y0 + 1), a));
    dst[i] = aug_sat_u8(__float2int_rn(__fadd_rn(__fmul_rn(r0, 1.f - b), __fmul_rn(r1, b))));
}

// dst = lut[n][src]
__global__ void lut_u8_kernel(const unsigned char* src, unsigned char* dst, const unsigned char* luts, int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = luts[(i / HW) * 256 + src[i]];
}

// counter-based uniform noise in [-1, 1): plane 0 (dx) and plane 1 (dy) of frame n are keyed by (seeds[n], plane, pixel)
__global__ void elastic_noise_kernel(const uint64_t* seeds, float* out, int HW, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * 2 * HW) return;
    const int n = (int)(i / (2 * (int64_t)HW));
    const uint64_t k = (uint64_t)(i - (int64_t)n * 2 * HW);      // plane * HW + pixel
    out[i] = __fadd_rn(__fmul_rn(hash_uniform(seeds[n], k), 2.f), -1.f);
}

// one pass of a separable filter with `ksize` taps (odd, <= 63), BORDER_REFLECT_101; horizontal = 1: along x, else along y
__global__ void sep_filter_f32_kernel(const float* src, float* dst, const float* taps, int ksize, int H, int W, int64_t planes,
                                      int horizontal) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float* S = src + (i / ((int64_t)W * H)) * ((int64_t)W * H);
    const int r = ksize / 2;
    float acc = 0.f;
    for (int t = 0; t < ksize; ++t) {                               // the tap order of a plain row / column filter
        const float v = horizontal ? S[(int64_t)y * W + aug_reflect101(x + t - r, W)] : S[(int64_t)aug_reflect101(y + t - r, H) * W + x];
        acc = __fadd_rn(acc, __fmul_rn(v, taps[t]));
    }
    dst[i] = acc;
}

// dst(x, y) = src(x + alpha[n] * dx(x, y), y + alpha[n] * dy(x, y)), BORDER_REFLECT_101; disp = [N][2][H][W]
__global__ void remap_u8_kernel(const unsigned char* src, unsigned char* dst, const float* disp, const float* alpha, int N, int H,
                                int W, int nearest) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int HW = H * W;
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / HW);
    const float al = alpha[n];
    if (al == 0.f) { dst[i] = src[i]; return; }                     // transform not drawn for this frame
    const float* D = disp + (int64_t)n * 2 * HW;
    const float fx = __fadd_rn((float)x, __fmul_rn(D[(int64_t)y * W + x], al));
    const float fy = __fadd_rn((float)y, __fmul_rn(D[HW + (int64_t)y * W + x], al));
    const unsigned char* S = src + (int64_t)n * HW;
    if (nearest) {
        const int xs = aug_reflect101((int)floorf(fx + 0.5f), W), ys = aug_reflect101((int)floorf(fy + 0.5f), H);
        dst[i] = S[(int64_t)ys * W + xs];
        return;
    }
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float a = fx - x0f, b = fy - y0f;
    const int xa = aug_reflect101(x0, W), xb = aug_reflect101(x0 + 1, W), ya = aug_reflect101(y0, H), yb = aug_reflect101(y0 + 1, H);
    const float r0 = __fadd_rn(__fmul_rn((float)S[(int64_t)ya * W + xa], 1.f - a), __fmul_rn((float)S[(int64_t)ya * W + xb], a));
    const float r1 = __fadd_rn(__fmul_rn((float)S[(int64_t)yb * W + xa], 1.f - a), __fmul_rn((float)S[(int64_t)yb * W + xb], a));
    dst[i] = aug_sat_u8(__float2int_rn(__fadd_rn(__fmul_rn(r0, 1.f - b), __fmul_rn(r1, b))));
}

// out[n] = flags[n] ? a[n] : b[n]  (CLAHE / MedianBlur are drawn per frame with p = 0.5 in the reference's Compose)
__global__ void select_frames_u8_kernel(const unsigned char* a, const unsigned char* b, const unsigned char* flags, unsigned char* out,
                                        int HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = flags[i / HW] ? a[i] : b[i];
}
// horizontal flip of the frames whose flag is set
__global__ void hflip_frames_u8_kernel(const unsigned char* src, unsigned char* dst, const unsigned char* flags, int H, int W, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W);
    const int64_t row = i - x;
    dst[i] = flags[i / ((int64_t)H * W)] ? src[row + (W - 1 - x)] : src[i];
}

}  // namespace aau

using namespace aau;

#define AUG_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)
#define AUG_CHECK_DIMS(fn, N, H, W) \
    AAU_REQUIRE((N) > 0 && (H) > 0 && (W) > 0 && (int64_t)(N) * (H) * (W) < 0x7fffffff, fn ": N=%d H=%d W=%d out of range", (int)(N), (int)(H), (int)(W))

extern "C" int aau_warp_affine_u8(const uint8_t* src, uint8_t* dst, const double* inv_mats, int N, int H, int W, int nearest,
                                  int border, void* stream) {
    AAU_REQUIRE(src && dst && inv_mats && src != dst && border >= 0 && border <= 255, "aau_warp_affine_u8: bad args");
    AUG_CHECK_DIMS("aau_warp_affine_u8", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(warp_affine_u8_kernel, AUG_GRID((int64_t)N * H * W), 0, s, src, dst, inv_mats, N, H, W, nearest, border);
    return check_launch("aau_warp_affine_u8");
}

extern "C" int aau_lut_u8(const uint8_t* src, uint8_t* dst, const uint8_t* luts, int N, int64_t HW, void* stream) {
    AAU_REQUIRE(src && dst && luts && N > 0 && HW > 0 && N * HW < 0x7fffffff, "aau_lut_u8: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(lut_u8_kernel, AUG_GRID(N * HW), 0, s, src, dst, luts, (int)HW, N * HW);
    return check_launch("aau_lut_u8");
}

extern "C" int aau_elastic_noise(const uint64_t* seeds, float* out, int N, int H, int W, void* stream) {
    AAU_REQUIRE(seeds && out, "aau_elastic_noise: bad args");
    AUG_CHECK_DIMS("aau_elastic_noise", 2 * N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(elastic_noise_kernel, AUG_GRID((int64_t)N * 2 * H * W), 0, s, seeds, out, H * W, N);
    return check_launch("aau_elastic_noise");
}

extern "C" int aau_gauss_sep_f32(const float* src, float* dst, float* tmp, const float* taps, int ksize, int64_t planes, int H, int W,
                                 void* stream) {
    AAU_REQUIRE(src && dst && tmp && taps && tmp != src && tmp != dst && ksize >= 1 && ksize <= 63 && (ksize & 1),
                "aau_gauss_sep_f32: bad args (odd ksize <= 63, distinct tmp)");
    AAU_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * H * W < 0x7fffffff, "aau_gauss_sep_f32: size out of range");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(sep_filter_f32_kernel, AUG_GRID(planes * H * W), 0, s, src, tmp, taps, ksize, H, W, planes, 1);
    hipLaunchKernelGGL(sep_filter_f32_kernel, AUG_GRID(planes * H * W), 0, s, (const float*)tmp, dst, taps, ksize, H, W, planes, 0);
    return check_launch("aau_gauss_sep_f32");
}

extern "C" int aau_remap_u8(const uint8_t* src, uint8_t* dst, const float* disp, const float* alpha, int N, int H, int W, int nearest,
                            void* stream) {
    AAU_REQUIRE(src && dst && disp && alpha && src != dst, "aau_remap_u8: bad args");
    AUG_CHECK_DIMS("aau_remap_u8", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(remap_u8_kernel, AUG_GRID((int64_t)N * H * W), 0, s, src, dst, disp, alpha, N, H, W, nearest);
    return check_launch("aau_remap_u8");
}

extern "C" int aau_select_frames_u8(const uint8_t* a, const uint8_t* b, const uint8_t* flags, uint8_t* out, int N, int64_t HW,
                                    void* stream) {
    AAU_REQUIRE(a && b && flags && out && N > 0 && HW > 0 && N * HW < 0x7fffffff, "aau_select_frames_u8: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(select_frames_u8_kernel, AUG_GRID(N * HW), 0, s, a, b, flags, out, (int)HW, N * HW);
    return check_launch("aau_select_frames_u8");
}

extern "C" int aau_hflip_frames_u8(const uint8_t* src, uint8_t* dst, const uint8_t* flags, int N, int H, int W, void* stream) {
    AAU_REQUIRE(src && dst && flags && src != dst, "aau_hflip_frames_u8: bad args");
    AUG_CHECK_DIMS("aau_hflip_frames_u8", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(2, 0, s);
    hipLaunchKernelGGL(hflip_frames_u8_kernel, AUG_GRID((int64_t)N * H * W), 0, s, src, dst, flags, H, W, (int64_t)N * H * W);
    return check_launch("aau_hflip_frames_u8");
}

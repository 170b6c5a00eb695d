// Frame pre- / post-processing of the reference's video loop on the device (HBM-bound byte kernels), so that only the
// uint8 frames cross PCIe and no per-frame image processing is left on the host:
//   gray_area_u8     cv2.resize(cv2.cvtColor(frame, COLOR_BGR2GRAY), (256, 256), interpolation=INTER_AREA) then
//                    x / 255 * 2 - 1                                   (reference main_new.py:639,653-656,664-667)
//   area_half_u8     cv2.resize(samples, (w/2, h/2), INTER_AREA) + cv2.cvtColor(COLOR_BGR2RGB)   (main_new.py:723-725)
// OpenCV is not in this image, so these follow OpenCV 4.x's documented algorithms and are pinned to a numpy restatement of
// them (oracle/frameio_ref.py), not to cv2 itself ("parity unpinned" against the reference for these two steps):
//   * BGR2GRAY, 8-bit: (B*3735 + G*19235 + R*9798 + 2^14) >> 15                       (imgproc color_rgb: RGB2Gray<uchar>)
//   * INTER_AREA, non-integer ratio: computeResizeAreaTab / ResizeArea_Invoker -- per destination cell the covered source
//     pixels with fractional weights in float, horizontal pass first, rows accumulated in order, saturate_cast<uchar>
//     (round half to even); every product is rounded before it is added (mul_rounded, common.h: no FMA contraction), as the
//     scalar C++ does
//   * INTER_AREA, ratio 2 x 2, 8-bit: (a + b + c + d + 2) >> 2                              (ResizeAreaFastVec_SIMD_8u)
#include "common.h"

namespace pws {

constexpr int AREA_MAX_TAPS = 12;   // source pixels per destination cell and axis (scale <= 10)

// computeResizeAreaTab for ONE destination index d (double arithmetic, float weights), source size ssize, scale = ssize/dsize:
// an optional partially covered first pixel, the fully covered ones sx1 .. sx2-1, an optional partially covered last one.
// Kept as scalars (no per-lane arrays: dynamic indexing would put them in scratch).
struct AreaAxis {
    int n, sx1, nfull, sx2;
    bool first;
    float a_first, a_full, a_last;
    __device__ __forceinline__ void tap(int k, int &si, float &alpha) const {
        if (first) {
            if (k == 0) {
                si = sx1 - 1, alpha = a_first;
                return;
            }
            --k;
        }
        if (k < nfull) si = sx1 + k, alpha = a_full;
        else si = sx2, alpha = a_last;
    }
};
__device__ __forceinline__ AreaAxis area_axis(int d, int ssize, double scale) {
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    AreaAxis a;
    a.sx1 = sx1, a.sx2 = sx2, a.nfull = sx2 - sx1;
    a.first = sx1 - fsx1 > 1e-3;
    a.a_first = (float)((sx1 - fsx1) / cell), a.a_full = (float)(1.0 / cell);
    const bool last = fsx2 - sx2 > 1e-3;
    a.a_last = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    a.n = (a.first ? 1 : 0) + a.nfull + (last ? 1 : 0);
    return a;
}

__device__ __forceinline__ int gray_u8(const unsigned char *p, int swap_rb) {  // p: B, G, R (or R, G, B when swap_rb)
    const int b = swap_rb ? p[2] : p[0], g = p[1], r = swap_rb ? p[0] : p[2];
    return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15;
}

// one lane per destination pixel
__global__ void __launch_bounds__(256) gray_area_u8_kernel(const unsigned char *__restrict__ frames, float *__restrict__ out, int H,
                                                           int W, int OH, int OW, double sy, double sx, int normalize, int swap_rb,
                                                           size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int dx = (int)(i % OW), dy = (int)((i / OW) % OH);
    const size_t n = i / ((size_t)OW * OH);
    const AreaAxis tx = area_axis(dx, W, sx), ty = area_axis(dy, H, sy);
    const unsigned char *f = frames + n * (size_t)H * W * 3;
    float sum = 0.f;
    for (int j = 0; j < ty.n; ++j) {
        int sj, sk;
        float beta, alpha;
        ty.tap(j, sj, beta);
        const unsigned char *row = f + (size_t)sj * W * 3;
        float buf = 0.f;
        for (int k = 0; k < tx.n; ++k) {
            tx.tap(k, sk, alpha);
            buf = buf + mul_rounded((float)gray_u8(row + (size_t)sk * 3, swap_rb), alpha);
        }
        sum = j == 0 ? mul_rounded(beta, buf) : sum + mul_rounded(beta, buf);
    }
    const float g8 = fminf(fmaxf(rintf(sum), 0.f), 255.f);   // saturate_cast<uchar>: round half to even, clamp
    out[i] = normalize ? g8 / 255 * 2 - 1 : g8;              // main_new.py:643: x.float() / 255 * 2 - 1
}

// one lane per destination pixel (3 channels): 2 x 2 mean with rounding, optional R <-> B swap
__global__ void __launch_bounds__(256) area_half_u8_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int H,
                                                           int W, int swap_rb, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int OW = W / 2, OH = H / 2;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const size_t n = i / ((size_t)OW * OH);
    const unsigned char *p = in + ((n * H + 2 * oy) * (size_t)W + 2 * ox) * 3;
    const unsigned char *q = p + (size_t)W * 3;
    unsigned char r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = (unsigned char)((p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2);
    unsigned char *o = out + i * 3;
    o[0] = swap_rb ? r[2] : r[0], o[1] = r[1], o[2] = swap_rb ? r[0] : r[2];
}

// ---- cv2.resize(frame, (ow, oh), INTER_AREA) of a 3-channel 8-bit frame for ANY down-scaling ratio (reference main_new.py:723:
// every output frame goes to (640, 360) whatever the source size).  OpenCV 4.x, imgproc/resize.cpp:
//   * both ratios integer (1080p -> 3 x 3, 2160p -> 6 x 6): resizeAreaFast_<uchar, int, ...>: the int sum of the iscale_y x iscale_x
//     source pixels times float(1 / area), saturate_cast<uchar> (cvRound: half to even).  (2 x 2 has its own SIMD path,
//     (a + b + c + d + 2) >> 2: area_half_u8_kernel above.)
//   * otherwise computeResizeAreaTab + ResizeArea_Invoker per channel, exactly as gray_area_u8_kernel does for one plane.
// one lane per destination pixel (3 channels)
__global__ void __launch_bounds__(256) area_int_u8_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int H, int W,
                                                          int OH, int OW, int ky, int kx, float scale, int swap_rb, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const size_t n = i / ((size_t)OW * OH);
    const unsigned char *p = in + ((n * H + (size_t)oy * ky) * W + (size_t)ox * kx) * 3;
    int s0 = 0, s1 = 0, s2 = 0;
    for (int j = 0; j < ky; ++j) {
        const unsigned char *q = p + (size_t)j * W * 3;
        for (int k = 0; k < kx; ++k) s0 += q[3 * k], s1 += q[3 * k + 1], s2 += q[3 * k + 2];
    }
    const float r0 = fminf(fmaxf(rintf(mul_rounded((float)s0, scale)), 0.f), 255.f);
    const float r1 = fminf(fmaxf(rintf(mul_rounded((float)s1, scale)), 0.f), 255.f);
    const float r2 = fminf(fmaxf(rintf(mul_rounded((float)s2, scale)), 0.f), 255.f);
    unsigned char *o = out + i * 3;
    o[0] = (unsigned char)(swap_rb ? r2 : r0), o[1] = (unsigned char)r1, o[2] = (unsigned char)(swap_rb ? r0 : r2);
}

__global__ void __launch_bounds__(256) area_tab_u8_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int H, int W,
                                                          int OH, int OW, double sy, double sx, int swap_rb, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int dx = (int)(i % OW), dy = (int)((i / OW) % OH);
    const size_t n = i / ((size_t)OW * OH);
    const AreaAxis tx = area_axis(dx, W, sx), ty = area_axis(dy, H, sy);
    const unsigned char *f = in + n * (size_t)H * W * 3;
    float sum[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ty.n; ++j) {
        int sj, sk;
        float beta, alpha;
        ty.tap(j, sj, beta);
        const unsigned char *row = f + (size_t)sj * W * 3;
        float buf[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < tx.n; ++k) {
            tx.tap(k, sk, alpha);
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[c] = buf[c] + mul_rounded((float)row[(size_t)sk * 3 + c], alpha);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) sum[c] = j == 0 ? mul_rounded(beta, buf[c]) : sum[c] + mul_rounded(beta, buf[c]);
    }
    unsigned char r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = (unsigned char)fminf(fmaxf(rintf(sum[c]), 0.f), 255.f);
    unsigned char *o = out + i * 3;
    o[0] = swap_rb ? r[2] : r[0], o[1] = r[1], o[2] = swap_rb ? r[0] : r[2];
}

}  // namespace pws

using namespace pws;

extern "C" int pws_gray_area_u8(const unsigned char *frames_hwc, float *out, int n, int h, int w, int oh, int ow, int normalize,
                                int swap_rb, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && oh <= h && ow <= w, "pws_gray_area_u8: bad shape (down-scaling only)");
    PWS_REQUIRE((double)h / oh <= AREA_MAX_TAPS - 2 && (double)w / ow <= AREA_MAX_TAPS - 2, "pws_gray_area_u8: scale above %d",
                AREA_MAX_TAPS - 2);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(frames_hwc && out, "pws_gray_area_u8: NULL pointer");
    const size_t total = (size_t)n * oh * ow;
    // cv::resize: inv_scale = (double)dsize / ssize, scale = 1. / inv_scale
    const double sy = 1.0 / ((double)oh / h), sx = 1.0 / ((double)ow / w);
    ProfScope prof(KID_OBJECTIVE, 10.0 * n * h * w, 3.0 * n * h * w + 4.0 * total, as_stream(stream));
    hipLaunchKernelGGL(gray_area_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), frames_hwc, out, h, w,
                       oh, ow, sy, sx, normalize, swap_rb, total);
    return check_launch("gray_area_u8_kernel");
}

extern "C" int pws_area_half_u8(const unsigned char *in_hwc, unsigned char *out_hwc, int n, int h, int w, int swap_rb,
                                pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "pws_area_half_u8: h and w must be even");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(in_hwc && out_hwc, "pws_area_half_u8: NULL pointer");
    const size_t total = (size_t)n * (h / 2) * (w / 2);
    ProfScope prof(KID_OBJECTIVE, 12.0 * total, 15.0 * total, as_stream(stream));
    hipLaunchKernelGGL(area_half_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in_hwc, out_hwc, h, w,
                       swap_rb, total);
    return check_launch("area_half_u8_kernel");
}

extern "C" int pws_area_resize_u8(const unsigned char *in_hwc, unsigned char *out_hwc, int n, int h, int w, int oh, int ow, int swap_rb,
                                  pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && oh <= h && ow <= w, "pws_area_resize_u8: bad shape (down-scaling only)");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(in_hwc && out_hwc, "pws_area_resize_u8: NULL pointer");
    if (h == 2 * oh && w == 2 * ow) return pws_area_half_u8(in_hwc, out_hwc, n, h, w, swap_rb, stream);   // OpenCV's 2 x 2 SIMD path
    const size_t total = (size_t)n * oh * ow;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 3.0 * n * h * w, 3.0 * n * h * w + 3.0 * total, as_stream(stream));
    if (h % oh == 0 && w % ow == 0) {
        // cv::resize: is_area_fast when both scale factors are integers; scale = 1.f / (iscale_x * iscale_y)
        const int ky = h / oh, kx = w / ow;
        hipLaunchKernelGGL(area_int_u8_kernel, dim3(nb), dim3(256), 0, as_stream(stream), in_hwc, out_hwc, h, w, oh, ow, ky, kx,
                           1.f / (float)(kx * ky), swap_rb, total);
        return check_launch("area_int_u8_kernel");
    }
    PWS_REQUIRE((double)h / oh <= AREA_MAX_TAPS - 2 && (double)w / ow <= AREA_MAX_TAPS - 2, "pws_area_resize_u8: scale above %d",
                AREA_MAX_TAPS - 2);
    const double sy = 1.0 / ((double)oh / h), sx = 1.0 / ((double)ow / w);
    hipLaunchKernelGGL(area_tab_u8_kernel, dim3(nb), dim3(256), 0, as_stream(stream), in_hwc, out_hwc, h, w, oh, ow, sy, sx, swap_rb, total);
    return check_launch("area_tab_u8_kernel");
}

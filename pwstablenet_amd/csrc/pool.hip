// Small HBM-bound kernels of the VGG-16 perceptual term (reference lib/utils.py:11-32: nn.Sequential(*vgg16.features[:31])
// -> 13 x (conv3x3 + ReLU, run by the conv kernels of this library) and 5 x MaxPool2d(2, 2); MSELoss between the features of
// the warped frame and of the stable frame).  NHWC fp32, one lane per (output pixel, 4 channels); bf16-storage variants of the pooling: 8 channels per lane.
#include "common.h"

namespace pws {

// y[n,oy,ox,c] = max over the 2x2 window (H, W even)
__global__ void __launch_bounds__(256) maxpool2x2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int OH, int OW, int C4,
                                                             size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const size_t op = i / C4;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH);
    const size_t n = op / ((size_t)OW * OH);
    const int W = 2 * OW, C = 4 * C4;
    const float *p = x + ((n * 2 * OH + 2 * oy) * W + 2 * ox) * C + 4 * c4;
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + C);
    const float4 c = *reinterpret_cast<const float4 *>(p + (size_t)W * C), d = *reinterpret_cast<const float4 *>(p + (size_t)W * C + C);
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    *reinterpret_cast<float4 *>(y + op * C + 4 * c4) = r;
}

// dx = dy routed to the FIRST maximum of the window in scan order (row-major), as ATen's saved argmax does; the other
// three positions get 0 (dx is fully overwritten)
__device__ __forceinline__ void route(float a, float b, float c, float d, float g, float &ga, float &gb, float &gc, float &gd) {
    int k = 0;
    float m = a;
    if (b > m) m = b, k = 1;
    if (c > m) m = c, k = 2;
    if (d > m) m = d, k = 3;
    ga = k == 0 ? g : 0.f, gb = k == 1 ? g : 0.f, gc = k == 2 ? g : 0.f, gd = k == 3 ? g : 0.f;
}
__global__ void __launch_bounds__(256) maxpool2x2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                             float *__restrict__ dx, int OH, int OW, int C4, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const size_t op = i / C4;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH);
    const size_t n = op / ((size_t)OW * OH);
    const int W = 2 * OW, C = 4 * C4;
    const size_t o = ((n * 2 * OH + 2 * oy) * W + 2 * ox) * C + 4 * c4;
    const float *p = x + o;
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + C);
    const float4 c = *reinterpret_cast<const float4 *>(p + (size_t)W * C), d = *reinterpret_cast<const float4 *>(p + (size_t)W * C + C);
    const float4 g = *reinterpret_cast<const float4 *>(dy + op * C + 4 * c4);
    float4 ga, gb, gc, gd;
    route(a.x, b.x, c.x, d.x, g.x, ga.x, gb.x, gc.x, gd.x);
    route(a.y, b.y, c.y, d.y, g.y, ga.y, gb.y, gc.y, gd.y);
    route(a.z, b.z, c.z, d.z, g.z, ga.z, gb.z, gc.z, gd.z);
    route(a.w, b.w, c.w, d.w, g.w, ga.w, gb.w, gc.w, gd.w);
    float *q = dx + o;
    *reinterpret_cast<float4 *>(q) = ga, *reinterpret_cast<float4 *>(q + C) = gb;
    *reinterpret_cast<float4 *>(q + (size_t)W * C) = gc, *reinterpret_cast<float4 *>(q + (size_t)W * C + C) = gd;
}

// bf16-storage variants (PWS_STORE_BF16): one lane per (output pixel, 8 channels) = one 16-byte load per window position.
// max / routing on the fp32 expansions are exact (the result is one of the inputs, or 0).
__device__ __forceinline__ void unpack8(const u32x4 u, float (&f)[8]) {
    f[0] = __builtin_bit_cast(float, u.x << 16), f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u.y << 16), f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, u.z << 16), f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, u.w << 16), f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 u;
    u.x = cvt_pk_bf16(f[0], f[1]), u.y = cvt_pk_bf16(f[2], f[3]), u.z = cvt_pk_bf16(f[4], f[5]), u.w = cvt_pk_bf16(f[6], f[7]);
    return u;
}
__global__ void __launch_bounds__(256) maxpool2x2_fwd16_kernel(const __bf16 *__restrict__ x, __bf16 *__restrict__ y, int OH, int OW, int C8,
                                                               size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    const size_t op = i / C8;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH);
    const size_t n = op / ((size_t)OW * OH);
    const int W = 2 * OW, C = 8 * C8;
    const __bf16 *p = x + ((n * 2 * OH + 2 * oy) * W + 2 * ox) * C + 8 * c8;
    float a[8], b[8], c[8], d[8], r[8];
    unpack8(*reinterpret_cast<const u32x4 *>(p), a), unpack8(*reinterpret_cast<const u32x4 *>(p + C), b);
    unpack8(*reinterpret_cast<const u32x4 *>(p + (size_t)W * C), c), unpack8(*reinterpret_cast<const u32x4 *>(p + (size_t)W * C + C), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
    *reinterpret_cast<u32x4 *>(y + op * C + 8 * c8) = pack8(r);
}
// relu_mask: x is the output of a ReLU; dx is additionally multiplied by ReLU'(x) = (x > 0), i.e. it is the gradient wrt the
// PRE-activation (the routed position holds the window maximum: the mask only matters where that maximum is 0)
__global__ void __launch_bounds__(256) maxpool2x2_bwd16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                               __bf16 *__restrict__ dx, int OH, int OW, int C8, size_t total,
                                                               int relu_mask) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    const size_t op = i / C8;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH);
    const size_t n = op / ((size_t)OW * OH);
    const int W = 2 * OW, C = 8 * C8;
    const size_t o = ((n * 2 * OH + 2 * oy) * W + 2 * ox) * C + 8 * c8;
    const __bf16 *p = x + o;
    float a[8], b[8], c[8], d[8], g[8], ga[8], gb[8], gc[8], gd[8];
    unpack8(*reinterpret_cast<const u32x4 *>(p), a), unpack8(*reinterpret_cast<const u32x4 *>(p + C), b);
    unpack8(*reinterpret_cast<const u32x4 *>(p + (size_t)W * C), c), unpack8(*reinterpret_cast<const u32x4 *>(p + (size_t)W * C + C), d);
    unpack8(*reinterpret_cast<const u32x4 *>(dy + op * C + 8 * c8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        route(a[k], b[k], c[k], d[k], g[k], ga[k], gb[k], gc[k], gd[k]);
        if (relu_mask) {
            ga[k] = a[k] > 0.f ? ga[k] : 0.f, gb[k] = b[k] > 0.f ? gb[k] : 0.f;
            gc[k] = c[k] > 0.f ? gc[k] : 0.f, gd[k] = d[k] > 0.f ? gd[k] : 0.f;
        }
    }
    __bf16 *q = dx + o;
    *reinterpret_cast<u32x4 *>(q) = pack8(ga), *reinterpret_cast<u32x4 *>(q + C) = pack8(gb);
    *reinterpret_cast<u32x4 *>(q + (size_t)W * C) = pack8(gc), *reinterpret_cast<u32x4 *>(q + (size_t)W * C + C) = pack8(gd);
}

// slots += sum (a - b)^2 (fp32 per lane, fp64 per workgroup, one f64 atomic per workgroup into one of PWS_OBJ_SLOTS slots)
__global__ void __launch_bounds__(256) sqdiff_sum_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t count4,
                                                         double *__restrict__ slots) {
    __shared__ double red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (size_t)gridDim.x * 256) {
        const float4 u = *reinterpret_cast<const float4 *>(a + i * 4), v = *reinterpret_cast<const float4 *>(b + i * 4);
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
        s += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    double d = (double)s;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(slots + (blockIdx.x % PWS_OBJ_SLOTS), red[0] + red[1] + red[2] + red[3]);
}

// ga = c * scale * 2 (a - b)
__global__ void __launch_bounds__(256) sqdiff_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t count4, float c,
                                                         const float *__restrict__ scale, float *__restrict__ ga) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    if (scale) c *= *scale;
    const float4 u = *reinterpret_cast<const float4 *>(a + i * 4), v = *reinterpret_cast<const float4 *>(b + i * 4);
    *reinterpret_cast<float4 *>(ga + i * 4) = make_float4(2.f * c * (u.x - v.x), 2.f * c * (u.y - v.y), 2.f * c * (u.z - v.z), 2.f * c * (u.w - v.w));
}

static inline bool a16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace pws

using namespace pws;

extern "C" int pws_maxpool2x2_fwd(const float *x, float *y, int n, int h, int w, int c, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0 && c % 4 == 0,
                "pws_maxpool2x2_fwd: h, w must be even and c a multiple of 4 (got %d x %d x %d)", h, w, c);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && y && a16(x) && a16(y), "pws_maxpool2x2_fwd: NULL or unaligned pointer");
    const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 4);
    ProfScope prof(KID_OBJECTIVE, 3.0 * total * 4, 20.0 * total * 4, as_stream(stream));
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, h / 2, w / 2,
                       c / 4, total);
    return check_launch("maxpool2x2_fwd_kernel");
}

extern "C" int pws_maxpool2x2_bwd(const float *x, const float *dy, float *dx, int n, int h, int w, int c, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0 && c % 4 == 0,
                "pws_maxpool2x2_bwd: h, w must be even and c a multiple of 4 (got %d x %d x %d)", h, w, c);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && dy && dx && a16(x) && a16(dy) && a16(dx), "pws_maxpool2x2_bwd: NULL or unaligned pointer");
    const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 4);
    ProfScope prof(KID_OBJECTIVE, 8.0 * total * 4, 36.0 * total * 4, as_stream(stream));
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, dy, dx, h / 2,
                       w / 2, c / 4, total);
    return check_launch("maxpool2x2_bwd_kernel");
}

extern "C" int pws_maxpool2x2_fwd_s(const void *x, void *y, int n, int h, int w, int c, int store, pws_stream_t stream) {
    if (store == PWS_STORE_FP32) return pws_maxpool2x2_fwd(static_cast<const float *>(x), static_cast<float *>(y), n, h, w, c, stream);
    PWS_REQUIRE(store == PWS_STORE_BF16, "pws_maxpool2x2_fwd_s: bad store %d", store);
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0,
                "pws_maxpool2x2_fwd_s: h, w must be even and c a multiple of 8 for bf16 storage (got %d x %d x %d)", h, w, c);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && y && a16(x) && a16(y), "pws_maxpool2x2_fwd_s: NULL or unaligned pointer");
    const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 8);
    ProfScope prof(KID_OBJECTIVE, 3.0 * total * 8, 10.0 * total * 8, as_stream(stream));
    hipLaunchKernelGGL(maxpool2x2_fwd16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16 *>(x), static_cast<__bf16 *>(y), h / 2, w / 2, c / 8, total);
    return check_launch("maxpool2x2_fwd16_kernel");
}

extern "C" int pws_maxpool2x2_bwd_s(const void *x, const void *dy, void *dx, int n, int h, int w, int c, int store, int relu_mask,
                                    pws_stream_t stream) {
    if (store == PWS_STORE_FP32) {
        PWS_REQUIRE(!relu_mask, "pws_maxpool2x2_bwd_s: relu_mask needs bf16 storage");
        return pws_maxpool2x2_bwd(static_cast<const float *>(x), static_cast<const float *>(dy), static_cast<float *>(dx), n, h, w, c, stream);
    }
    PWS_REQUIRE(store == PWS_STORE_BF16, "pws_maxpool2x2_bwd_s: bad store %d", store);
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0,
                "pws_maxpool2x2_bwd_s: h, w must be even and c a multiple of 8 for bf16 storage (got %d x %d x %d)", h, w, c);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && dy && dx && a16(x) && a16(dy) && a16(dx), "pws_maxpool2x2_bwd_s: NULL or unaligned pointer");
    const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 8);
    ProfScope prof(KID_OBJECTIVE, 8.0 * total * 8, 18.0 * total * 8, as_stream(stream));
    hipLaunchKernelGGL(maxpool2x2_bwd16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16 *>(x), static_cast<const __bf16 *>(dy), static_cast<__bf16 *>(dx), h / 2, w / 2, c / 8, total,
                       relu_mask ? 1 : 0);
    return check_launch("maxpool2x2_bwd16_kernel");
}

extern "C" int pws_sqdiff_sum(const float *a, const float *b, size_t count, double *slots, pws_stream_t stream) {
    PWS_REQUIRE(count % 4 == 0, "pws_sqdiff_sum: count must be a multiple of 4");
    if (count == 0) return PWS_OK;
    PWS_REQUIRE(a && b && slots && a16(a) && a16(b), "pws_sqdiff_sum: NULL or unaligned pointer");
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sqdiff_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, b, count / 4, slots);
    return check_launch("sqdiff_sum_kernel");
}

extern "C" int pws_sqdiff_bwd(const float *a, const float *b, size_t count, float c, const float *scale, float *ga,
                              pws_stream_t stream) {
    PWS_REQUIRE(count % 4 == 0, "pws_sqdiff_bwd: count must be a multiple of 4");
    if (count == 0) return PWS_OK;
    PWS_REQUIRE(a && b && ga && a16(a) && a16(b) && a16(ga), "pws_sqdiff_bwd: NULL or unaligned pointer");
    hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, count / 4, c,
                       scale, ga);
    return check_launch("sqdiff_bwd_kernel");
}

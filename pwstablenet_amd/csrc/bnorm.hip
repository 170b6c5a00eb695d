// Training-mode BatchNorm2d + activation for the use_BN variant of the generator (reference lib/networks_cascading.py:253-341:
// conv -> nn.BatchNorm2d -> LeakyReLU / ReLU / Tanh in every block), NHWC fp32, any channel count.
//   forward : mean / biased variance per channel over the `pixels` rows of z, y = act(gamma * (z - mean) * invstd + beta),
//             running_mean / running_var updated with `momentum` (unbiased variance), mean and invstd saved for backward
//   backward: dyh = dy * act'(y); dgamma += sum dyh * xh; dbeta += sum dyh;
//             dz = gamma * invstd * (dyh - mean(dyh) - xh * mean(dyh * xh)), written over dy
// Reductions: a workgroup owns 64 channels x a pixel range (lane = channel, 4 pixel lanes), partial sums per workgroup go
// through slabs, a second small launch adds them in double precision (deterministic, no atomics).
#include "common.h"

namespace pws {

constexpr int BN_MAX_SLABS = 1024;

__device__ __forceinline__ float bn_act(float v, int act) {
    if (act == PWS_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == PWS_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}
__device__ __forceinline__ float bn_act_grad(float y, int act) {   // in terms of the OUTPUT of the activation
    if (act == PWS_ACT_LRELU) return y > 0.f ? 1.f : 0.2f;
    if (act == PWS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}

// MODE 0: (sum z, sum z^2)        MODE 1: (sum dyh, sum dyh * xh) with dyh = dy * act'(y), xh = (z - mean) * invstd
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float *__restrict__ z, const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ stats, size_t pixels, int c, int act,
                                                        float *__restrict__ slabs) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + cl;
    float a = 0.f, b = 0.f;
    if (ch < c) {
        float mean = 0.f, invstd = 0.f;
        if (MODE == 1) mean = stats[ch], invstd = stats[c + ch];
        for (size_t p = (size_t)blockIdx.x * 4 + pl; p < pixels; p += (size_t)gridDim.x * 4) {
            const float v = z[p * c + ch];
            if (MODE == 0) {
                a += v, b += v * v;
            } else {
                const float g = dy[p * c + ch] * (y ? bn_act_grad(y[p * c + ch], act) : 1.f);
                a += g, b += g * (v - mean) * invstd;
            }
        }
    }
    red[0][pl][cl] = a, red[1][pl][cl] = b;
    __syncthreads();
    if (pl == 0 && ch < c) {
        float *s = slabs + (size_t)blockIdx.x * 2 * c;
        s[ch] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        s[c + ch] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}

// forward finalize: stats = (mean, invstd); running statistics updated `repeat` times with the same batch statistics (a
// module the reference calls twice on the same input, lib/networks_cascading.py:178,200, is computed once here)
__global__ void bn_fwd_finalize_kernel(const float *__restrict__ slabs, int nslabs, size_t pixels, int c, float eps, float momentum,
                                       int repeat, float *__restrict__ stats, float *__restrict__ running_mean,
                                       float *__restrict__ running_var) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nslabs; ++i) s += slabs[(size_t)i * 2 * c + ch], q += slabs[(size_t)i * 2 * c + c + ch];
    const double mean = s / (double)pixels;
    double var = q / (double)pixels - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[ch] = (float)mean, stats[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && running_var) {
        const double unbiased = pixels > 1 ? var * (double)pixels / (double)(pixels - 1) : var;
        float rm = running_mean[ch], rv = running_var[ch];
        for (int r = 0; r < repeat; ++r) rm = (1.f - momentum) * rm + momentum * (float)mean, rv = (1.f - momentum) * rv + momentum * (float)unbiased;
        running_mean[ch] = rm, running_var[ch] = rv;
    }
}

// backward finalize: sums = (sum dyh, sum dyh * xh); dgamma += sum dyh xh, dbeta += sum dyh
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ slabs, int nslabs, int c, float *__restrict__ sums,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nslabs; ++i) s += slabs[(size_t)i * 2 * c + ch], q += slabs[(size_t)i * 2 * c + c + ch];
    sums[ch] = (float)s, sums[c + ch] = (float)q;
    if (dbeta) dbeta[ch] += (float)s;
    if (dgamma) dgamma[ch] += (float)q;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float *__restrict__ z, const float *__restrict__ stats,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, int c, int act,
                                                       size_t total, float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c);
    y[i] = bn_act(gamma[ch] * ((z[i] - stats[ch]) * stats[c + ch]) + beta[ch], act);
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                           const float *__restrict__ stats, const float *__restrict__ sums,
                                                           const float *__restrict__ gamma, int c, int act, size_t total, float inv_m,
                                                           float *__restrict__ dy) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c);
    const float invstd = stats[c + ch];
    const float xh = (z[i] - stats[ch]) * invstd;
    const float g = dy[i] * (y ? bn_act_grad(y[i], act) : 1.f);
    dy[i] = gamma[ch] * invstd * (g - sums[ch] * inv_m - xh * sums[c + ch] * inv_m);
}

static int bn_slabs(size_t pixels) {
    size_t b = (pixels + 255) / 256;   // >= 64 pixels per pixel lane
    if (b > (size_t)BN_MAX_SLABS) b = BN_MAX_SLABS;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace pws

using namespace pws;

extern "C" size_t pws_bn_ws_bytes(int c) { return c > 0 ? sizeof(float) * ((size_t)BN_MAX_SLABS * 2 * c + 2 * (size_t)c) : 0; }

extern "C" int pws_bn_train_fwd(const float *z, size_t pixels, int c, const float *gamma, const float *beta, int act, float *y,
                                float *stats, float *running_mean, float *running_var, float momentum, float eps, int repeat,
                                void *ws, size_t ws_bytes, pws_stream_t stream) {
    PWS_REQUIRE(c > 0 && act >= PWS_ACT_NONE && act <= PWS_ACT_RELU && repeat >= 0, "pws_bn_train_fwd: bad c / act / repeat");
    if (pixels == 0) return PWS_OK;
    PWS_REQUIRE(pixels > 1, "pws_bn_train_fwd: training-mode BatchNorm needs more than 1 value per channel (as torch: ValueError)");
    PWS_REQUIRE(z && gamma && beta && y && stats && ws && ws_bytes >= pws_bn_ws_bytes(c), "pws_bn_train_fwd: NULL pointer or workspace "
                "smaller than pws_bn_ws_bytes(c)");
    PWS_REQUIRE((running_mean != nullptr) == (running_var != nullptr), "pws_bn_train_fwd: running_mean and running_var go together");
    hipStream_t st = as_stream(stream);
    float *slabs = static_cast<float *>(ws);
    const int nslabs = bn_slabs(pixels);
    ProfScope prof(KID_OBJECTIVE, 8.0 * pixels * c, 12.0 * pixels * c, st);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3((unsigned)nslabs, (unsigned)((c + 63) / 64)), dim3(256), 0, st, z, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, pixels, c, act, slabs);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(64), 0, st, slabs, nslabs, pixels, c, eps, momentum,
                       repeat, stats, running_mean, running_var);
    const size_t total = pixels * c;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, stats, gamma, beta, c, act, total, y);
    return check_launch("bn_train_fwd kernels");
}

extern "C" int pws_bn_train_bwd(float *dy, const float *y, const float *z, const float *stats, const float *gamma, int act, size_t pixels,
                                int c, float *dgamma, float *dbeta, void *ws, size_t ws_bytes, pws_stream_t stream) {
    PWS_REQUIRE(c > 0 && act >= PWS_ACT_NONE && act <= PWS_ACT_RELU, "pws_bn_train_bwd: bad c / act");
    if (pixels == 0) return PWS_OK;
    PWS_REQUIRE(dy && z && stats && gamma && ws && ws_bytes >= pws_bn_ws_bytes(c) && (y || act == PWS_ACT_NONE),
                "pws_bn_train_bwd: NULL pointer or workspace smaller than pws_bn_ws_bytes(c)");
    hipStream_t st = as_stream(stream);
    float *slabs = static_cast<float *>(ws);
    float *sums = slabs + (size_t)BN_MAX_SLABS * 2 * c;
    const int nslabs = bn_slabs(pixels);
    ProfScope prof(KID_OBJECTIVE, 16.0 * pixels * c, 28.0 * pixels * c, st);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3((unsigned)nslabs, (unsigned)((c + 63) / 64)), dim3(256), 0, st, z, dy, y, stats, pixels, c,
                       act, slabs);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(64), 0, st, slabs, nslabs, c, sums, dgamma, dbeta);
    const size_t total = pixels * c;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, y, stats, sums, gamma, c, act, total,
                       1.f / (float)pixels, dy);
    return check_launch("bn_train_bwd kernels");
}

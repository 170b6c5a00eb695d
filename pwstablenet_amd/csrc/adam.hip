// Fused Adam over a flat fp32 parameter buffer (reference main_new.py:63,216: optim.Adam(lr, betas=(beta1,0.999))).
// HBM-bound: 16 B read + 12 B written per parameter; float4 per lane, grid-stride.
#include <cmath>

#include "common.h"

namespace pws {

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n4, size_t count, float b1, float b2,
                                                   float eps, float step_size, float inv_bc2_sqrt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            M[k] = b1 * M[k] + (1.f - b1) * G[k];
            V[k] = b2 * V[k] + (1.f - b2) * G[k] * G[k];
            P[k] -= step_size * (M[k] / (sqrtf(V[k]) * inv_bc2_sqrt + eps));
        }
        reinterpret_cast<float4 *>(p)[i] = pp, reinterpret_cast<float4 *>(m)[i] = mm, reinterpret_cast<float4 *>(v)[i] = vv;
    }
    // ragged tail (count % 4) handled by the first lanes of block 0
    const size_t tail0 = n4 * 4;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < count) {
        const size_t i = tail0 + threadIdx.x;
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi, v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    }
}

// Many tensors in one launch (a generator has 92; one launch per tensor is launch-latency-bound for the biases and small
// layers): the pointer table travels in the kernel arguments, a workgroup of 256 lanes owns 4096 consecutive elements of one
// tensor and finds it by its block index.
constexpr int kAdamMaxTensors = 48;
struct AdamMultiArgs {
    int ntensors;
    unsigned first_block[kAdamMaxTensors];
    unsigned long long count[kAdamMaxTensors];
    float *p[kAdamMaxTensors];
    const float *g[kAdamMaxTensors];
    float *m[kAdamMaxTensors];
    float *v[kAdamMaxTensors];
};
static_assert(sizeof(AdamMultiArgs) <= 4096, "kernel argument block");

__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamMultiArgs a, float b1, float b2, float eps, float step_size,
                                                         float inv_bc2_sqrt) {
    int lo = 0, hi = a.ntensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    float *p = a.p[lo], *m = a.m[lo], *v = a.v[lo];
    const float *g = a.g[lo];
    const size_t count = a.count[lo];
    const size_t base = (size_t)(blockIdx.x - a.first_block[lo]) * 4096;
    const bool al = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                      reinterpret_cast<size_t>(v)) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t e = base + (size_t)(threadIdx.x + 256 * j) * 4;
        if (al && e + 4 <= count) {
            float4 pp = *reinterpret_cast<float4 *>(p + e), gg = *reinterpret_cast<const float4 *>(g + e);
            float4 mm = *reinterpret_cast<float4 *>(m + e), vv = *reinterpret_cast<float4 *>(v + e);
            float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                M[k] = b1 * M[k] + (1.f - b1) * G[k];
                V[k] = b2 * V[k] + (1.f - b2) * G[k] * G[k];
                P[k] -= step_size * (M[k] / (sqrtf(V[k]) * inv_bc2_sqrt + eps));
            }
            *reinterpret_cast<float4 *>(p + e) = pp, *reinterpret_cast<float4 *>(m + e) = mm, *reinterpret_cast<float4 *>(v + e) = vv;
        } else {
            for (size_t i = e; i < e + 4 && i < count; ++i) {
                const float gi = g[i];
                const float mi = b1 * m[i] + (1.f - b1) * gi;
                const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
                m[i] = mi, v[i] = vi;
                p[i] -= step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
            }
        }
    }
}

}  // namespace pws

extern "C" int pws_adam_step_multi(float *const *p, const float *const *g, float *const *m, float *const *v, const size_t *counts,
                                   int ntensors, float lr, float beta1, float beta2, float eps, int step, pws_stream_t stream) {
    PWS_REQUIRE(step >= 1, "pws_adam_step_multi: step counts from 1 (got %d)", step);
    PWS_REQUIRE(ntensors >= 0 && (ntensors == 0 || (p && g && m && v && counts)), "pws_adam_step_multi: bad arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
    double total = 0;
    for (int i = 0; i < ntensors; ++i) total += (double)counts[i];
    pws::ProfScope prof(pws::KID_ADAM, 12.0 * total, 28.0 * total, pws::as_stream(stream));
    for (int i = 0; i < ntensors;) {
        pws::AdamMultiArgs a{};
        unsigned nb = 0;
        int k = 0;
        for (; i < ntensors && k < pws::kAdamMaxTensors; ++i) {
            if (counts[i] == 0) continue;
            PWS_REQUIRE(p[i] && g[i] && m[i] && v[i], "pws_adam_step_multi: NULL pointer for tensor %d", i);
            a.p[k] = p[i], a.g[k] = g[i], a.m[k] = m[i], a.v[k] = v[i], a.count[k] = counts[i];
            a.first_block[k] = nb;
            nb += (unsigned)((counts[i] + 4095) / 4096);
            ++k;
        }
        a.ntensors = k;
        if (k == 0) continue;
        hipLaunchKernelGGL(pws::adam_multi_kernel, dim3(nb), dim3(256), 0, pws::as_stream(stream), a, beta1, beta2, eps, step_size,
                           inv_bc2_sqrt);
    }
    return pws::check_launch("adam_multi_kernel");
}

extern "C" int pws_adam_step(float *p, const float *g, float *m, float *v, size_t count, float lr, float beta1, float beta2,
                             float eps, int step, pws_stream_t stream) {
    PWS_REQUIRE(step >= 1, "pws_adam_step: step counts from 1 (got %d)", step);
    if (count == 0) return PWS_OK;
    PWS_REQUIRE(p && g && m && v, "pws_adam_step: NULL pointer");
    const bool al = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                      reinterpret_cast<size_t>(v)) & 15) == 0;
    const size_t n4 = al ? count / 4 : 0;
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    if (!al) {
        // unaligned buffers: scalar path = tail loop only handles < 256 elements, so fall back to one lane per element
        PWS_REQUIRE(count <= 256, "pws_adam_step: buffers must be 16-byte aligned (count %zu)", count);
    }
    pws::ProfScope prof(pws::KID_ADAM, 12.0 * count, 28.0 * count, pws::as_stream(stream));
    hipLaunchKernelGGL(pws::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, pws::as_stream(stream), p, g, m, v, n4, count,
                       beta1, beta2, eps, step_size, inv_bc2_sqrt);
    return pws::check_launch("adam_kernel");
}

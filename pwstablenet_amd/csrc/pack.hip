// Weight re-layout: torch state_dict layouts (conv OIHW, transposed conv IOHW; reference
// lib/networks_cascading.py:248,306,330,339) -> the kernels' [class][tap][cin padded to 16][cout] layout,
// in which the MFMA B-fragment (one k row, 32 consecutive output channels) is a contiguous 128-B read.
//
//   conv  k x k          : P[ky*k+kx][ci][co]            = W[co][ci][ky][kx]
//   convT k3 s1 p1       : P[r*3+s][ci][co]              = W[ci][co][2-r][2-s]      (flipped correlation)
//   convT k4 s2 p1       : P[py*2+px][dy*2+dx][ci][co]   = W[ci][co][3-py-2dy][3-px-2dx]
//       output (2y+py, 2x+px) = sum_{dy,dx,ci} in(y+py-1+dy, x+px-1+dx, ci) * P[...]   (4 sub-pixel 2x2 convs)
#include "common.h"

namespace pws {

__host__ __device__ inline int kind_k(int kind) {
    switch (kind) {
    case PWS_CONV_K3S1:
    case PWS_CONV_K3S2:
    case PWS_CONVT_K3S1:
    case PWS_CONV_K3S1_OUT: return 3;
    case PWS_CONV_K5S1: return 5;
    case PWS_CONVT_K4S2: return 4;
    case PWS_CONV_K2S1P0: return 2;
    case PWS_CONV_K1: return 1;
    default: return 0;
    }
}

__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ out, int kind, int cin, int cin_pad,
                                   int cout, int k, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = idx % cout;
    size_t t = idx / cout;
    const int ci = t % cin_pad;
    t /= cin_pad;  // tap (and class) index
    float v = 0.f;
    if (ci < cin) {
        if (kind == PWS_CONVT_K4S2) {
            const int tap = t % 4, cls = t / 4;
            const int dy = tap >> 1, dx = tap & 1, py = cls >> 1, px = cls & 1;
            const int ky = 3 - py - 2 * dy, kx = 3 - px - 2 * dx;
            v = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
        } else if (kind == PWS_CONVT_K3S1) {
            const int r = t / 3, s = t % 3;
            v = w[(((size_t)ci * cout + co) * 3 + (2 - r)) * 3 + (2 - s)];
        } else {
            const int ky = t / k, kx = t % k;
            v = w[(((size_t)co * cin + ci) * k + ky) * k + kx];
        }
    }
    out[idx] = v;
}

// Inverse of pack_weight_kernel: one lane per torch-layout element.
__global__ void unpack_weight_kernel(const float *__restrict__ packed, float *__restrict__ w, int kind, int cin, int cin_pad,
                                     int cout, int k, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kx = idx % k, ky = (idx / k) % k;
    const size_t t = idx / ((size_t)k * k);
    size_t tapcls;
    int ci, co;
    if (kind == PWS_CONVT_K4S2 || kind == PWS_CONVT_K3S1) {  // IOHW
        co = t % cout, ci = t / cout;
        if (kind == PWS_CONVT_K4S2) {
            const int py = (3 - ky) & 1, dy = (3 - ky) >> 1, px = (3 - kx) & 1, dx = (3 - kx) >> 1;
            tapcls = (size_t)(py * 2 + px) * 4 + dy * 2 + dx;
        } else {
            tapcls = (size_t)(2 - ky) * 3 + (2 - kx);
        }
    } else {  // OIHW
        ci = t % cin, co = t / cin;
        tapcls = (size_t)ky * k + kx;
    }
    w[idx] = packed[(tapcls * cin_pad + ci) * cout + co];
}

// Data-gradient layouts [class][tap][co_f][ci_f] (rows = forward output channels = the gradient conv's input).
//   conv  k3 s1 p1 : P[r*3+s][co][ci]            = W[co][ci][2-r][2-s]
//   convT k3 s1 p1 : P[r*3+s][co][ci]            = W[ci][co][r][s]
//   conv  k3 s2 p1 : P[py*2+px][dy*2+dx][co][ci] = W[co][ci][R(py,dy)][R(px,dx)],  R(0,0)=1, R(1,0)=2, R(1,1)=0, R(0,1)=none
//       dx(2m+p) = sum_d dy(m+d) W[R(p,d)]   (gradient of y(m) = sum_r x(2m-1+r) W[r])
//   convT k4 s2 p1 : P[ky*4+kx][co][ci]          = W[ci][co][ky][kx]   (a conv k4 s2 p1 over dy)
__global__ void pack_dgrad_kernel(const float *__restrict__ w, float *__restrict__ out, int kind, int cin, int cout,
                                  size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ci = idx % cin;
    size_t t = idx / cin;
    const int co = t % cout;
    t /= cout;
    float v = 0.f;
    if (kind == PWS_CONV_K3S1) {
        const int r = t / 3, s_ = t % 3;
        v = w[(((size_t)co * cin + ci) * 3 + (2 - r)) * 3 + (2 - s_)];
    } else if (kind == PWS_CONVT_K3S1) {
        const int r = t / 3, s_ = t % 3;
        v = w[(((size_t)ci * cout + co) * 3 + r) * 3 + s_];
    } else if (kind == PWS_CONV_K3S2) {
        const int tap = t % 4, cls = t / 4;
        const int dy = tap >> 1, dx = tap & 1, py = cls >> 1, px = cls & 1;
        const int ry = py ? (dy ? 0 : 2) : (dy ? -1 : 1), rx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
        if (ry >= 0 && rx >= 0) v = w[(((size_t)co * cin + ci) * 3 + ry) * 3 + rx];
    } else {  // PWS_CONVT_K4S2
        const int ky = t / 4, kx = t % 4;
        v = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
    }
    out[idx] = v;
}

}  // namespace pws

extern "C" size_t pws_packed_weight_floats(int kind, int cin, int cout) {
    const int k = pws::kind_k(kind);
    if (k == 0 || cin <= 0 || cout <= 0) return 0;
    const size_t cin_pad = (size_t)(cin + 15) / 16 * 16;
    return (size_t)k * k * cin_pad * cout;
}

extern "C" int pws_pack_conv_weight(const float *w_torch, float *w_packed, int kind, int cin, int cout,
                                    pws_stream_t stream) {
    const int k = pws::kind_k(kind);
    PWS_REQUIRE(k != 0, "pws_pack_conv_weight: unknown kind %d", kind);
    PWS_REQUIRE(w_torch && w_packed && cin > 0 && cout > 0, "pws_pack_conv_weight: bad arguments");
    const int cin_pad = (cin + 15) / 16 * 16;
    const size_t total = (size_t)k * k * cin_pad * cout;
    const int threads = 256;
    const size_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(pws::pack_weight_kernel, dim3((unsigned)blocks), dim3(threads), 0, pws::as_stream(stream), w_torch,
                       w_packed, kind, cin, cin_pad, cout, k, total);
    return pws::check_launch("pack_weight_kernel");
}

extern "C" int pws_unpack_conv_weight(const float *w_packed, float *w_torch, int kind, int cin, int cout, pws_stream_t stream) {
    const int k = pws::kind_k(kind);
    PWS_REQUIRE(k != 0 && w_packed && w_torch && cin > 0 && cout > 0, "pws_unpack_conv_weight: bad arguments");
    const int cin_pad = (cin + 15) / 16 * 16;
    const size_t total = (size_t)k * k * cin * cout;
    hipLaunchKernelGGL(pws::unpack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pws::as_stream(stream),
                       w_packed, w_torch, kind, cin, cin_pad, cout, k, total);
    return pws::check_launch("unpack_weight_kernel");
}

static int dgrad_taps(int kind) {
    switch (kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1: return 9;
    case PWS_CONV_K3S2:
    case PWS_CONVT_K4S2: return 16;
    default: return 0;
    }
}

extern "C" size_t pws_packed_dgrad_floats(int kind, int cin, int cout) {
    if (dgrad_taps(kind) == 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)dgrad_taps(kind) * cin * cout;
}

extern "C" int pws_pack_conv_weight_dgrad(const float *w_torch, float *w_packed, int kind, int cin, int cout,
                                          pws_stream_t stream) {
    PWS_REQUIRE(dgrad_taps(kind) != 0, "pws_pack_conv_weight_dgrad: kind %d has no data gradient", kind);
    PWS_REQUIRE(w_torch && w_packed && cin > 0 && cout > 0 && cout % 16 == 0,
                "pws_pack_conv_weight_dgrad: bad arguments (cout must be a multiple of 16)");
    const size_t total = (size_t)dgrad_taps(kind) * cin * cout;
    hipLaunchKernelGGL(pws::pack_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pws::as_stream(stream), w_torch,
                       w_packed, kind, cin, cout, total);
    return pws::check_launch("pack_dgrad_kernel");
}

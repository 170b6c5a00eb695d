// Internal helpers shared by the HIP translation units of libpwstable_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/pwstable.h"

namespace pws {

void set_error(const char *fmt, ...);  // abi.cpp (thread-local buffer)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PWS_EHIP;
    }
    return PWS_OK;
}

#define PWS_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            pws::set_error(__VA_ARGS__); \
            return PWS_EINVAL;          \
        }                               \
    } while (0)

inline hipStream_t as_stream(pws_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kXcds = 8;    // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

// Bijective XCD-aware remap of a 1-D block id: consecutive logical ids [k*chunk, (k+1)*chunk) run on the
// same XCD (same private L2), so neighbouring tiles share halo lines / gather footprints in one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / kXcds, r = nblocks % kXcds;
    const unsigned xcd = bid % kXcds, slot = bid / kXcds;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// One-time launcher state (LDS opt-in via hipFuncSetAttribute, CU count) is per DEVICE: the attribute lands on the current
// device's function object and two devices of one process may differ -- include/pwstable.h promises use on several devices.
constexpr int kMaxDevices = 32;
inline int current_device_slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (int)((unsigned)d % kMaxDevices);
}
struct PerDeviceFlag {
    bool v[kMaxDevices] = {};
    bool &cur() { return v[current_device_slot()]; }
};
struct PerDeviceInt {
    int v[kMaxDevices] = {};
    int &cur() { return v[current_device_slot()]; }
};

#if defined(__HIPCC__)
// a * b rounded to fp32 as a value of its own.  hipcc's default -ffp-contract=fast fuses a product into the add / subtract that
// consumes it (v_fma / v_fmac on the EXACT product) -- also through __fmul_rn / __fadd_rn, which inline to plain fmul / fadd (seen
// in the ISA).  Where the CPU code this library restates rounds the product first (torch's source index, OpenCV's scalar area
// resize), the empty asm pins the rounded product in a register and the consumer cannot fuse it.
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
#endif

// Kernel ids for the measurement hooks (pws_prof_*).
enum KernelId {
    KID_CONV_K3S1_BIG = 0, KID_CONV_K3S1_SMALL, KID_CONV_K3S2_BIG, KID_CONV_K3S2_SMALL, KID_CONV_K5S1, KID_CONVT4_BIG,
    KID_CONVT4_SMALL, KID_THETA_HEAD, KID_FIELD_HEAD, KID_GRID_SAMPLE_FWD, KID_GRID_SAMPLE_BWD, KID_UPSAMPLE_GRID_SAMPLE_FWD,
    KID_UPSAMPLE, KID_AFFINE_GRID, KID_ADAM, KID_PACK, KID_DGRAD_K4S2, KID_DGRAD_SP3, KID_WGRAD, KID_ACT_BWD, KID_FIELD_HEAD_BWD,
    KID_THETA_HEAD_BWD, KID_CONV_WINO, KID_CONV_BF16, KID_WGRAD_BF16, KID_CONV_WINO_CT4, KID_UPSAMPLE_GRID_SAMPLE_U8, KID_OBJECTIVE, KID_CONV_RING, KID_CONV_RINGF, KID_CONV_WRING, KID_CONV_WRING_CT4, KID_CONV_SKINNY, KID_WGRAD_RING, KID_CONV_SKINNY16, KID_CONV_FIRST, KID_CONV_FIRST_WINO, KID_COUNT
};
// Deterministic accumulation (PWS_NETG_DETERMINISTIC / pws_conv_bwd_weight_args.deterministic): while set on the calling thread,
// every launcher that accumulates with fp32 atomics gives each address ONE adding workgroup per launch (no pixel split), the head
// kernels sum in a fixed order.  Set for the duration of one entry-point call only.
extern thread_local bool t_deterministic;
struct DeterministicScope {
    bool prev;
    explicit DeterministicScope(bool on) : prev(t_deterministic) { t_deterministic = on || prev; }
    ~DeterministicScope() { t_deterministic = prev; }
};
extern bool g_two_queues;
extern int g_math;  // PWS_OPT_MATH
extern int g_store;  // PWS_OPT_STORE (effective only with PWS_MATH_BF16)
extern int g_experiment;  // PWS_OPT_EXPERIMENT: selects measured kernel variants (tools/*_bench.py); 0 = product default
extern bool g_prof_on;
extern thread_local int g_prof_tag;  // the layer a launch belongs to, per calling thread
bool prof_begin(int kernel_id, double flops, double bytes, hipStream_t st);  // false: nothing recorded (the stream is being captured)
void prof_end(hipStream_t st);
struct ProfScope {  // brackets one launch with events when profiling is enabled; free otherwise
    hipStream_t st;
    bool on;
    ProfScope(int kernel_id, double flops, double bytes, hipStream_t s) : st(s), on(g_prof_on) {
        if (on) on = prof_begin(kernel_id, flops, bytes, st);
    }
    ~ProfScope() {
        if (on) prof_end(st);
    }
};

int nchw_to_nhwc_pad_strided(const float *x, size_t sample_stride, float *out, int n, int c, int h, int w, int cpad, int store, hipStream_t st);   // conv_bf16.hip

struct ProfHint {
    double flops, bytes;
};
// use_BN training path: pieces of the two heads sequenced by netg.cpp (head.hip / head_bwd.hip)
int theta_z1(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat, float *ws, float *z1, hipStream_t st);
int theta_z2(const float *h, int n, int hidden, const float *w_lin, const float *b_lin, float *z2, hipStream_t st);
int field_head_raw(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *b_out, float *z, hipStream_t st);
int field_bn_finish(const float *yhat, const float *theta, int n, int h, int w, int ac, float *resid, float *grid, hipStream_t st);
int field_bwd_gz(const float *resid, const float *g_grid, const float *g_resid, int n, int h, int w, int ac, float *gz, float *db_out,
                 float *dtheta, hipStream_t st);
int field_bwd_dx_dw(const float *x, int ld, const float *gz, int n, int h, int w, int c, const float *w_out, float *dx, int dx_ld,
                    int dx_accumulate, float *dw_out, int store, hipStream_t st, int dx_act = 0);
int theta_bwd_bn_lin(const float *dz2, const float *h, int n, int hidden, const float *w_lin, float *dw_lin, float *dh, hipStream_t st);
int theta_bwd_flat(const float *x, int n, int c, int hidden, const float *w_flat, const float *dz1, float *dw_flat, float *dx,
                   int dx_accumulate, hipStream_t st);
int wino_k3s1_launch(const pws_conv_args *a, const ProfHint &ph, hipStream_t st);  // conv_wino.hip
// conv_wring.hip: Winograd on the persistent LDS ring (F(2x2,3x3) for the 3x3 stride-1 kinds, F(2x2,2x2) per parity class for the
// transposed k4 s2 kind); weights = pws_conv_args.w_wring (pws_pack_conv_weight_wring).
int wring_try(const pws_conv_args *a, const ProfHint &ph, hipStream_t st);   // 1: not covered
int wring_pack(const float *pk, float *ur, int cin_pad, int cout, int ct4, hipStream_t st);
// Ring layout [32-channel block][16-channel chunk][component xi][nt][kq][co16][st]: co = 32 block + 2 co16 + nt,
// ci = 16 chunk + 4 kq + st -- one (block, chunk) is NC x 2 KB contiguous, and inside it a 1 KB run is one lane-linear B-operand
// image of v_mfma_f32_16x16x4_f32 (lane = kq * 16 + co16 supplies the k-steps st = 0..3 of its channel).
__host__ __device__ inline size_t wring_index(int nc, int xi, int ci, int co, int nchunks) {
    const int cob = co >> 5, c32 = co & 31, co16 = c32 >> 1, nt = c32 & 1;
    const int chunk = ci >> 4, kq = (ci >> 2) & 3, st = ci & 3;
    return ((((((size_t)cob * nchunks + chunk) * nc + xi) * 2 + nt) * 4 + kq) * 16 + co16) * 4 + st;
}
#ifdef __HIPCC__
// element i = ci * cout + co of a layer's ring-layout Winograd weights from its packed weights pk[plane index][cin_pad][cout]
__device__ inline void wring_pack_element(const float *__restrict__ pk, float *__restrict__ ur, size_t plane, size_t i, int cin_pad, int cout,
                                          int ct4) {
    const int co = (int)(i % cout), ci = (int)(i / cout), nch = cin_pad / 16;
    if (!ct4) {   // F(2x2,3x3): U = G g G^T, G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g[r][s] = pk[(size_t)(r * 3 + s) * plane + i];
        float u[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            u[0][s] = g[0][s], u[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]), u[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            u[3][s] = g[2][s];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ur[wring_index(16, r * 4 + 0, ci, co, nch)] = u[r][0];
            ur[wring_index(16, r * 4 + 1, ci, co, nch)] = 0.5f * (u[r][0] + u[r][1] + u[r][2]);
            ur[wring_index(16, r * 4 + 2, ci, co, nch)] = 0.5f * (u[r][0] - u[r][1] + u[r][2]);
            ur[wring_index(16, r * 4 + 3, ci, co, nch)] = u[r][2];
        }
    } else if (ct4 == 2) {   // F(2x2,5x5) of the first layer (conv_first_wino.hip), points {0, 1, -1, 2, -1/2, inf}: U = G g G^T in double, rounded once;
        // layout [k-step = ci / 4][component / 4][channel block co / 16][kq = ci % 4][co % 16][component % 4]: a k-step is 36 KB, a lane's A operands of four components 16 bytes
        const double G[6][5] = {{1, 0, 0, 0, 0},
                                {-1.0 / 3, -1.0 / 3, -1.0 / 3, -1.0 / 3, -1.0 / 3},
                                {1.0 / 3, -1.0 / 3, 1.0 / 3, -1.0 / 3, 1.0 / 3},
                                {1.0 / 15, 2.0 / 15, 4.0 / 15, 8.0 / 15, 16.0 / 15},
                                {-16.0 / 15, 8.0 / 15, -4.0 / 15, 2.0 / 15, -1.0 / 15},
                                {0, 0, 0, 0, 1}};
        double t[6][5];
        for (int r = 0; r < 6; ++r)
            for (int v = 0; v < 5; ++v) {
                double a = 0;
                for (int q = 0; q < 5; ++q) a += G[r][q] * (double)pk[(size_t)(q * 5 + v) * plane + i];
                t[r][v] = a;
            }
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                double a = 0;
                for (int v = 0; v < 5; ++v) a += t[r][v] * G[c][v];
                ur[((((size_t)((ci >> 2) * 9 + ((r * 6 + c) >> 2)) * (cout / 16) + (co >> 4)) * 4 + (ci & 3)) * 16 + (co & 15)) * 4 + ((r * 6 + c) & 3)] = (float)a;
            }
    } else {      // F(2x2,2x2) per parity class: G = [[1,0],[1,1],[0,1]]; [py][... 18 components = (px, i, j) ...]
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float g[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s) g[r][s] = pk[(size_t)((py * 2 + px) * 4 + r * 2 + s) * plane + i];
                float u[3][2];
#pragma unroll
                for (int s = 0; s < 2; ++s) u[0][s] = g[0][s], u[1][s] = g[0][s] + g[1][s], u[2][s] = g[1][s];
                float *o = ur + (size_t)py * 18 * plane;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    o[wring_index(18, px * 9 + r * 3 + 0, ci, co, nch)] = u[r][0];
                    o[wring_index(18, px * 9 + r * 3 + 1, ci, co, nch)] = u[r][0] + u[r][1];
                    o[wring_index(18, px * 9 + r * 3 + 2, ci, co, nch)] = u[r][1];
                }
            }
    }
}
#endif

// bf16 operand helpers (conv_bf16.hip, wgrad_bf16.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two floats -> two bf16 (round to nearest even) in one dword: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}

// 4 consecutive elements of an fp32 or bf16 (IO16) tensor; `idx` counts elements and is a multiple of 4
template <bool IO16>
__device__ __forceinline__ float4 ld4(const float *base, size_t idx) {
    if constexpr (IO16) {
        const uint2 u = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __bf16 *>(base) + idx);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4 *>(base + idx);
    }
}
template <bool IO16>
__device__ __forceinline__ void st4(float *base, size_t idx, const float4 &v) {
    if constexpr (IO16) {
        *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(base) + idx) = make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
    } else {
        *reinterpret_cast<float4 *>(base + idx) = v;
    }
}

// Branch-free on purpose: with `act` a kernel argument hipcc turned the three-way if into scalar compares and branches PER VALUE
// (s_cmp / s_cbranch x 3 around every element of an epilogue).  Same results as
//     LRELU: v > 0 ? v : 0.2 v      RELU: v > 0 ? v : 0      NONE: v
// for every input, NaN and infinities included, in three instructions (mul, and, max) with no VCC round trip:
// max(v, t) with t = 0.2 v / +0 / v picks v for v > 0 and t otherwise (0.2 v > v for v < 0), and a NaN v gives NaN, 0, NaN.
__device__ __forceinline__ float act_apply(float v, int act) {
    const float slope = act == PWS_ACT_LRELU ? 0.2f : 1.f;
    const unsigned keep = act == PWS_ACT_RELU ? 0u : 0xffffffffu;
    const float t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, slope * v) & keep);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));   // the builtin would canonicalise t first (one more instruction)
    return r;
}

}  // namespace pws

// Training objective around the hot path (HBM-bound, fp32 data, fp64 sums): the per-step losses the reference's
// train() wraps round netG + grid_sample (reference main_new.py:101-118,184-212; lib/utils.py:246-255,339-362,
// 405-447), as fused forward / backward kernels.
//
//   u8_normalize            images.float()*(1/255)*2-1                                     lib/utils.py:247
//   warp_norm fwd/bwd       fake = grid_sample((rgb+1)*127.5, field)/127.5-1  + sum|stable-fake|   main_new.py:106-107, lib/utils.py:349
//   temporal_l1 fwd/bwd     sum|grid_sample(fake2, affine_grid(adjacent)) - fake1|          main_new.py:195-198
//   feature_loss fwd/bwd    sum_k |unstable_k - field[stable_k]|^2                          lib/utils.py:341-345
//   field_smoothness        sum|d field/dx|, sum|d field/dy| (monitor only)                 lib/utils.py:351-357
//   shape_loss fwd/bwd      fp64 L1 residual of a per-block bilinear least-squares fit      lib/utils.py:405-425
//
// Batched layout: the reference runs the generator twice per step (item, item one frame later); without BatchNorm the two
// forwards are one batch of 2n samples, so every tensor here carries m = 2n samples, branch 1 first.
// Sums: every kernel reduces in fp32 per lane, fp64 per workgroup, and adds ONE double per workgroup into one of
// PWS_OBJ_SLOTS slots of its quantity (hardware f64 atomics; 64 slots keep same-address contention off the L2 atomics
// units); pws_objective_finalize adds the slots up in order.  The order of the per-slot additions is not fixed, which
// moves a sum by ~1e-16 relative: invisible after the final conversion to fp32.
#include "common.h"

namespace pws {

constexpr int kSlots = PWS_OBJ_SLOTS;

struct __attribute__((packed, aligned(4))) F2Uo {
    float x, y;
};

__device__ __forceinline__ float unnorm_o(float g, int size) {  // align_corners=False (what the reference's torch runs)
    return fmaf(g, 0.5f * (float)size, 0.5f * (float)(size - 1));
}

// workgroup sum of a per-lane float -> one f64 atomic per workgroup
template <int NV>
__device__ __forceinline__ void block_accumulate(const float (&v)[NV], double *const (&dst)[NV]) {
    __shared__ double red[NV][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double d = (double)v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) red[k][wave] = d;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int nw = (blockDim.x + 63) >> 6;
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[threadIdx.x][w];
        unsafeAtomicAdd(dst[threadIdx.x] + (blockIdx.x % kSlots), s);
    }
}

// the same for per-lane DOUBLES (kernels whose workgroups walk several blocks: fp32 per block, f64 across blocks)
template <int NV>
__device__ __forceinline__ void block_accumulate_d(const double (&v)[NV], double *const (&dst)[NV]) {
    __shared__ double red[NV][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double d = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) red[k][wave] = d;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int nw = (blockDim.x + 63) >> 6;
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[threadIdx.x][w];
        unsafeAtomicAdd(dst[threadIdx.x] + (blockIdx.x % kSlots), s);
    }
}
constexpr unsigned kLossGrid = 2048;   // workgroups of a capped loss launch (8 per CU): 2048 f64 atomics per quantity instead of one per 256 lanes

// ------------------------------------------------------------------------------------------------ pre-processing
__global__ void __launch_bounds__(256) u8_normalize_kernel(const unsigned char *__restrict__ src, size_t src_nstride,
                                                           float *__restrict__ dst, size_t dst_nstride, size_t per4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // group of 4 bytes inside one sample
    if (i >= per4) return;
    const unsigned u = *reinterpret_cast<const unsigned *>(src + (size_t)blockIdx.y * src_nstride + i * 4);
    float4 o;
    o.x = (float)(u & 0xff) * (1.f / 255) * 2 - 1;   // the reference's operation order: x * (1/255) * 2 - 1
    o.y = (float)((u >> 8) & 0xff) * (1.f / 255) * 2 - 1;
    o.z = (float)((u >> 16) & 0xff) * (1.f / 255) * 2 - 1;
    o.w = (float)(u >> 24) * (1.f / 255) * 2 - 1;
    *reinterpret_cast<float4 *>(dst + (size_t)blockIdx.y * dst_nstride + i * 4) = o;
}

// ------------------------------------------------------------------------------------------------ warp + L1
struct TapsO {
    int o0, o1;
    float a0, b0, a1, b1;
};
__device__ __forceinline__ TapsO make_taps_o(float gx, float gy, int H, int W) {
    const float ix = unnorm_o(gx, W), iy = unnorm_o(gy, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int xs = min(max(x0, 0), W - 2);
    const int sel = x0 - xs;
    const float wl = sel == 0 ? (vx0 ? wx0 : 0.f) : (sel == -1 ? (vx1 ? wx1 : 0.f) : 0.f);
    const float wr = sel == 0 ? (vx1 ? wx1 : 0.f) : (sel == 1 ? (vx0 ? wx0 : 0.f) : 0.f);
    const float r0 = vy0 ? wy0 : 0.f, r1 = vy1 ? wy1 : 0.f;
    TapsO t;
    t.o0 = min(max(y0, 0), H - 1) * W + xs, t.o1 = min(max(y1, 0), H - 1) * W + xs;
    t.a0 = wl * r0, t.b0 = wr * r0, t.a1 = wl * r1, t.b1 = wr * r1;
    return t;
}

// 4 pixels per lane: float4 field reads, float4 plane stores, paired 8-byte taps (as grid_sample_fwd2_kernel)
__global__ void __launch_bounds__(256) warp_norm_fwd_kernel(const float *__restrict__ src, size_t src_nstride,
                                                            const float *__restrict__ grid, float *__restrict__ fake,
                                                            const float *__restrict__ target, size_t tgt_nstride,
                                                            double *__restrict__ l1_slots, int H, int W, size_t total_groups,
                                                            unsigned nblocks) {
    const int HW = H * W;
    double sum[1] = {0.0};
    // (a workgroup walks several blocks of 256 lanes when the grid is capped -- kLossGrid -- so that the f64 atomics on the 64 slots stay few:
    //  at 256 samples one atomic per 256-lane block was 65 536 atomics on four cache lines and cost more than the kernel's own traffic)
    for (unsigned b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const unsigned blk = xcd_remap(b, nblocks);
        const size_t gidx = (size_t)blk * 256 + threadIdx.x;
        if (gidx >= total_groups) continue;
        float acc[1] = {0.f};
        const size_t p0 = gidx * 4;
        const int n = (int)(p0 / HW), hw = (int)(p0 % HW);
        const float4 ga = *reinterpret_cast<const float4 *>(grid + p0 * 2);
        const float4 gb = *reinterpret_cast<const float4 *>(grid + p0 * 2 + 4);
        TapsO t[4] = {make_taps_o(ga.x, ga.y, H, W), make_taps_o(ga.z, ga.w, H, W), make_taps_o(gb.x, gb.y, H, W),
                      make_taps_o(gb.z, gb.w, H, W)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *ip = src + (size_t)n * src_nstride + (size_t)c * HW;
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const F2Uo u = *reinterpret_cast<const F2Uo *>(ip + t[i].o0);
                const F2Uo v = *reinterpret_cast<const F2Uo *>(ip + t[i].o1);
                const float s = ((u.x + 1.f) * 127.5f) * t[i].a0 + ((u.y + 1.f) * 127.5f) * t[i].b0 +
                                ((v.x + 1.f) * 127.5f) * t[i].a1 + ((v.y + 1.f) * 127.5f) * t[i].b1;
                r[i] = s / 127.5f - 1.f;
            }
            *reinterpret_cast<float4 *>(fake + ((size_t)n * 3 + c) * HW + hw) = make_float4(r[0], r[1], r[2], r[3]);
            if (target) {
                const float4 tg = *reinterpret_cast<const float4 *>(target + (size_t)n * tgt_nstride + (size_t)c * HW + hw);
                acc[0] += fabsf(tg.x - r[0]) + fabsf(tg.y - r[1]) + fabsf(tg.z - r[2]) + fabsf(tg.w - r[3]);
            }
        }
        sum[0] += (double)acc[0];
    }
    if (l1_slots) {
        double *const dst[1] = {l1_slots};
        block_accumulate_d<1>(sum, dst);
    }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// one lane per pixel: gfake = c_l1*sign(fake - target) + gextra ; ggrid = (d fake / d field)^T gfake
__global__ void __launch_bounds__(256) warp_norm_bwd_kernel(const float *__restrict__ src, size_t src_nstride,
                                                            const float *__restrict__ grid,
                                                            const float *__restrict__ target, size_t tgt_nstride, float c_l1,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ gextra, float *__restrict__ ggrid,
                                                            int accumulate, int H, int W, size_t total, unsigned nblocks) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t p = (size_t)blk * 256 + threadIdx.x;
    if (p >= total) return;
    const int HW = H * W;
    const int n = (int)(p / HW), hw = (int)(p % HW);
    const float2 g = *reinterpret_cast<const float2 *>(grid + p * 2);
    if (scale) c_l1 *= *scale;
    const float ix = unnorm_o(g.x, W), iy = unnorm_o(g.y, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *ip = src + (size_t)n * src_nstride + (size_t)c * HW;
        const float v00 = (vy0 && vx0) ? (ip[cy0 * W + cx0] + 1.f) * 127.5f : 0.f;
        const float v01 = (vy0 && vx1) ? (ip[cy0 * W + cx1] + 1.f) * 127.5f : 0.f;
        const float v10 = (vy1 && vx0) ? (ip[cy1 * W + cx0] + 1.f) * 127.5f : 0.f;
        const float v11 = (vy1 && vx1) ? (ip[cy1 * W + cx1] + 1.f) * 127.5f : 0.f;
        float gf = gextra ? gextra[((size_t)n * 3 + c) * HW + hw] : 0.f;
        if (target) {
            const float fk = (v00 * (wx0 * wy0) + v01 * (wx1 * wy0) + v10 * (wx0 * wy1) + v11 * (wx1 * wy1)) / 127.5f - 1.f;
            gf += c_l1 * sgn(fk - target[(size_t)n * tgt_nstride + (size_t)c * HW + hw]);
        }
        gix += gf * ((v01 - v00) * wy0 + (v11 - v10) * wy1);
        giy += gf * ((v10 - v00) * wx0 + (v11 - v01) * wx1);
    }
    const float sx = 0.5f * (float)W / 127.5f, sy = 0.5f * (float)H / 127.5f;
    float2 o = make_float2(gix * sx, giy * sy);
    float2 *q = reinterpret_cast<float2 *>(ggrid + p * 2);
    if (accumulate) {
        const float2 old = *q;
        o.x += old.x, o.y += old.y;
    }
    *q = o;
}

// 4 pixels per lane (round 4): float4 reads of the field / the upstream gradient / the target planes, a float4 read-modify-write of the result,
// the two taps of a source row as ONE unaligned 8-byte load (as warp_norm_fwd_kernel) -- 24 loads of 8 bytes per lane where the kernel above
// issues 48 scalar gathers, a third of its streaming instructions.  Same arithmetic per pixel.
__global__ void __launch_bounds__(256) warp_norm_bwd4_kernel(const float *__restrict__ src, size_t src_nstride, const float *__restrict__ grid,
                                                             const float *__restrict__ target, size_t tgt_nstride, float c_l1,
                                                             const float *__restrict__ scale, const float *__restrict__ gextra,
                                                             float *__restrict__ ggrid, int accumulate, int H, int W, size_t total_groups,
                                                             unsigned nblocks) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx = (size_t)blk * 256 + threadIdx.x;
    if (gidx >= total_groups) return;
    const int HW = H * W;
    const size_t p0 = gidx * 4;
    const int n = (int)(p0 / HW), hw = (int)(p0 % HW);
    if (scale) c_l1 *= *scale;
    const float4 ga = *reinterpret_cast<const float4 *>(grid + p0 * 2), gb = *reinterpret_cast<const float4 *>(grid + p0 * 2 + 4);
    const float gxy[4][2] = {{ga.x, ga.y}, {ga.z, ga.w}, {gb.x, gb.y}, {gb.z, gb.w}};
    int o0[4], o1[4], sel[4];
    float wx0[4], wx1[4], wy0[4], wy1[4];
    bool vx0[4], vx1[4], vy0[4], vy1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float ix = unnorm_o(gxy[i][0], W), iy = unnorm_o(gxy[i][1], H);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        wx1[i] = ix - fx, wx0[i] = 1.f - wx1[i], wy1[i] = iy - fy, wy0[i] = 1.f - wy1[i];
        vx0[i] = x0 >= 0 && x0 < W, vx1[i] = x1 >= 0 && x1 < W, vy0[i] = y0 >= 0 && y0 < H, vy1[i] = y1 >= 0 && y1 < H;
        const int xs = min(max(x0, 0), W - 2);   // the pair (xs, xs + 1) holds whichever of x0 / x1 lie inside the row
        sel[i] = x0 - xs;                        // 0: (x0, x1) = the pair; -1: x1 = pair.x (x0 = -1); 1: x0 = pair.y (x1 = W); else both outside
        o0[i] = min(max(y0, 0), H - 1) * W + xs, o1[i] = min(max(y1, 0), H - 1) * W + xs;
    }
    float gix[4] = {0.f, 0.f, 0.f, 0.f}, giy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *ip = src + (size_t)n * src_nstride + (size_t)c * HW;
        float4 ge = make_float4(0.f, 0.f, 0.f, 0.f), tg = ge;
        if (gextra) ge = *reinterpret_cast<const float4 *>(gextra + ((size_t)n * 3 + c) * HW + hw);
        if (target) tg = *reinterpret_cast<const float4 *>(target + (size_t)n * tgt_nstride + (size_t)c * HW + hw);
        const float gev[4] = {ge.x, ge.y, ge.z, ge.w}, tgv[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const F2Uo u = *reinterpret_cast<const F2Uo *>(ip + o0[i]);
            const F2Uo v = *reinterpret_cast<const F2Uo *>(ip + o1[i]);
            const float l0 = sel[i] == 0 ? u.x : (sel[i] == 1 ? u.y : 0.f), r0 = sel[i] == 0 ? u.y : (sel[i] == -1 ? u.x : 0.f);
            const float l1 = sel[i] == 0 ? v.x : (sel[i] == 1 ? v.y : 0.f), r1 = sel[i] == 0 ? v.y : (sel[i] == -1 ? v.x : 0.f);
            const float v00 = (vy0[i] && vx0[i]) ? (l0 + 1.f) * 127.5f : 0.f, v01 = (vy0[i] && vx1[i]) ? (r0 + 1.f) * 127.5f : 0.f;
            const float v10 = (vy1[i] && vx0[i]) ? (l1 + 1.f) * 127.5f : 0.f, v11 = (vy1[i] && vx1[i]) ? (r1 + 1.f) * 127.5f : 0.f;
            float gf = gev[i];
            if (target) {
                const float fk = (v00 * (wx0[i] * wy0[i]) + v01 * (wx1[i] * wy0[i]) + v10 * (wx0[i] * wy1[i]) + v11 * (wx1[i] * wy1[i])) / 127.5f - 1.f;
                gf += c_l1 * sgn(fk - tgv[i]);
            }
            gix[i] += gf * ((v01 - v00) * wy0[i] + (v11 - v10) * wy1[i]);
            giy[i] += gf * ((v10 - v00) * wx0[i] + (v11 - v01) * wx1[i]);
        }
    }
    const float sx = 0.5f * (float)W / 127.5f, sy = 0.5f * (float)H / 127.5f;
    float4 oa = make_float4(gix[0] * sx, giy[0] * sy, gix[1] * sx, giy[1] * sy), ob = make_float4(gix[2] * sx, giy[2] * sy, gix[3] * sx, giy[3] * sy);
    float4 *q = reinterpret_cast<float4 *>(ggrid + p0 * 2);
    if (accumulate) {
        const float4 a = q[0], b = q[1];
        oa.x += a.x, oa.y += a.y, oa.z += a.z, oa.w += a.w, ob.x += b.x, ob.y += b.y, ob.z += b.z, ob.w += b.w;
    }
    q[0] = oa, q[1] = ob;
}

// ------------------------------------------------------------------------------------------------ temporal consistency
__device__ __forceinline__ float base_o(int j, int size) { return (2.f * j + 1.f) / (float)size - 1.f; }

struct Taps4 {
    int o00, o01, o10, o11;
    float w00, w01, w10, w11;
};
__device__ __forceinline__ Taps4 make_taps4(float gx, float gy, int H, int W) {
    const float ix = unnorm_o(gx, W), iy = unnorm_o(gy, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    Taps4 t;
    t.o00 = cy0 * W + cx0, t.o01 = cy0 * W + cx1, t.o10 = cy1 * W + cx0, t.o11 = cy1 * W + cx1;
    t.w00 = (vy0 && vx0) ? wx0 * wy0 : 0.f, t.w01 = (vy0 && vx1) ? wx1 * wy0 : 0.f;
    t.w10 = (vy1 && vx0) ? wx0 * wy1 : 0.f, t.w11 = (vy1 && vx1) ? wx1 * wy1 : 0.f;
    return t;
}

// BWD=false: sum |o21 - fake1| ; BWD=true: gfake1 += -c*sign(d) (plain), gfake2 += c*sign(d)*weights (atomics)
template <bool BWD>
__global__ void __launch_bounds__(256) temporal_l1_kernel(const float *__restrict__ fake1, const float *__restrict__ fake2,
                                                          const float *__restrict__ theta, double *__restrict__ slots, float c,
                                                          const float *__restrict__ scale, float *__restrict__ gfake1, float *__restrict__ gfake2, int H, int W,
                                                          size_t total, unsigned nblocks, float *__restrict__ sbuf = nullptr, int g_pair = 0) {
    const int HW = H * W;
    double sum[1] = {0.0};
    if constexpr (BWD) {
        if (scale) c *= *scale;
    }
    for (unsigned b = blockIdx.x; b < nblocks; b += gridDim.x) {   // (forward: capped grid, see warp_norm_fwd_kernel)
        const unsigned blk = xcd_remap(b, nblocks);
        size_t p = (size_t)blk * 256 + threadIdx.x;
        if (BWD && g_pair == 2) {
            // a workgroup = 128 columns x 2 rows, a wave = 32 columns x 2 rows (lanes l and l + 32 are vertical neighbours)
            const unsigned tiles_x = (unsigned)W / 128u, per_sample = tiles_x * ((unsigned)H / 2u);
            const unsigned ns = blk / per_sample, tb = blk % per_sample;
            const unsigned row = (tb / tiles_x) * 2u + ((threadIdx.x >> 5) & 1u), col = (tb % tiles_x) * 128u + (threadIdx.x >> 6) * 32u + (threadIdx.x & 31u);
            p = (size_t)ns * HW + (size_t)row * W + col;
        }
        if (p >= total) continue;
        float acc[1] = {0.f};
        const int n = (int)(p / HW), hw = (int)(p % HW);
        const int y_ = hw / W, x_ = hw % W;
        const float *th = theta + (size_t)n * 6;
        const float bx = base_o(x_, W), by = base_o(y_, H);
        const Taps4 t = make_taps4(th[0] * bx + th[1] * by + th[2], th[3] * bx + th[4] * by + th[5], H, W);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t pl = ((size_t)n * 3 + ch) * HW;
            const float *ip = fake2 + pl;
            const float o21 = ip[t.o00] * t.w00 + ip[t.o01] * t.w01 + ip[t.o10] * t.w10 + ip[t.o11] * t.w11;
            const float d = o21 - fake1[pl + hw];
            if constexpr (BWD) {
                const float s = c * sgn(d);
                gfake1[pl + hw] -= s;
                if (sbuf) {   // deterministic mode: the scatter is done as an ordered gather by temporal_gather_kernel
                    sbuf[pl + hw] = s;
                } else if (g_pair) {
                    // (round 4) the kernel runs at the L2's atomic rate (12 per pixel: ~230 G/s), and for an affine map close to the identity
                    // the RIGHT taps of a pixel are the LEFT taps of its right-hand neighbour = the next lane: a lane whose neighbour has
                    // the same addresses hands its two right-tap values over (wave shuffle) and the neighbour adds them to its own -- 4
                    // atomics per pixel and plane become ~2.  Every lane of the wave takes part in the shuffles (no divergence here).
                    const float v00 = s * t.w00, v01 = s * t.w01, v10 = s * t.w10, v11 = s * t.w11;
                    const int lane = threadIdx.x & 63;
                    // (g_pair is only set when total % 256 == 0: every lane of every wave is here)
                    const int rowlen = g_pair == 2 ? 32 : 64, li = lane & (rowlen - 1);   // lanes of one image row inside the wave
                    const bool nxt_same = li < rowlen - 1 && __shfl_down(t.o00, 1, 64) == t.o01 && __shfl_down(t.o10, 1, 64) == t.o11;   // my right taps == the next lane's left taps
                    const bool prv_same = __shfl_up(nxt_same ? 1 : 0, 1, 64) == 1 && li > 0;
                    const float a01 = __shfl_up(v01, 1, 64), a11 = __shfl_up(v11, 1, 64);
                    float u00 = v00 + (prv_same ? a01 : 0.f), u10 = v10 + (prv_same ? a11 : 0.f);   // left taps, the previous lane's right taps folded in
                    float r01 = nxt_same ? 0.f : v01, r11 = nxt_same ? 0.f : v11;                    // right taps, unless the next lane takes them
                    bool lower_out = true;   // this lane issues its bottom-row taps itself
                    if (g_pair == 2) {
                        // ... and the BOTTOM taps of a pixel are the TOP taps of the pixel below = lane + 32
                        const bool ver_same = lane < 32 && __shfl_down(t.o00, 32, 64) == t.o10 && __shfl_down(t.o01, 32, 64) == t.o11;
                        const bool from_up = __shfl_up(ver_same ? 1 : 0, 32, 64) == 1 && lane >= 32;
                        const float c_in = __shfl_up(u10, 32, 64), d_in = __shfl_up(r11, 32, 64);
                        if (from_up) u00 += c_in, r01 += d_in;
                        lower_out = !ver_same;
                    }
                    float *gp = gfake2 + pl;
                    if (u00 != 0.f) atomicAdd(gp + t.o00, u00);
                    if (r01 != 0.f) atomicAdd(gp + t.o01, r01);
                    if (lower_out) {
                        if (u10 != 0.f) atomicAdd(gp + t.o10, u10);
                        if (r11 != 0.f) atomicAdd(gp + t.o11, r11);
                    }
                } else if (s != 0.f) {
                    float *gp = gfake2 + pl;
                    if (t.w00 != 0.f) atomicAdd(gp + t.o00, s * t.w00);
                    if (t.w01 != 0.f) atomicAdd(gp + t.o01, s * t.w01);
                    if (t.w10 != 0.f) atomicAdd(gp + t.o10, s * t.w10);
                    if (t.w11 != 0.f) atomicAdd(gp + t.o11, s * t.w11);
                }
            } else {
                acc[0] += fabsf(d);
            }
        }
        sum[0] += (double)acc[0];
    }
    if constexpr (!BWD) {
        double *const dst[1] = {slots};
        block_accumulate_d<1>(sum, dst);
    }
}

// Backward of the temporal term on 32 x 32-pixel tiles (round 4).  The warp is AFFINE, so the four taps of a tile's pixels fall into
// the bounding box of the tile's image (+ 1 pixel): the workgroup scatters into an LDS copy of that box (ds_add_f32) and adds every
// box cell to gfake2 ONCE.  temporal_l1_kernel<true> sends 12 fp32 atomics per pixel to memory (25 M per launch at 32 x 256 x 256:
// 135 us, 3 launches per training step); here it is ~3.4 per pixel (a 34 x 34 box x 3 planes per 1024 pixels).  A tile whose box
// does not fit the LDS (strong zoom-out / shear in feature_adjacent) takes the per-pixel global atomics -- block-uniform.
// MEASURED SLOWER than temporal_l1_kernel<true> and not taken by the product (see pws_temporal_l1_bwd); kept for the A/B.
constexpr int kTT = 32;                 // tile edge
constexpr int kTBoxCells = 44 * 44;     // LDS box: 3 planes x 1936 floats = 23 KB (a near-identity map needs 35 x 35): 6 workgroups per CU
__global__ void __launch_bounds__(256) temporal_l1_bwd_tiled_kernel(const float *__restrict__ fake1, const float *__restrict__ fake2,
                                                                    const float *__restrict__ theta, float c, const float *__restrict__ scale,
                                                                    float *__restrict__ gfake1, float *__restrict__ gfake2, int H, int W, int tiles_x,
                                                                    int tiles_y) {
    __shared__ float box[3][kTBoxCells];
    const int tile = blockIdx.x % (tiles_x * tiles_y), n = blockIdx.x / (tiles_x * tiles_y);
    const int px0 = (tile % tiles_x) * kTT, py0 = (tile / tiles_x) * kTT;
    const int HW = H * W;
    if (scale) c *= *scale;
    const float *th = theta + (size_t)n * 6;
    // bounding box (in clamped source pixels) of the taps of this tile: the sample coordinate is affine in the pixel, so its extremes
    // over the tile are at the tile's corners; taps are floor(.) and floor(.) + 1, clamped to the image as make_taps4 clamps them
    float ixmin = 3.0e38f, ixmax = -3.0e38f, iymin = 3.0e38f, iymax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = min(px0 + ((k & 1) ? kTT - 1 : 0), W - 1), cy = min(py0 + ((k & 2) ? kTT - 1 : 0), H - 1);
        const float bx = base_o(cx, W), by = base_o(cy, H);
        const float ix = unnorm_o(th[0] * bx + th[1] * by + th[2], W), iy = unnorm_o(th[3] * bx + th[4] * by + th[5], H);
        ixmin = fminf(ixmin, ix), ixmax = fmaxf(ixmax, ix), iymin = fminf(iymin, iy), iymax = fmaxf(iymax, iy);
    }
    // (one pixel of slack either side: the corner values and the per-pixel values round differently)
    const bool finite = ixmin > -1.0e9f && ixmax < 1.0e9f && iymin > -1.0e9f && iymax < 1.0e9f;   // (NaN compares false)
    const int x_lo = finite ? min(max((int)floorf(ixmin) - 1, 0), W - 1) : 0, x_hi = finite ? min(max((int)floorf(ixmax) + 2, 0), W - 1) : W - 1;
    const int y_lo = finite ? min(max((int)floorf(iymin) - 1, 0), H - 1) : 0, y_hi = finite ? min(max((int)floorf(iymax) + 2, 0), H - 1) : H - 1;
    const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
    const bool in_lds = finite && bw * bh <= kTBoxCells;   // block-uniform
    if (in_lds) {
        for (int i = threadIdx.x; i < bw * bh; i += 256) box[0][i] = 0.f, box[1][i] = 0.f, box[2][i] = 0.f;
        __syncthreads();
    }
    // thread t: row t / 8 of the tile, 4 consecutive pixels
    const int y_ = py0 + (int)(threadIdx.x >> 3), xq = px0 + (int)(threadIdx.x & 7) * 4;
    if (y_ < H && xq < W) {
        const bool vec = xq + 3 < W && (W & 3) == 0;
        float f1[3][4], g1[3][4];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t o = ((size_t)n * 3 + ch) * HW + (size_t)y_ * W + xq;
            if (vec) {
                const float4 a = *reinterpret_cast<const float4 *>(fake1 + o), b = *reinterpret_cast<const float4 *>(gfake1 + o);
                f1[ch][0] = a.x, f1[ch][1] = a.y, f1[ch][2] = a.z, f1[ch][3] = a.w, g1[ch][0] = b.x, g1[ch][1] = b.y, g1[ch][2] = b.z, g1[ch][3] = b.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) f1[ch][i] = xq + i < W ? fake1[o + i] : 0.f, g1[ch][i] = xq + i < W ? gfake1[o + i] : 0.f;
            }
        }
        const float by = base_o(y_, H);
        // the four pixels' taps first, then all 48 gathers of fake2 back to back (one lane's loads are independent: the kernel is
        // bound by their latency, not by bytes), then the scatter
        Taps4 tp[4];
        int cell[4][4];
        bool live[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            live[i] = xq + i < W;
            const float bx = base_o(min(xq + i, W - 1), W);
            tp[i] = make_taps4(th[0] * bx + th[1] * by + th[2], th[3] * bx + th[4] * by + th[5], H, W);
            cell[i][0] = cell[i][1] = cell[i][2] = cell[i][3] = 0;
            if (in_lds) {   // the taps' cells inside the box (it holds them by construction; the clamps only guard that reasoning)
                const int r0 = tp[i].o00 / W, q0 = tp[i].o00 - r0 * W, r1 = tp[i].o11 / W, q1 = tp[i].o11 - r1 * W;   // (cy0, cx0), (cy1, cx1)
                const int a0 = min(max(r0 - y_lo, 0), bh - 1), a1 = min(max(r1 - y_lo, 0), bh - 1);
                const int b0 = min(max(q0 - x_lo, 0), bw - 1), b1 = min(max(q1 - x_lo, 0), bw - 1);
                cell[i][0] = a0 * bw + b0, cell[i][1] = a0 * bw + b1, cell[i][2] = a1 * bw + b0, cell[i][3] = a1 * bw + b1;
            }
        }
        float v[3][4][4];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float *ip = fake2 + ((size_t)n * 3 + ch) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[ch][i][0] = ip[tp[i].o00], v[ch][i][1] = ip[tp[i].o01], v[ch][i][2] = ip[tp[i].o10], v[ch][i][3] = ip[tp[i].o11];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float *gp = gfake2 + ((size_t)n * 3 + ch) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const Taps4 &t = tp[i];
                const float o21 = v[ch][i][0] * t.w00 + v[ch][i][1] * t.w01 + v[ch][i][2] * t.w10 + v[ch][i][3] * t.w11;
                const float sg = live[i] ? c * sgn(o21 - f1[ch][i]) : 0.f;
                g1[ch][i] -= sg;
                if (sg == 0.f) continue;
                if (in_lds) {
                    if (t.w00 != 0.f) atomicAdd(&box[ch][cell[i][0]], sg * t.w00);
                    if (t.w01 != 0.f) atomicAdd(&box[ch][cell[i][1]], sg * t.w01);
                    if (t.w10 != 0.f) atomicAdd(&box[ch][cell[i][2]], sg * t.w10);
                    if (t.w11 != 0.f) atomicAdd(&box[ch][cell[i][3]], sg * t.w11);
                } else {
                    if (t.w00 != 0.f) atomicAdd(gp + t.o00, sg * t.w00);
                    if (t.w01 != 0.f) atomicAdd(gp + t.o01, sg * t.w01);
                    if (t.w10 != 0.f) atomicAdd(gp + t.o10, sg * t.w10);
                    if (t.w11 != 0.f) atomicAdd(gp + t.o11, sg * t.w11);
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t o = ((size_t)n * 3 + ch) * HW + (size_t)y_ * W + xq;
            if (vec) {
                *reinterpret_cast<float4 *>(gfake1 + o) = make_float4(g1[ch][0], g1[ch][1], g1[ch][2], g1[ch][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xq + i < W) gfake1[o + i] = g1[ch][i];
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < bw * bh; i += 256) {
            const size_t q = (size_t)(y_lo + i / bw) * W + (x_lo + i % bw);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float v = box[ch][i];
                if (v != 0.f) atomicAdd(gfake2 + ((size_t)n * 3 + ch) * HW + q, v);
            }
        }
    }
}

// The lean form of the tiled variant: 16 x 16 pixels per workgroup, ONE pixel per lane (few registers, a 7 KB box: many workgroups per
// CU).  PWS_OPT_EXPERIMENT 99 (A/B, tools/temporal_ab.py).
constexpr int kT16Box = 24 * 24;
__global__ void __launch_bounds__(256) temporal_l1_bwd_tile16_kernel(const float *__restrict__ fake1, const float *__restrict__ fake2,
                                                                     const float *__restrict__ theta, float c, const float *__restrict__ scale,
                                                                     float *__restrict__ gfake1, float *__restrict__ gfake2, int H, int W, int tiles_x,
                                                                     int tiles_y) {
    __shared__ float box[3][kT16Box];
    const int tile = blockIdx.x % (tiles_x * tiles_y), n = blockIdx.x / (tiles_x * tiles_y);
    const int px0 = (tile % tiles_x) * 16, py0 = (tile / tiles_x) * 16;
    const int HW = H * W;
    if (scale) c *= *scale;
    const float *th = theta + (size_t)n * 6;
    float ixmin = 3.0e38f, ixmax = -3.0e38f, iymin = 3.0e38f, iymax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = min(px0 + ((k & 1) ? 15 : 0), W - 1), cy = min(py0 + ((k & 2) ? 15 : 0), H - 1);
        const float bx = base_o(cx, W), by = base_o(cy, H);
        const float ix = unnorm_o(th[0] * bx + th[1] * by + th[2], W), iy = unnorm_o(th[3] * bx + th[4] * by + th[5], H);
        ixmin = fminf(ixmin, ix), ixmax = fmaxf(ixmax, ix), iymin = fminf(iymin, iy), iymax = fmaxf(iymax, iy);
    }
    const bool finite = ixmin > -1.0e9f && ixmax < 1.0e9f && iymin > -1.0e9f && iymax < 1.0e9f;
    const int x_lo = finite ? min(max((int)floorf(ixmin) - 1, 0), W - 1) : 0, x_hi = finite ? min(max((int)floorf(ixmax) + 2, 0), W - 1) : W - 1;
    const int y_lo = finite ? min(max((int)floorf(iymin) - 1, 0), H - 1) : 0, y_hi = finite ? min(max((int)floorf(iymax) + 2, 0), H - 1) : H - 1;
    const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
    const bool in_lds = finite && bw * bh <= kT16Box;   // block-uniform
    if (in_lds) {
        for (int i = threadIdx.x; i < bw * bh; i += 256) box[0][i] = 0.f, box[1][i] = 0.f, box[2][i] = 0.f;
        __syncthreads();
    }
    const int y_ = py0 + (int)(threadIdx.x >> 4), x_ = px0 + (int)(threadIdx.x & 15);
    if (y_ < H && x_ < W) {
        const float bx = base_o(x_, W), by = base_o(y_, H);
        const Taps4 t = make_taps4(th[0] * bx + th[1] * by + th[2], th[3] * bx + th[4] * by + th[5], H, W);
        int c00 = 0, c01 = 0, c10 = 0, c11 = 0;
        if (in_lds) {
            const int r0 = t.o00 / W, q0 = t.o00 - r0 * W, r1 = t.o11 / W, q1 = t.o11 - r1 * W;
            const int a0 = min(max(r0 - y_lo, 0), bh - 1), a1 = min(max(r1 - y_lo, 0), bh - 1);
            const int b0 = min(max(q0 - x_lo, 0), bw - 1), b1 = min(max(q1 - x_lo, 0), bw - 1);
            c00 = a0 * bw + b0, c01 = a0 * bw + b1, c10 = a1 * bw + b0, c11 = a1 * bw + b1;
        }
        const int hw = y_ * W + x_;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const size_t pl = ((size_t)n * 3 + ch) * HW;
            const float *ip = fake2 + pl;
            const float o21 = ip[t.o00] * t.w00 + ip[t.o01] * t.w01 + ip[t.o10] * t.w10 + ip[t.o11] * t.w11;
            const float sg = c * sgn(o21 - fake1[pl + hw]);
            gfake1[pl + hw] -= sg;
            if (sg == 0.f) continue;
            if (in_lds) {
                if (t.w00 != 0.f) atomicAdd(&box[ch][c00], sg * t.w00);
                if (t.w01 != 0.f) atomicAdd(&box[ch][c01], sg * t.w01);
                if (t.w10 != 0.f) atomicAdd(&box[ch][c10], sg * t.w10);
                if (t.w11 != 0.f) atomicAdd(&box[ch][c11], sg * t.w11);
            } else {
                float *gp = gfake2 + pl;
                if (t.w00 != 0.f) atomicAdd(gp + t.o00, sg * t.w00);
                if (t.w01 != 0.f) atomicAdd(gp + t.o01, sg * t.w01);
                if (t.w10 != 0.f) atomicAdd(gp + t.o10, sg * t.w10);
                if (t.w11 != 0.f) atomicAdd(gp + t.o11, sg * t.w11);
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < bw * bh; i += 256) {
            const size_t q = (size_t)(y_lo + i / bw) * W + (x_lo + i % bw);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float v = box[ch][i];
                if (v != 0.f) atomicAdd(gfake2 + ((size_t)n * 3 + ch) * HW + q, v);
            }
        }
    }
}

// Deterministic adjoint of the affine warp (pws_temporal_l1_bwd_det): gfake2[q] += sum over the output pixels p one of whose four taps
// is q of S[p] * w(p -> q), as a GATHER -- one lane per source pixel walks its candidates in row-major order, so every element is
// written by one lane in a fixed order (the scatter adds with fp32 atomics in arrival order).  The sample coordinate is affine in p:
// ix = a00 px + a01 py + b0, iy = a10 px + a11 py + b1; p is a candidate of q when |ix - qx| < 1 and |iy - qy| < 1, i.e. inside the
// parallelogram M^-1 ([-1, 1]^2) round M^-1 (q - b): its bounding box (+ a margin) is walked and every candidate's taps are
// recomputed with the forward's own arithmetic (make_taps4), so membership and weights are exactly the scatter's.
__global__ void __launch_bounds__(256) temporal_gather_kernel(const float *__restrict__ sbuf, const float *__restrict__ theta,
                                                              float *__restrict__ gfake2, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int HW = H * W;
    const int n = (int)(i / HW), q = (int)(i % HW);
    const int qy = q / W, qx = q % W;
    const float *th = theta + (size_t)n * 6;
    const float a00 = th[0], a01 = th[1] * (float)W / (float)H, a10 = th[3] * (float)H / (float)W, a11 = th[4];
    const float b0 = th[0] * (0.5f - 0.5f * W) + th[1] * (0.5f * W / H - 0.5f * W) + th[2] * 0.5f * W + 0.5f * (W - 1);
    const float b1 = th[3] * (0.5f * H / W - 0.5f * H) + th[4] * (0.5f - 0.5f * H) + th[5] * 0.5f * H + 0.5f * (H - 1);
    const float det = a00 * a11 - a01 * a10;
    int px_lo = 0, px_hi = W - 1, py_lo = 0, py_hi = H - 1;   // degenerate map: every pixel is a candidate
    if (fabsf(det) > 1e-6f) {
        const float rx = (float)qx - b0, ry = (float)qy - b1;
        const float pcx = (a11 * rx - a01 * ry) / det, pcy = (-a10 * rx + a00 * ry) / det;
        const float ex = (fabsf(a11) + fabsf(a01)) / fabsf(det) + 1.01f, ey = (fabsf(a10) + fabsf(a00)) / fabsf(det) + 1.01f;
        px_lo = max(0, (int)ceilf(pcx - ex)), px_hi = min(W - 1, (int)floorf(pcx + ex));
        py_lo = max(0, (int)ceilf(pcy - ey)), py_hi = min(H - 1, (int)floorf(pcy + ey));
    }
    float acc[3] = {0.f, 0.f, 0.f};
    for (int py = py_lo; py <= py_hi; ++py)
        for (int px = px_lo; px <= px_hi; ++px) {
            const float bx = base_o(px, W), by = base_o(py, H);
            const Taps4 t = make_taps4(th[0] * bx + th[1] * by + th[2], th[3] * bx + th[4] * by + th[5], H, W);
            // the scatter adds tap by tap (o00, o01, o10, o11): clamped taps of a border pixel may coincide, all of them count
            float wq = 0.f;
            if (t.o00 == q) wq += t.w00;
            if (t.o01 == q) wq += t.w01;
            if (t.o10 == q) wq += t.w10;
            if (t.o11 == q) wq += t.w11;
            if (wq != 0.f) {
                const size_t pp = (size_t)n * 3 * HW + (size_t)py * W + px;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) acc[ch] = fmaf(sbuf[pp + (size_t)ch * HW], wq, acc[ch]);
            }
        }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) gfake2[((size_t)n * 3 + ch) * HW + q] += acc[ch];
}

// ------------------------------------------------------------------------------------------------ feature points
// features: (m, nf, 6) = [stable x, stable y, 1, unstable x, unstable y, 1] (lib/utils.py:225, :246-254).
// index = int((coord+1)*size/2): truncation toward zero, negative indices wrap once like Python's (the reference indexes a
// tensor with them); anything still outside is an IndexError in the reference and is clamped here (host wrapper validates).
__device__ __forceinline__ int feat_index(float coord, int size) {
    int i = (int)((coord + 1.f) * (float)size / 2.f);
    if (i < 0) i += size;
    return min(max(i, 0), size - 1);
}
template <bool BWD>
__global__ void __launch_bounds__(256) feature_loss_kernel(const float *__restrict__ grid, const float *__restrict__ features,
                                                           double *__restrict__ slots, float c, const float *__restrict__ scale,
                                                           float *__restrict__ ggrid, int nf, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // (sample, point)
    float acc[1] = {0.f};
    if (i < total) {
        const int n = (int)(i / nf);
        const float *f = features + i * 6;
        const int ix = feat_index(f[0], W), iy = feat_index(f[1], H);
        const size_t q = (((size_t)n * H + iy) * W + ix) * 2;
        const float dx = f[3] - grid[q], dy = f[4] - grid[q + 1];
        if constexpr (BWD) {
            if (scale) c *= *scale;
            atomicAdd(ggrid + q, -2.f * c * dx);   // several points may share a pixel
            atomicAdd(ggrid + q + 1, -2.f * c * dy);
        } else {
            acc[0] = dx * dx + dy * dy;
        }
    }
    if constexpr (!BWD) {
        double *const dst[1] = {slots};
        block_accumulate<1>(acc, dst);
    }
}

// Deterministic variant of the backward above: one lane per SAMPLE walks its points in order (plain read-modify-write: several points
// may share a pixel, and the atomics of the kernel above arrive in any order).
__global__ void __launch_bounds__(64) feature_loss_bwd_serial_kernel(const float *__restrict__ grid, const float *__restrict__ features, float c,
                                                                    const float *__restrict__ scale, float *__restrict__ ggrid, int m, int nf,
                                                                    int H, int W) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= m) return;
    if (scale) c *= *scale;
    for (int k = 0; k < nf; ++k) {
        const float *f = features + ((size_t)n * nf + k) * 6;
        const int ix = feat_index(f[0], W), iy = feat_index(f[1], H);
        const size_t q = (((size_t)n * H + iy) * W + ix) * 2;
        const float dx = f[3] - grid[q], dy = f[4] - grid[q + 1];
        ggrid[q] += -2.f * c * dx;
        ggrid[q + 1] += -2.f * c * dy;
    }
}

// ------------------------------------------------------------------------------------------------ smoothness (monitor)
__global__ void __launch_bounds__(256) field_smoothness_kernel(const float *__restrict__ grid, double *__restrict__ slots_dx,
                                                               double *__restrict__ slots_dy, int H, int W, size_t total,
                                                               unsigned nblocks) {
    double sum[2] = {0.0, 0.0};
    for (unsigned b = blockIdx.x; b < nblocks; b += gridDim.x) {   // (capped grid, see warp_norm_fwd_kernel)
        const unsigned blk = xcd_remap(b, nblocks);
        const size_t p = (size_t)blk * 256 + threadIdx.x;
        if (p >= total) continue;
        const int hw = (int)(p % ((size_t)H * W));
        const int y_ = hw / W, x_ = hw % W;
        const float2 g = *reinterpret_cast<const float2 *>(grid + p * 2);
        if (x_ + 1 < W) {
            const float2 r = *reinterpret_cast<const float2 *>(grid + (p + 1) * 2);
            sum[0] += (double)(fabsf(g.x - r.x) + fabsf(g.y - r.y));
        }
        if (y_ + 1 < H) {
            const float2 d = *reinterpret_cast<const float2 *>(grid + (p + W) * 2);
            sum[1] += (double)(fabsf(g.x - d.x) + fabsf(g.y - d.y));
        }
    }
    double *const dst[2] = {slots_dx, slots_dy};
    block_accumulate_d<2>(sum, dst);
}

// The same sums with FOUR pixels of a row per lane (W % 4 == 0): two 16-byte loads for the lane's pixels, two for the row below, 8 bytes for the
// right-hand neighbour of its last pixel -- the one-pixel kernel's 8-byte loads kept it at 2 TB/s (68 us per launch at 256 samples, where the field is
// 134 MB).  Per lane the eight terms of a direction are summed in fp32 (exact to 1e-7 of the group), the groups in fp64 as before.
__global__ void __launch_bounds__(256) field_smoothness4_kernel(const float *__restrict__ grid, double *__restrict__ slots_dx,
                                                                double *__restrict__ slots_dy, int H, int W, size_t total_groups,
                                                                unsigned nblocks) {
    double sum[2] = {0.0, 0.0};
    const size_t HW = (size_t)H * W;
    for (unsigned b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const unsigned blk = xcd_remap(b, nblocks);
        const size_t gidx = (size_t)blk * 256 + threadIdx.x;
        if (gidx >= total_groups) continue;
        const size_t p0 = gidx * 4;
        const int hw = (int)(p0 % HW);
        const int y_ = hw / W, x_ = hw % W;
        const float4 a = *reinterpret_cast<const float4 *>(grid + p0 * 2), c = *reinterpret_cast<const float4 *>(grid + p0 * 2 + 4);
        float sx = fabsf(a.x - a.z) + fabsf(a.y - a.w) + fabsf(a.z - c.x) + fabsf(a.w - c.y) + fabsf(c.x - c.z) + fabsf(c.y - c.w);
        if (x_ + 4 < W) {
            const float2 r = *reinterpret_cast<const float2 *>(grid + (p0 + 4) * 2);
            sx += fabsf(c.z - r.x) + fabsf(c.w - r.y);
        }
        sum[0] += (double)sx;
        if (y_ + 1 < H) {
            const float4 d = *reinterpret_cast<const float4 *>(grid + (p0 + W) * 2), e = *reinterpret_cast<const float4 *>(grid + (p0 + W) * 2 + 4);
            const float sy = fabsf(a.x - d.x) + fabsf(a.y - d.y) + fabsf(a.z - d.z) + fabsf(a.w - d.w) + fabsf(c.x - e.x) + fabsf(c.y - e.y) +
                             fabsf(c.z - e.z) + fabsf(c.w - e.w);
            sum[1] += (double)sy;
        }
    }
    double *const dst[2] = {slots_dx, slots_dy};
    block_accumulate_d<2>(sum, dst);
}

// ------------------------------------------------------------------------------------------------ shape loss (fp64)
// One workgroup per bs x bs block of the residual field (bs*bs <= 1024 lanes).  The reference's basis A (bs*bs x 4,
// generate_affine_matrix) is separable: column k = a_{k/2}(y) * a_{k%2}(x), a_0(t) = (L-t)/L, a_1(t) = t/L, L = bs-1, so
// (A^T A)^-1 = G^-1 (x) G^-1 with the 2x2 Gram matrix G of {a_0, a_1}; its three distinct entries come in as arguments.
// r = B - A (A^T A)^-1 A^T B per coordinate; forward: sum|r|; backward: g = c * (s - A (A^T A)^-1 A^T s), s = sign(r).
__device__ __forceinline__ void reduce8(double (&v)[8], double (*sh)[16], int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if (lane == 0) sh[k][wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += sh[k][w];
        v[k] = s;
    }
    __syncthreads();
}

template <bool BWD>
__global__ void __launch_bounds__(1024) shape_loss_kernel(const float *__restrict__ resid, double *__restrict__ slots, double c,
                                                          const float *__restrict__ scale, float *__restrict__ gresid, int size, int nblk, int bs, double gi00,
                                                          double gi01, double gi11) {
    __shared__ double sh[8][16];
    const int b = blockIdx.x;  // (sample, block row, block col)
    const int n = b / (nblk * nblk), by = (b / nblk) % nblk, bx = b % nblk;
    const int px = threadIdx.x % bs, py = threadIdx.x / bs;
    const bool live = py < bs;
    const int nwaves = (blockDim.x + 63) >> 6;
    const double L = (double)(bs - 1);
    const double ax[2] = {(L - px) / L, px / L}, ay[2] = {(L - py) / L, py / L};
    double a[4] = {ay[0] * ax[0], ay[0] * ax[1], ay[1] * ax[0], ay[1] * ax[1]};  // Q11, Q21, Q12, Q22 (lib/utils.py:438-441)
    const size_t q = (((size_t)n * size + (size_t)by * bs + py) * size + (size_t)bx * bs + px) * 2;
    double bxy[2] = {0.0, 0.0};
    if (live) {
        const float2 r = *reinterpret_cast<const float2 *>(resid + q);
        bxy[0] = r.x, bxy[1] = r.y;
    } else {
        a[0] = a[1] = a[2] = a[3] = 0.0;
    }
    const double gi[2][2] = {{gi00, gi01}, {gi01, gi11}};
    auto project = [&](const double (&val)[2], double (&out)[2]) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[2 * k] = a[k] * val[0], t[2 * k + 1] = a[k] * val[1];
        reduce8(t, sh, nwaves);  // A^T val, all lanes hold it
        out[0] = out[1] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double cx = 0.0, cy = 0.0;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const double g = gi[k >> 1][l >> 1] * gi[k & 1][l & 1];
                cx += g * t[2 * l], cy += g * t[2 * l + 1];
            }
            out[0] += a[k] * cx, out[1] += a[k] * cy;
        }
    };
    double proj[2];
    project(bxy, proj);
    const double r0 = live ? proj[0] - bxy[0] : 0.0, r1 = live ? proj[1] - bxy[1] : 0.0;   // AB - B, as torch.dist(AB, B, 1)
    if constexpr (!BWD) {
        double t[8] = {fabs(r0) + fabs(r1), 0, 0, 0, 0, 0, 0, 0};
        reduce8(t, sh, nwaves);
        if (threadIdx.x == 0) unsafeAtomicAdd(slots + (blockIdx.x % kSlots), t[0]);
    } else {
        // d sum|PB - B| / dB = (P - I)^T s = P s - s   (P symmetric), s = sign(PB - B)
        const double s[2] = {r0 > 0 ? 1.0 : (r0 < 0 ? -1.0 : 0.0), r1 > 0 ? 1.0 : (r1 < 0 ? -1.0 : 0.0)};
        double ps[2];
        project(s, ps);
        if (scale) c *= (double)*scale;
        if (live) *reinterpret_cast<float2 *>(gresid + q) = make_float2((float)(c * (ps[0] - s[0])), (float)(c * (ps[1] - s[1])));
    }
}

// bs == 16 (the reference's configuration: 256 x 256 fields in 16 x 16 blocks), round 4: ONE WAVE per block, 4 pixels of a row per
// lane (a 32-byte load), the eight sums of A^T val by a 6-step butterfly over the wave -- no shared memory, no workgroup barrier.  The
// kernel above spends its time in two (backward: four) workgroup reductions of 8 doubles per 256 pixels: 633 us per backward at
// 64 x 256 x 256 (422 GB/s) for 67 MB of traffic.  Same arithmetic in double; the summation ORDER differs (last bits of the doubles).
template <bool BWD>
__global__ void __launch_bounds__(256) shape_loss16_kernel(const float *__restrict__ resid, double *__restrict__ slots, double c,
                                                           const float *__restrict__ scale, float *__restrict__ gresid, int size, int nblk, unsigned nblocks,
                                                           double gi00, double gi01, double gi11) {
    const int lane = threadIdx.x & 63;
    double l1_total = 0.0;   // forward: this lane's share over every block its wave walks (capped grid: few f64 atomics on the 64 slots)
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nblocks; b += gridDim.x * 4u) {   // (sample, block row, block col); whole waves
    const int n = (int)(b / (unsigned)(nblk * nblk)), by = (int)((b / (unsigned)nblk) % (unsigned)nblk), bx = (int)(b % (unsigned)nblk);
    const int py = lane >> 2, px0 = (lane & 3) * 4;
    const double L = 15.0;
    const double ay[2] = {(L - py) / L, py / L};
    double ax[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ax[i][0] = (L - (px0 + i)) / L, ax[i][1] = (px0 + i) / L;
    const size_t q = (((size_t)n * size + (size_t)by * 16 + py) * size + (size_t)bx * 16 + px0) * 2;
    const float4 v0 = *reinterpret_cast<const float4 *>(resid + q), v1 = *reinterpret_cast<const float4 *>(resid + q + 4);
    const double bxy[4][2] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
    const double gi[2][2] = {{gi00, gi01}, {gi01, gi11}};
    // out[i] = (A (A^T A)^-1 A^T val)[pixel i of this lane]
    auto project = [&](const double (&val)[4][2], double (&out)[4][2]) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // t[2 k + comp], k = 2 * (y basis) + (x basis): Q11, Q21, Q12, Q22 (lib/utils.py:438-441)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double a = ay[k >> 1] * ax[i][k & 1];
                t[2 * k] += a * val[i][0], t[2 * k + 1] += a * val[i][1];
            }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) t[k] += __shfl_xor(t[k], off, 64);
        double cx[4], cy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cx[k] = cy[k] = 0.0;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const double g = gi[k >> 1][l >> 1] * gi[k & 1][l & 1];
                cx[k] += g * t[2 * l], cy[k] += g * t[2 * l + 1];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[i][0] = out[i][1] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double a = ay[k >> 1] * ax[i][k & 1];
                out[i][0] += a * cx[k], out[i][1] += a * cy[k];
            }
        }
    };
    double proj[4][2];
    project(bxy, proj);
    if constexpr (!BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) l1_total += fabs(proj[i][0] - bxy[i][0]) + fabs(proj[i][1] - bxy[i][1]);   // AB - B, as torch.dist(AB, B, 1)
    } else {
        // d sum|PB - B| / dB = (P - I)^T s = P s - s   (P symmetric), s = sign(PB - B)
        double sg[4][2], ps[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double r = proj[i][k] - bxy[i][k];
                sg[i][k] = r > 0 ? 1.0 : (r < 0 ? -1.0 : 0.0);
            }
        project(sg, ps);
        if (scale) c *= (double)*scale;
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[2 * i] = (float)(c * (ps[i][0] - sg[i][0])), o[2 * i + 1] = (float)(c * (ps[i][1] - sg[i][1]));
        *reinterpret_cast<float4 *>(gresid + q) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(gresid + q + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    }   // blocks of this wave
    if constexpr (!BWD) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) l1_total += __shfl_xor(l1_total, off, 64);
        if (lane == 0) unsafeAtomicAdd(slots + ((blockIdx.x * 4u + (threadIdx.x >> 6)) % kSlots), l1_total);
    }
}

// ------------------------------------------------------------------------------------------------ finalize
__global__ void objective_finalize_kernel(const double *__restrict__ slots, int nq, const double *__restrict__ coef,
                                          int nout, float *__restrict__ out) {
    // out[j] = sum_q coef[j*nq + q] * (sum of quantity q's slots); one lane per output, slots added in index order
    const int j = threadIdx.x;
    if (j >= nout) return;
    double r = 0.0;
    for (int q = 0; q < nq; ++q) {
        const double cf = coef[(size_t)j * nq + q];
        if (cf == 0.0) continue;
        double s = 0.0;
        for (int k = 0; k < kSlots; ++k) s += slots[(size_t)q * kSlots + k];
        r += cf * s;
    }
    out[j] = (float)r;
}

static inline bool al16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace pws

using namespace pws;

extern "C" int pws_u8_normalize(const unsigned char *src, size_t src_nstride, float *dst, size_t dst_nstride, int m,
                                size_t per_sample, pws_stream_t stream) {
    PWS_REQUIRE(m >= 0, "pws_u8_normalize: bad shape");
    if (m == 0 || per_sample == 0) return PWS_OK;
    PWS_REQUIRE(src && dst, "pws_u8_normalize: NULL pointer");
    PWS_REQUIRE(per_sample % 4 == 0 && src_nstride % 4 == 0 && dst_nstride % 4 == 0 && (reinterpret_cast<size_t>(src) & 3) == 0 &&
                    al16(dst), "pws_u8_normalize: sizes / strides must be multiples of 4 elements, src 4-byte and dst 16-byte aligned");
    PWS_REQUIRE(m <= 65535, "pws_u8_normalize: more than 65535 samples");
    const size_t per4 = per_sample / 4;
    ProfScope prof(KID_OBJECTIVE, 3.0 * m * (double)per_sample, 5.0 * m * (double)per_sample, as_stream(stream));
    hipLaunchKernelGGL(u8_normalize_kernel, dim3((unsigned)((per4 + 255) / 256), m), dim3(256), 0, as_stream(stream), src,
                       src_nstride, dst, dst_nstride, per4);
    return check_launch("u8_normalize_kernel");
}

extern "C" int pws_warp_norm_fwd(const float *src, size_t src_nstride, const float *grid, float *fake, const float *target,
                                 size_t tgt_nstride, double *l1_slots, int m, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && h > 0 && w >= 2, "pws_warp_norm_fwd: bad shape");
    if (m == 0) return PWS_OK;
    PWS_REQUIRE(src && grid && fake, "pws_warp_norm_fwd: NULL pointer");
    PWS_REQUIRE((target != nullptr) == (l1_slots != nullptr), "pws_warp_norm_fwd: target and l1_slots go together");
    PWS_REQUIRE(((size_t)h * w) % 4 == 0 && (size_t)h * w < (1u << 30) && al16(grid) && al16(fake) &&
                    (!target || (al16(target) && tgt_nstride % 4 == 0)),
                "pws_warp_norm_fwd: h*w must be a multiple of 4 and grid / fake / target 16-byte aligned");
    const size_t groups = (size_t)m * h * w / 4;
    const unsigned nb = (unsigned)((groups + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 120.0 * m * h * w, (double)m * h * w * (8.0 + 12.0 + 12.0 + (target ? 12.0 : 0.0)), as_stream(stream));
    hipLaunchKernelGGL(warp_norm_fwd_kernel, dim3(target && l1_slots && nb > kLossGrid ? kLossGrid : nb), dim3(256), 0, as_stream(stream), src, src_nstride, grid, fake, target,
                       tgt_nstride, l1_slots, h, w, groups, nb);
    return check_launch("warp_norm_fwd_kernel");
}

extern "C" int pws_warp_norm_bwd(const float *src, size_t src_nstride, const float *grid, const float *target,
                                 size_t tgt_nstride, float c_l1, const float *scale, const float *gextra, float *ggrid,
                                 int accumulate, int m, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && h > 0 && w > 0, "pws_warp_norm_bwd: bad shape");
    if (m == 0) return PWS_OK;
    PWS_REQUIRE(src && grid && ggrid, "pws_warp_norm_bwd: NULL pointer");
    PWS_REQUIRE((size_t)h * w < (1u << 30), "pws_warp_norm_bwd: plane too large");
    const size_t total = (size_t)m * h * w;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 150.0 * total, (double)total * (8.0 + 12.0 + 8.0 + (target ? 12.0 : 0.0) + (gextra ? 12.0 : 0.0)),
                   as_stream(stream));
    if (((size_t)h * w) % 4 == 0 && w >= 2 && al16(grid) && al16(ggrid) && (!gextra || al16(gextra)) && (!target || (al16(target) && tgt_nstride % 4 == 0)) &&
        g_experiment != 94) {   // 4 pixels per lane (94: one pixel per lane, A/B and tests)
        const size_t groups = total / 4;
        const unsigned nb4 = (unsigned)((groups + 255) / 256);
        hipLaunchKernelGGL(warp_norm_bwd4_kernel, dim3(nb4), dim3(256), 0, as_stream(stream), src, src_nstride, grid, target, tgt_nstride, c_l1, scale,
                           gextra, ggrid, accumulate, h, w, groups, nb4);
        return check_launch("warp_norm_bwd4_kernel");
    }
    hipLaunchKernelGGL(warp_norm_bwd_kernel, dim3(nb), dim3(256), 0, as_stream(stream), src, src_nstride, grid, target,
                       tgt_nstride, c_l1, scale, gextra, ggrid, accumulate, h, w, total, nb);
    return check_launch("warp_norm_bwd_kernel");
}

extern "C" int pws_temporal_l1_fwd(const float *fake1, const float *fake2, const float *theta, double *slots, int n, int h,
                                   int w, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0, "pws_temporal_l1_fwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(fake1 && fake2 && theta && slots, "pws_temporal_l1_fwd: NULL pointer");
    const size_t total = (size_t)n * h * w;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 90.0 * total, 24.0 * total, as_stream(stream));
    hipLaunchKernelGGL(temporal_l1_kernel<false>, dim3(nb > kLossGrid ? kLossGrid : nb), dim3(256), 0, as_stream(stream), fake1, fake2, theta, slots, 0.f,
                       (const float *)nullptr, (float *)nullptr, (float *)nullptr, h, w, total, nb);
    return check_launch("temporal_l1_kernel<fwd>");
}

extern "C" int pws_temporal_l1_bwd(const float *fake1, const float *fake2, const float *theta, float c, const float *scale,
                                   float *gfake1, float *gfake2, int n, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0, "pws_temporal_l1_bwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(fake1 && fake2 && theta && gfake1 && gfake2, "pws_temporal_l1_bwd: NULL pointer");
    const size_t total = (size_t)n * h * w;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 110.0 * total, 72.0 * total, as_stream(stream));
    // Measured (tools/temporal_ab.py, random frames): 108 / 416 us per launch at 32 / 128 samples for one lane per pixel with 12 memory
    // atomics, 156 / 598 us for the tiles with the scatter in LDS -- the L2 retires fp32 atomics at ~230 G/s, the tiled kernel's two
    // barriers, box zeroing / flush and 186 registers cost more than the atomics it saves.  NOT taken (PWS_OPT_EXPERIMENT 98 takes it).
    if (g_experiment == 99) {
        const int tx = (w + 15) / 16, ty = (h + 15) / 16;
        hipLaunchKernelGGL(temporal_l1_bwd_tile16_kernel, dim3((unsigned)(n * tx * ty)), dim3(256), 0, as_stream(stream), fake1, fake2, theta, c, scale,
                           gfake1, gfake2, h, w, tx, ty);
        return check_launch("temporal_l1_bwd_tile16_kernel");
    }
    if (g_experiment == 98) {
        const int tx = (w + kTT - 1) / kTT, ty = (h + kTT - 1) / kTT;
        hipLaunchKernelGGL(temporal_l1_bwd_tiled_kernel, dim3((unsigned)(n * tx * ty)), dim3(256), 0, as_stream(stream), fake1, fake2, theta, c, scale,
                           gfake1, gfake2, h, w, tx, ty);
        return check_launch("temporal_l1_bwd_tiled_kernel");
    }
    // neighbouring lanes merge the atomics of the taps they share -- only when a wave never straddles two samples (h * w % 64 == 0: the
    // hand-over compares plane-relative offsets, and two samples have different theta) (PWS_OPT_EXPERIMENT 93: never; 101: vertical neighbours as well -- a wave as
    // 32 columns x 2 rows: 1.5 instead of 2 atomics per pixel and plane, measured SLOWER, 91 vs 72 us: the narrower rows and the extra shuffles cost more)
    hipLaunchKernelGGL(temporal_l1_kernel<true>, dim3(nb), dim3(256), 0, as_stream(stream), fake1, fake2, theta,
                       (double *)nullptr, c, scale, gfake1, gfake2, h, w, total, nb, (float *)nullptr,
                       (total % 256 != 0 || ((size_t)h * w) % 64 != 0 || g_experiment == 93) ? 0 : ((w % 128 == 0 && h % 2 == 0 && g_experiment == 101) ? 2 : 1));
    return check_launch("temporal_l1_kernel<bwd>");
}

extern "C" int pws_temporal_l1_bwd_det(const float *fake1, const float *fake2, const float *theta, float c, const float *scale,
                                       float *gfake1, float *gfake2, float *scratch, int n, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0, "pws_temporal_l1_bwd_det: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(fake1 && fake2 && theta && gfake1 && gfake2 && scratch, "pws_temporal_l1_bwd_det: NULL pointer (scratch: n * 3 * h * w floats)");
    const size_t total = (size_t)n * h * w;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 400.0 * total, 120.0 * total, as_stream(stream));
    hipLaunchKernelGGL(temporal_l1_kernel<true>, dim3(nb), dim3(256), 0, as_stream(stream), fake1, fake2, theta, (double *)nullptr, c, scale,
                       gfake1, gfake2, h, w, total, nb, scratch);
    hipLaunchKernelGGL(temporal_gather_kernel, dim3(nb), dim3(256), 0, as_stream(stream), scratch, theta, gfake2, h, w, total);
    return check_launch("temporal_gather_kernel");
}

extern "C" int pws_feature_loss_bwd_det(const float *grid, const float *features, float c, const float *scale, float *ggrid, int m,
                                        int nf, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && nf >= 0 && h > 0 && w > 0, "pws_feature_loss_bwd_det: bad shape");
    if (m == 0 || nf == 0) return PWS_OK;
    PWS_REQUIRE(grid && features && ggrid, "pws_feature_loss_bwd_det: NULL pointer");
    hipLaunchKernelGGL(feature_loss_bwd_serial_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, as_stream(stream), grid, features, c, scale,
                       ggrid, m, nf, h, w);
    return check_launch("feature_loss_bwd_serial_kernel");
}

extern "C" int pws_feature_loss_fwd(const float *grid, const float *features, double *slots, int m, int nf, int h, int w,
                                    pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && nf >= 0 && h > 0 && w > 0, "pws_feature_loss_fwd: bad shape");
    if (m == 0 || nf == 0) return PWS_OK;
    PWS_REQUIRE(grid && features && slots, "pws_feature_loss_fwd: NULL pointer");
    const size_t total = (size_t)m * nf;
    hipLaunchKernelGGL(feature_loss_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), grid,
                       features, slots, 0.f, (const float *)nullptr, (float *)nullptr, nf, h, w, total);
    return check_launch("feature_loss_kernel<fwd>");
}

extern "C" int pws_feature_loss_bwd(const float *grid, const float *features, float c, const float *scale, float *ggrid, int m,
                                    int nf, int h, int w, pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && nf >= 0 && h > 0 && w > 0, "pws_feature_loss_bwd: bad shape");
    if (m == 0 || nf == 0) return PWS_OK;
    PWS_REQUIRE(grid && features && ggrid, "pws_feature_loss_bwd: NULL pointer");
    const size_t total = (size_t)m * nf;
    hipLaunchKernelGGL(feature_loss_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), grid,
                       features, (double *)nullptr, c, scale, ggrid, nf, h, w, total);
    return check_launch("feature_loss_kernel<bwd>");
}

extern "C" int pws_field_smoothness(const float *grid, double *slots_dx, double *slots_dy, int m, int h, int w,
                                    pws_stream_t stream) {
    PWS_REQUIRE(m >= 0 && h > 0 && w > 0, "pws_field_smoothness: bad shape");
    if (m == 0) return PWS_OK;
    PWS_REQUIRE(grid && slots_dx && slots_dy, "pws_field_smoothness: NULL pointer");
    const size_t total = (size_t)m * h * w;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_OBJECTIVE, 10.0 * total, 8.0 * total, as_stream(stream));
    if (w % 4 == 0 && al16(grid) && g_experiment != 110) {   // (110: one pixel per lane, A/B and tests)
        const size_t groups = total / 4;
        const unsigned nb4 = (unsigned)((groups + 255) / 256);
        hipLaunchKernelGGL(field_smoothness4_kernel, dim3(nb4 > kLossGrid ? kLossGrid : nb4), dim3(256), 0, as_stream(stream), grid, slots_dx, slots_dy, h, w, groups, nb4);
        return check_launch("field_smoothness4_kernel");
    }
    hipLaunchKernelGGL(field_smoothness_kernel, dim3(nb > kLossGrid ? kLossGrid : nb), dim3(256), 0, as_stream(stream), grid, slots_dx, slots_dy, h, w, total, nb);
    return check_launch("field_smoothness_kernel");
}

static int shape_args(const char *who, int m, int size, int block, int *bs, double *gi) {
    PWS_REQUIRE(m >= 0 && size > 0 && block > 1 && size % block == 0, "%s: bad shape", who);
    *bs = size / block;
    // the reference builds the basis for a block x block patch and applies it to (size/block)^2-pixel patches
    // (lib/utils.py:413-414, main_new.py:78): the two only agree when size == block^2
    PWS_REQUIRE(*bs == block, "%s: the reference's basis requires size == block*block (got size %d, block %d)", who, size, block);
    PWS_REQUIRE(*bs * *bs <= 1024, "%s: blocks of more than 1024 pixels", who);
    const double L = *bs - 1;
    double g00 = 0, g01 = 0, g11 = 0;
    for (int t = 0; t < *bs; ++t) {
        const double a0 = (L - t) / L, a1 = t / L;
        g00 += a0 * a0, g01 += a0 * a1, g11 += a1 * a1;
    }
    const double det = g00 * g11 - g01 * g01;
    gi[0] = g11 / det, gi[1] = -g01 / det, gi[2] = g00 / det;
    return PWS_OK;
}

extern "C" int pws_shape_loss_fwd(const float *resid, double *slots, int m, int size, int block, pws_stream_t stream) {
    int bs = 0;
    double gi[3];
    const int rc = shape_args("pws_shape_loss_fwd", m, size, block, &bs, gi);
    if (rc != PWS_OK) return rc;
    if (m == 0) return PWS_OK;
    PWS_REQUIRE(resid && slots, "pws_shape_loss_fwd: NULL pointer");
    const int threads = ((bs * bs + 63) / 64) * 64;
    ProfScope prof(KID_OBJECTIVE, 120.0 * m * size * size, 8.0 * m * size * size, as_stream(stream));
    if (bs == 16 && al16(resid) && g_experiment != 97) {   // one wave per block (97: the general kernel, A/B and tests)
        const unsigned nb = (unsigned)(m * block * block);
        hipLaunchKernelGGL(shape_loss16_kernel<false>, dim3((nb + 3) / 4 > kLossGrid ? kLossGrid : (nb + 3) / 4), dim3(256), 0, as_stream(stream), resid, slots, 0.0, (const float *)nullptr,
                           (float *)nullptr, size, block, nb, gi[0], gi[1], gi[2]);
        return check_launch("shape_loss16_kernel<fwd>");
    }
    hipLaunchKernelGGL(shape_loss_kernel<false>, dim3((unsigned)(m * block * block)), dim3(threads), 0, as_stream(stream), resid,
                       slots, 0.0, (const float *)nullptr, (float *)nullptr, size, block, bs, gi[0], gi[1], gi[2]);
    return check_launch("shape_loss_kernel<fwd>");
}

extern "C" int pws_shape_loss_bwd(const float *resid, double c, const float *scale, float *gresid, int m, int size, int block,
                                  pws_stream_t stream) {
    int bs = 0;
    double gi[3];
    const int rc = shape_args("pws_shape_loss_bwd", m, size, block, &bs, gi);
    if (rc != PWS_OK) return rc;
    if (m == 0) return PWS_OK;
    PWS_REQUIRE(resid && gresid, "pws_shape_loss_bwd: NULL pointer");
    const int threads = ((bs * bs + 63) / 64) * 64;
    ProfScope prof(KID_OBJECTIVE, 240.0 * m * size * size, 16.0 * m * size * size, as_stream(stream));
    if (bs == 16 && al16(resid) && al16(gresid) && g_experiment != 97) {
        const unsigned nb = (unsigned)(m * block * block);
        hipLaunchKernelGGL(shape_loss16_kernel<true>, dim3((nb + 3) / 4), dim3(256), 0, as_stream(stream), resid, (double *)nullptr, c, scale, gresid, size,
                           block, nb, gi[0], gi[1], gi[2]);
        return check_launch("shape_loss16_kernel<bwd>");
    }
    hipLaunchKernelGGL(shape_loss_kernel<true>, dim3((unsigned)(m * block * block)), dim3(threads), 0, as_stream(stream), resid,
                       (double *)nullptr, c, scale, gresid, size, block, bs, gi[0], gi[1], gi[2]);
    return check_launch("shape_loss_kernel<bwd>");
}

extern "C" int pws_objective_finalize(const double *slots, int nq, const double *coef, int nout, float *out, pws_stream_t stream) {
    PWS_REQUIRE(nq > 0 && nout > 0 && nout <= 64, "pws_objective_finalize: bad shape");
    PWS_REQUIRE(slots && coef && out, "pws_objective_finalize: NULL pointer");
    hipLaunchKernelGGL(objective_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), slots, nq, coef, nout, out);
    return check_launch("objective_finalize_kernel");
}

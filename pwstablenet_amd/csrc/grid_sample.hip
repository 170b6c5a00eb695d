// Bilinear warp kernels (HBM-bound): F.grid_sample forward / backward (bilinear, zeros padding), F.affine_grid,
// UpsamplingBilinear2d (align_corners=True) and the fused "resize the 256x256 field to the frame size and
// apply it" path of the reference's video loop (reference main_new.py:106-118,197,706-716).
//
// Layout: frames NCHW fp32, fields N,H,W,2 fp32 -- both exactly as the reference hands them over.
// Mapping: one lane owns PPT consecutive output pixels of a row, so the field read (8*PPT B/lane) and each
// channel-plane store (4*PPT B/lane) are fully coalesced; the 4*C taps per pixel are gathers that hit L1/L2
// because neighbouring lanes sample neighbouring source pixels when the field is smooth.  Workgroup ids are
// remapped so that one XCD (one private L2) walks a contiguous range of rows of the same image.
// Algorithmic traffic (C=3): fwd 8 (field) + 12 (frame) + 12 (out) = 32 B/pixel.
#include "common.h"

namespace pws {

struct Taps {
    int o00, o01, o10, o11;  // clamped linear offsets inside one channel plane
    float w00, w01, w10, w11;  // bilinear weights, zeroed for out-of-range taps
    float wx0, wx1, wy0, wy1;
    bool v00, v01, v10, v11;
};

__device__ __forceinline__ float unnormalize(float g, int size, bool align_corners) {
    // align_corners=False: ((g+1)*size-1)/2 ; True: (g+1)/2*(size-1) -- evaluated with one rounding
    return align_corners ? fmaf(g, 0.5f * (float)(size - 1), 0.5f * (float)(size - 1))
                         : fmaf(g, 0.5f * (float)size, 0.5f * (float)(size - 1));
}

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W, bool ac) {
    Taps t;
    const float ix = unnormalize(gx, W, ac), iy = unnormalize(gy, H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    // NaN / huge coordinates: the float->int conversion saturates, every tap is then out of range -> 0
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    t.wx1 = ix - fx, t.wx0 = 1.f - t.wx1, t.wy1 = iy - fy, t.wy0 = 1.f - t.wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    t.v00 = vy0 && vx0, t.v01 = vy0 && vx1, t.v10 = vy1 && vx0, t.v11 = vy1 && vx1;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    t.o00 = cy0 * W + cx0, t.o01 = cy0 * W + cx1, t.o10 = cy1 * W + cx0, t.o11 = cy1 * W + cx1;
    t.w00 = t.v00 ? t.wx0 * t.wy0 : 0.f;
    t.w01 = t.v01 ? t.wx1 * t.wy0 : 0.f;
    t.w10 = t.v10 ? t.wx0 * t.wy1 : 0.f;
    t.w11 = t.v11 ? t.wx1 * t.wy1 : 0.f;
    return t;
}

// ---------------------------------------------------------------------------------------------- forward
template <int PPT>
__global__ void __launch_bounds__(256) grid_sample_fwd_kernel(const float *__restrict__ input,
                                                              const float *__restrict__ grid, float *__restrict__ out,
                                                              int C, int H, int W, int HoWo, size_t total_groups,
                                                              unsigned nblocks, int ac) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx = (size_t)blk * 256 + threadIdx.x;  // group of PPT consecutive output pixels
    if (gidx >= total_groups) return;
    const size_t p0 = gidx * PPT;  // HoWo % PPT == 0, so a group never straddles two images
    const int n = (int)(p0 / HoWo);
    const int hw = (int)(p0 % HoWo);
    float g[2 * PPT];
    if constexpr (PPT == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(grid + p0 * 2);
        const float4 b = *reinterpret_cast<const float4 *>(grid + p0 * 2 + 4);
        g[0] = a.x, g[1] = a.y, g[2] = a.z, g[3] = a.w, g[4] = b.x, g[5] = b.y, g[6] = b.z, g[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 2 * PPT; ++i) g[i] = grid[p0 * 2 + i];
    }
    Taps t[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) t[i] = make_taps(g[2 * i], g[2 * i + 1], H, W, ac != 0);
    const size_t plane = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float *ip = input + ((size_t)n * C + c) * plane;
        float r[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            r[i] = ip[t[i].o00] * t[i].w00 + ip[t[i].o01] * t[i].w01 + ip[t[i].o10] * t[i].w10 + ip[t[i].o11] * t[i].w11;
        float *op = out + ((size_t)n * C + c) * HoWo + hw;
        if constexpr (PPT == 4) {
            *reinterpret_cast<float4 *>(op) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) op[i] = r[i];
        }
    }
}

// Paired taps: the two taps of one source row are adjacent in memory, so they are fetched with ONE 8-byte load
// (global_load_dwordx2 needs only 4-byte alignment) -- half the gather instructions of the 4-scalar-tap form.
// xs = clamp(x0, 0, W-2) is the pair start; at the left/right image border the valid tap moves to the other half.
struct __attribute__((packed, aligned(4))) F2U {
    float x, y;
};
struct __attribute__((packed, aligned(4))) F4U {
    float x, y, z, w;
};
struct Taps2 {
    int o0, o1;          // offsets of the two row pairs inside a channel plane
    float a0, b0, a1, b1;  // weights of (pair.x, pair.y) for row 0 and row 1
    int xs, r0, r1;      // pair start column and the two (clamped) rows: o0 = r0 * W + xs (row-window variant of the u8 kernel)
};

__device__ __forceinline__ Taps2 make_taps2(float gx, float gy, int H, int W, bool ac) {
    const float ix = unnormalize(gx, W, ac), iy = unnormalize(gy, H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int xs = min(max(x0, 0), W - 2);
    const int sel = x0 - xs;
    const float wl = sel == 0 ? (vx0 ? wx0 : 0.f) : (sel == -1 ? (vx1 ? wx1 : 0.f) : 0.f);
    const float wr = sel == 0 ? (vx1 ? wx1 : 0.f) : (sel == 1 ? (vx0 ? wx0 : 0.f) : 0.f);
    const float r0 = vy0 ? wy0 : 0.f, r1 = vy1 ? wy1 : 0.f;
    Taps2 t;
    t.xs = xs, t.r0 = min(max(y0, 0), H - 1), t.r1 = min(max(y1, 0), H - 1);
    t.o0 = t.r0 * W + xs, t.o1 = t.r1 * W + xs;
    t.a0 = wl * r0, t.b0 = wr * r0, t.a1 = wl * r1, t.b1 = wr * r1;
    return t;
}

// NT: streaming (non-temporal) field loads and output stores -- data touched exactly once -- so that L2 / Infinity Cache keep
// the frame lines that neighbouring rows re-read.  Pays when the launch's traffic exceeds the 256 MB Infinity Cache
// (N=256: 140 -> 127 us); below that it costs (N=64: 27 -> 34 us), so the entry point picks by size.
// ROWWIN (PWS_OPT_EXPERIMENT 5, measured and NOT taken -- DESIGN.md "wavefront shuffles"): a lane whose 4 pixels read at most 5
// consecutive columns of ONE source row pair fetches each row of a plane as one 16-byte load and takes the fifth column from the
// next lane's window through a wave shuffle (when that window starts exactly 4 columns further), instead of 4 paired 8-byte gathers
// per row: 2 (+ fallback) instead of 8 vector memory instructions per lane and plane.
template <int PPT, bool NT, bool ROWWIN = false>
__global__ void __launch_bounds__(256) grid_sample_fwd2_kernel(const float *__restrict__ input,
                                                               const float *__restrict__ grid, float *__restrict__ out,
                                                               int C, int H, int W, int HoWo, size_t total_groups,
                                                               unsigned nblocks, int ac) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx_raw = (size_t)blk * 256 + threadIdx.x;
    if (!ROWWIN && gidx_raw >= total_groups) return;
    const bool live = gidx_raw < total_groups;   // ROWWIN: lanes past the end run on the last group (the shuffles want every lane)
    const size_t gidx = live ? gidx_raw : total_groups - 1;
    const size_t p0 = gidx * PPT;
    const int n = (int)(p0 / HoWo);
    const int hw = (int)(p0 % HoWo);
    float g[2 * PPT];
    if constexpr (PPT == 4) {
        f32x4 a, b;
        if constexpr (NT) {
            a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(grid + p0 * 2));
            b = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(grid + p0 * 2 + 4));
        } else {
            a = *reinterpret_cast<const f32x4 *>(grid + p0 * 2);
            b = *reinterpret_cast<const f32x4 *>(grid + p0 * 2 + 4);
        }
        g[0] = a.x, g[1] = a.y, g[2] = a.z, g[3] = a.w, g[4] = b.x, g[5] = b.y, g[6] = b.z, g[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 2 * PPT; ++i) g[i] = grid[p0 * 2 + i];
    }
    Taps2 t[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) t[i] = make_taps2(g[2 * i], g[2 * i + 1], H, W, ac != 0);
    const size_t plane = (size_t)H * W;
    bool fast = false;
    int lo = 0;
    bool loads = false;
    if constexpr (ROWWIN && PPT == 4) {
        lo = t[0].xs;
        int hi_ = t[0].xs;
        bool same = true;
#pragma unroll
        for (int i = 1; i < PPT; ++i) {
            lo = min(lo, t[i].xs), hi_ = max(hi_, t[i].xs);
            same = same && t[i].r0 == t[0].r0 && t[i].r1 == t[0].r1;
        }
        loads = same && (size_t)t[0].r0 * W + lo + 4 <= plane && (size_t)t[0].r1 * W + lo + 4 <= plane;   // the window stays inside the plane
        const int span = hi_ - lo;   // the last pixel's pair ends at column lo + span + 1
        const int nb_lo = __shfl_down(lo, 1, 64), nb_r0 = __shfl_down(t[0].r0, 1, 64), nb_r1 = __shfl_down(t[0].r1, 1, 64);
        const bool nb_loads = __shfl_down((int)loads, 1, 64) != 0;
        const bool nb_ok = (threadIdx.x & 63) != 63 && nb_loads && nb_lo == lo + 4 && nb_r0 == t[0].r0 && nb_r1 == t[0].r1;
        fast = loads && (span <= 2 || (span == 3 && nb_ok));
    }
    for (int c = 0; c < C; ++c) {
        const float *ip = input + ((size_t)n * C + c) * plane;
        float r[PPT];
        if constexpr (ROWWIN && PPT == 4) {
            float w0[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, w1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            if (loads) {
                const F4U a = *reinterpret_cast<const F4U *>(ip + (size_t)t[0].r0 * W + lo);
                const F4U b = *reinterpret_cast<const F4U *>(ip + (size_t)t[0].r1 * W + lo);
                w0[0] = a.x, w0[1] = a.y, w0[2] = a.z, w0[3] = a.w, w1[0] = b.x, w1[1] = b.y, w1[2] = b.z, w1[3] = b.w;
            }
            w0[4] = __shfl_down(w0[0], 1, 64), w1[4] = __shfl_down(w1[0], 1, 64);   // the next lane's first column = this lane's fifth
            if (fast) {
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    const int d = t[i].xs - lo;   // 0 .. 3
                    const float ux = d == 0 ? w0[0] : (d == 1 ? w0[1] : (d == 2 ? w0[2] : w0[3]));
                    const float uy = d == 0 ? w0[1] : (d == 1 ? w0[2] : (d == 2 ? w0[3] : w0[4]));
                    const float vx = d == 0 ? w1[0] : (d == 1 ? w1[1] : (d == 2 ? w1[2] : w1[3]));
                    const float vy = d == 0 ? w1[1] : (d == 1 ? w1[2] : (d == 2 ? w1[3] : w1[4]));
                    r[i] = ux * t[i].a0 + uy * t[i].b0 + vx * t[i].a1 + vy * t[i].b1;
                }
            }
        }
        if (!fast) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const F2U u = *reinterpret_cast<const F2U *>(ip + t[i].o0);
                const F2U v = *reinterpret_cast<const F2U *>(ip + t[i].o1);
                r[i] = u.x * t[i].a0 + u.y * t[i].b0 + v.x * t[i].a1 + v.y * t[i].b1;
            }
        }
        if (ROWWIN && !live) continue;
        float *op = out + ((size_t)n * C + c) * HoWo + hw;
        if constexpr (PPT == 4) {
            const f32x4 o = {r[0], r[1], r[2], r[3]};
            if constexpr (NT)
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(op));
            else
                *reinterpret_cast<f32x4 *>(op) = o;
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) op[i] = r[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
// One lane per output pixel.  grad wrt the field is a per-pixel reduction over channels (no atomics);
// grad wrt the frame is a 4*C-tap scatter-add (fp32 atomics in L2; order-dependent in the last ulps).
__global__ void __launch_bounds__(256) grid_sample_bwd_kernel(const float *__restrict__ gout,
                                                              const float *__restrict__ input,
                                                              const float *__restrict__ grid, float *__restrict__ ginput,
                                                              float *__restrict__ ggrid, int C, int H, int W, int HoWo,
                                                              size_t total, unsigned nblocks, int ac) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t p = (size_t)blk * 256 + threadIdx.x;
    if (p >= total) return;
    const int n = (int)(p / HoWo);
    const int hw = (int)(p % HoWo);
    const float2 g = *reinterpret_cast<const float2 *>(grid + p * 2);
    const Taps t = make_taps(g.x, g.y, H, W, ac != 0);
    const size_t plane = (size_t)H * W;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float go = gout[((size_t)n * C + c) * HoWo + hw];
        const float *ip = input + ((size_t)n * C + c) * plane;
        const float v00 = t.v00 ? ip[t.o00] : 0.f, v01 = t.v01 ? ip[t.o01] : 0.f;
        const float v10 = t.v10 ? ip[t.o10] : 0.f, v11 = t.v11 ? ip[t.o11] : 0.f;
        gix += go * ((v01 - v00) * t.wy0 + (v11 - v10) * t.wy1);
        giy += go * ((v10 - v00) * t.wx0 + (v11 - v01) * t.wx1);
        if (ginput) {
            float *gp = ginput + ((size_t)n * C + c) * plane;
            if (t.v00) atomicAdd(gp + t.o00, go * t.w00);
            if (t.v01) atomicAdd(gp + t.o01, go * t.w01);
            if (t.v10) atomicAdd(gp + t.o10, go * t.w10);
            if (t.v11) atomicAdd(gp + t.o11, go * t.w11);
        }
    }
    if (ggrid) {
        const float sx = ac ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
        const float sy = ac ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
        *reinterpret_cast<float2 *>(ggrid + p * 2) = make_float2(gix * sx, giy * sy);
    }
}

// Gradient wrt the FIELD only (the frame is data: reference main_new.py:106-118, what loss_g.backward() needs of every warp):
// no scatter, so the forward's access pattern applies -- 4 consecutive pixels per lane, one 32-byte field read, one 16-byte read of
// every upstream channel plane, the two taps of a source row as ONE unaligned 8-byte gather, one 32-byte store of the result.
//   d out / d ix = (v01 - v00) wy0 + (v11 - v10) wy1 ,  d out / d iy = (v10 - v00) wx0 + (v11 - v01) wx1   (out-of-range taps = 0)
// are folded into 8 coefficients per pixel of the four fetched values (left / right of row pair 0 and 1); a channel then costs
// 2 gathers + 10 FMAs per pixel.  Algorithmic traffic (C = 3): 8 (field) + 12 (upstream) + 12 (frame) + 8 (result) = 40 B/pixel.
struct TapsG {
    int o0, o1;              // offsets of the two row pairs inside a channel plane
    float xl0, xr0, xl1, xr1;  // d/d ix coefficients of (pair0.x, pair0.y, pair1.x, pair1.y)
    float yl0, yr0, yl1, yr1;  // d/d iy
};
__device__ __forceinline__ TapsG make_taps_g(float gx, float gy, int H, int W, bool ac) {
    const float ix = unnormalize(gx, W, ac), iy = unnormalize(gy, H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int xs = min(max(x0, 0), W - 2);
    const int sel = x0 - xs;   // 0: pair = (x0, x1); -1: pair.x = x1 (x0 left of the image); +1: pair.y = x0 (x1 right of it)
    // coefficient of tap (x0, row) / (x1, row), zero when the tap is out of range
    const float cx0_0 = vx0 && vy0 ? -wy0 : 0.f, cx1_0 = vx1 && vy0 ? wy0 : 0.f, cx0_1 = vx0 && vy1 ? -wy1 : 0.f, cx1_1 = vx1 && vy1 ? wy1 : 0.f;
    const float cy0_0 = vx0 && vy0 ? -wx0 : 0.f, cy1_0 = vx1 && vy0 ? -wx1 : 0.f, cy0_1 = vx0 && vy1 ? wx0 : 0.f, cy1_1 = vx1 && vy1 ? wx1 : 0.f;
    TapsG t;
    t.o0 = min(max(y0, 0), H - 1) * W + xs, t.o1 = min(max(y1, 0), H - 1) * W + xs;
    t.xl0 = sel == 0 ? cx0_0 : (sel == -1 ? cx1_0 : 0.f), t.xr0 = sel == 0 ? cx1_0 : (sel == 1 ? cx0_0 : 0.f);
    t.xl1 = sel == 0 ? cx0_1 : (sel == -1 ? cx1_1 : 0.f), t.xr1 = sel == 0 ? cx1_1 : (sel == 1 ? cx0_1 : 0.f);
    t.yl0 = sel == 0 ? cy0_0 : (sel == -1 ? cy1_0 : 0.f), t.yr0 = sel == 0 ? cy1_0 : (sel == 1 ? cy0_0 : 0.f);
    t.yl1 = sel == 0 ? cy0_1 : (sel == -1 ? cy1_1 : 0.f), t.yr1 = sel == 0 ? cy1_1 : (sel == 1 ? cy0_1 : 0.f);
    return t;
}

template <bool NT>
__global__ void __launch_bounds__(256) grid_sample_bwd_field_kernel(const float *__restrict__ gout, const float *__restrict__ input,
                                                                    const float *__restrict__ grid, float *__restrict__ ggrid, int C,
                                                                    int H, int W, int HoWo, size_t total_groups, unsigned nblocks,
                                                                    int ac) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx = (size_t)blk * 256 + threadIdx.x;
    if (gidx >= total_groups) return;
    const size_t p0 = gidx * 4;   // HoWo % 4 == 0: the 4 pixels stay inside one image
    const int n = (int)(p0 / HoWo);
    const int hw = (int)(p0 % HoWo);
    f32x4 a, b;
    if constexpr (NT) {
        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(grid + p0 * 2));
        b = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(grid + p0 * 2 + 4));
    } else {
        a = *reinterpret_cast<const f32x4 *>(grid + p0 * 2);
        b = *reinterpret_cast<const f32x4 *>(grid + p0 * 2 + 4);
    }
    const float g[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    TapsG t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = make_taps_g(g[2 * i], g[2 * i + 1], H, W, ac != 0);
    const size_t plane = (size_t)H * W;
    float gix[4] = {0.f, 0.f, 0.f, 0.f}, giy[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float *ip = input + ((size_t)n * C + c) * plane;
        const float *gp = gout + ((size_t)n * C + c) * HoWo + hw;
        f32x4 go;
        if constexpr (NT) go = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(gp));
        else go = *reinterpret_cast<const f32x4 *>(gp);
        const float gov[4] = {go.x, go.y, go.z, go.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const F2U u = *reinterpret_cast<const F2U *>(ip + t[i].o0);
            const F2U v = *reinterpret_cast<const F2U *>(ip + t[i].o1);
            gix[i] += gov[i] * (u.x * t[i].xl0 + u.y * t[i].xr0 + v.x * t[i].xl1 + v.y * t[i].xr1);
            giy[i] += gov[i] * (u.x * t[i].yl0 + u.y * t[i].yr0 + v.x * t[i].yl1 + v.y * t[i].yr1);
        }
    }
    const float sx = ac ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
    const float sy = ac ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
    const f32x4 o0 = {gix[0] * sx, giy[0] * sy, gix[1] * sx, giy[1] * sy}, o1 = {gix[2] * sx, giy[2] * sy, gix[3] * sx, giy[3] * sy};
    if constexpr (NT) {
        __builtin_nontemporal_store(o0, reinterpret_cast<f32x4 *>(ggrid + p0 * 2));
        __builtin_nontemporal_store(o1, reinterpret_cast<f32x4 *>(ggrid + p0 * 2 + 4));
    } else {
        *reinterpret_cast<f32x4 *>(ggrid + p0 * 2) = o0;
        *reinterpret_cast<f32x4 *>(ggrid + p0 * 2 + 4) = o1;
    }
}

// ---------------------------------------------------------------------------------------------- affine_grid
__device__ __forceinline__ float base_coord(int j, int size, bool ac) {
    if (ac) return size > 1 ? (2.f * j) / (float)(size - 1) - 1.f : 0.f;
    return (2.f * j + 1.f) / (float)size - 1.f;
}

__global__ void affine_grid_kernel(const float *__restrict__ theta, float *__restrict__ grid, int H, int W, size_t total,
                                   int ac) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int x_ = (int)(p % W), y_ = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
    const float *t = theta + (size_t)n * 6;
    const float x = base_coord(x_, W, ac != 0), y = base_coord(y_, H, ac != 0);
    *reinterpret_cast<float2 *>(grid + p * 2) = make_float2(t[0] * x + t[1] * y + t[2], t[3] * x + t[4] * y + t[5]);
}

// ---------------------------------------------------------------------------------------------- upsample (ac=True)
// Source index of an output pixel as torch computes it: scale * index ROUNDED to fp32 (mul_rounded, common.h), then split into
// floor and fraction.  A contracted `scale * index - floor` (one fma on the exact product) moves the fraction by up to an ulp of the
// index, 1.5e-5 at 255, i.e. 5e-5 on the interpolated value.
__global__ void upsample_bilinear_ac_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W, int Ho,
                                            int Wo, float ry, float rx, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const size_t nc = i / ((size_t)Wo * Ho);
    const float sy = mul_rounded(ry, (float)oy), sx = mul_rounded(rx, (float)ox);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly, lx = sx - x0, hx = 1.f - lx;
    const float *ip = in + nc * (size_t)H * W;
    out[i] = hy * (hx * ip[y0 * W + x0] + lx * ip[y0 * W + x1]) + ly * (hx * ip[y1 * W + x0] + lx * ip[y1 * W + x1]);
}

// Adjoint of the resize above (autograd of UpsamplingBilinear2d when the driver's video loop runs with gradients enabled,
// reference main_new.py:697-710 has no no_grad): one lane per INPUT element gathers the output elements whose taps touch it,
// re-deriving every output's (y0, y1, ly) with the forward's own arithmetic, so the two are exact adjoints and the sum is
// deterministic (no atomics).
__global__ void upsample_bilinear_ac_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin, int H, int W, int Ho,
                                                int Wo, float ry, float rx, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t nc = i / ((size_t)W * H);
    // outputs with ry*oy in (y-1, y+1), widened by one on both sides against rounding; membership is re-checked below
    int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
    if (ry > 0.f) oy_lo = max(0, (int)floorf((float)(y - 1) / ry) - 1), oy_hi = min(Ho - 1, (int)ceilf((float)(y + 1) / ry) + 1);
    if (rx > 0.f) ox_lo = max(0, (int)floorf((float)(x - 1) / rx) - 1), ox_hi = min(Wo - 1, (int)ceilf((float)(x + 1) / rx) + 1);
    const float *gp = gout + nc * (size_t)Ho * Wo;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float sy = mul_rounded(ry, (float)oy);
        const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float ly = sy - y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const float sx = mul_rounded(rx, (float)ox);
            const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx = sx - x0;
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            row = fmaf(wx, gp[(size_t)oy * Wo + ox], row);
        }
        acc = fmaf(wy, row, acc);
    }
    gin[i] = acc;
}

// Adjoint of affine_grid: gtheta[n] = sum_{h,w} ggrid[n,h,w,:] (x) [x_w, y_h, 1]  (autograd of F.affine_grid when theta carries a
// gradient).  One workgroup per sample, fp64 partial sums per lane, LDS tree: deterministic.
__global__ void __launch_bounds__(256) affine_grid_bwd_kernel(const float *__restrict__ ggrid, float *__restrict__ gtheta, int H,
                                                              int W, int ac) {
    const int n = blockIdx.x;
    const float2 *g = reinterpret_cast<const float2 *>(ggrid) + (size_t)n * H * W;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int p = threadIdx.x; p < H * W; p += 256) {
        const float x = base_coord(p % W, W, ac != 0), y = base_coord(p / W, H, ac != 0);
        const float2 v = g[p];
        s[0] += (double)v.x * x, s[1] += (double)v.x * y, s[2] += v.x;
        s[3] += (double)v.y * x, s[4] += (double)v.y * y, s[5] += v.y;
    }
    __shared__ double red[6][256];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x < 6) gtheta[(size_t)n * 6 + threadIdx.x] = (float)red[threadIdx.x][0];
}

// The field's bilinear interpolation with its roundings spelled out (one fma chain): left as a plain expression, hipcc contracted
// it differently in different instantiations of the kernels below, and a last-ulp difference of a coordinate flips a byte of the
// uint8 output where the blend lands on an integer.
__device__ __forceinline__ float field_lerp(float hy, float ly, float hx, float lx, float a, float b, float c, float d) {
    const float top = fmaf(lx, b, hx * a), bot = fmaf(lx, d, hx * c);
    return fmaf(ly, bot, hy * top);
}

// ---------------------------------------------------------------------------------------------- fused 720p path
// field [n,fh,fw,2] --(bilinear, align_corners=True, never materialised)--> per-pixel (gx,gy) --> 4-tap gather.
// NARROW: the PPT pixels of a lane span less than one field cell ((PPT-1)*rx < 1, e.g. 256 -> 1280 columns), so the lane
// fetches the 3 field columns (x 2 rows) it can touch once instead of 4 loads per pixel.
template <int PPT, bool NARROW, bool NT>
__global__ void __launch_bounds__(256) upsample_grid_sample_fwd_kernel(const float *__restrict__ input,
                                                                       const float *__restrict__ field,
                                                                       float *__restrict__ out, int C, int H, int W, int fh,
                                                                       int fw, float ry, float rx, size_t total_groups,
                                                                       unsigned nblocks, int ac) {
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx = (size_t)blk * 256 + threadIdx.x;
    if (gidx >= total_groups) return;
    const int HW = H * W;
    const size_t p0 = gidx * PPT;
    const int n = (int)(p0 / HW);
    const int hw = (int)(p0 % HW);
    const int oy = hw / W, ox0 = hw % W;  // W % PPT == 0: the group stays inside one row
    const float sy = mul_rounded(ry, (float)oy);
    const int y0 = (int)sy, y1 = y0 + (y0 < fh - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly;
    const float2 *f0 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y0) * fw;
    const float2 *f1 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y1) * fw;
    Taps2 t[PPT];
    if constexpr (NARROW) {
        const int cb = (int)mul_rounded(rx, (float)ox0);  // first column this lane can touch; it needs cb .. cb+2 at most
        float2 r0[3], r1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int col = min(cb + k, fw - 1);
            r0[k] = f0[col], r1[k] = f1[col];
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float sx = mul_rounded(rx, (float)(ox0 + i));
            const int x0 = (int)sx;
            const float lx = sx - x0, hx = 1.f - lx;
            const bool second = x0 > cb;  // x0 - cb is 0 or 1; x1 = min(x0 + 1, fw - 1) is column x0 - cb + 1 (clamped loads)
            const float2 a = second ? r0[1] : r0[0], b = second ? r0[2] : r0[1];
            const float2 c = second ? r1[1] : r1[0], d = second ? r1[2] : r1[1];
            const float gx = field_lerp(hy, ly, hx, lx, a.x, b.x, c.x, d.x);
            const float gy = field_lerp(hy, ly, hx, lx, a.y, b.y, c.y, d.y);
            t[i] = make_taps2(gx, gy, H, W, ac != 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float sx = mul_rounded(rx, (float)(ox0 + i));
            const int x0 = (int)sx, x1 = x0 + (x0 < fw - 1 ? 1 : 0);
            const float lx = sx - x0, hx = 1.f - lx;
            const float2 a = f0[x0], b = f0[x1], c = f1[x0], d = f1[x1];
            const float gx = field_lerp(hy, ly, hx, lx, a.x, b.x, c.x, d.x);
            const float gy = field_lerp(hy, ly, hx, lx, a.y, b.y, c.y, d.y);
            t[i] = make_taps2(gx, gy, H, W, ac != 0);
        }
    }
    for (int c = 0; c < C; ++c) {
        const float *ip = input + ((size_t)n * C + c) * HW;
        float r[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const F2U u = *reinterpret_cast<const F2U *>(ip + t[i].o0);
            const F2U v = *reinterpret_cast<const F2U *>(ip + t[i].o1);
            r[i] = u.x * t[i].a0 + u.y * t[i].b0 + v.x * t[i].a1 + v.y * t[i].b1;
        }
        float *op = out + ((size_t)n * C + c) * HW + hw;
        if constexpr (PPT == 4) {
            const f32x4 o = {r[0], r[1], r[2], r[3]};
            if constexpr (NT)
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(op));
            else
                *reinterpret_cast<f32x4 *>(op) = o;
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) op[i] = r[i];
        }
    }
}

// uint8 HWC variant of the fused 720p path: the frame is what cv2 hands over (main_new.py:679-684: BGR uint8 [h][w][3], converted
// to RGB float CHW there) and the result is what the reference writes (main_new.py:717-721: float -> astype(uint8) HWC), so a
// frame moves 3 + 3 bytes per pixel instead of 12 + 12.  Field interpolation and tap weights are the float kernel's, the blend
// is evaluated in the same order on the same fp32 values, then truncated like numpy's astype(uint8) (values are in [0, 255]).
// swap_rb: output channel c = input channel 2 - c (the reference's COLOR_BGR2RGB before the warp).
//
// What bounds this kernel (round 6, 8 frames of 1280 x 720, profiles/r06_warp_u8_rewrite.txt): NOT its 6 bytes per pixel, and not the count of its
// vector-ALU instructions either (the round-5 cut ran 35 us with 613 of them per lane and 35 us with 309) but, in this order,
//   1. the NUMBER of vector memory instructions (a CU's address unit spends ~16-30 cycles per wave instruction whatever its width) and their
//      ALIGNMENT (a byte-aligned dword / 8-byte load costs about twice an aligned one): 23 per lane (per tap row a 4- and a 2-byte
//      unaligned load, 6 field loads, a store) -> 11 (per tap row ONE aligned 12-byte load + two v_alignbyte; the field's row pair fetched
//      once per wave, one column per lane, and handed round with ds_bpermute; a 12-byte store): 35 -> 22 us;
//   2. then the vector ALU (tools/probes/valu_rate_probe.hip: conversions, VOP3 forms and packed fp32 issue in ~4.5 cycles, plain
//      fp32 / moves / logic in ~2.5): a WAVE owns 256 consecutive pixels of ONE output row, so sample, row, the field's row pair, its vertical
//      weights and every base address are wave-uniform (scalar registers, no per-lane division, loads and stores are base + 32-bit offset);
//      a lane's 4 pixels run as two PAIRS through packed fp32 (the same IEEE roundings as the scalar forms: field interpolation,
//      un-normalisation, tap weights and the 4-term blend are the float kernel's operations in the float kernel's order); the border logic
//      runs only in waves that have a tap outside the frame (wave-uniform branch); float -> byte is one SDWA convert per byte.
struct __attribute__((packed, aligned(1))) U8x4 {
    unsigned v;
};
struct __attribute__((packed, aligned(1))) U8x2 {
    unsigned short v;
};
struct __attribute__((aligned(4))) U32x3 {
    unsigned x, y, z;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 pk_floor(f32x2 v) { return f32x2{floorf(v.x), floorf(v.y)}; }
__device__ __forceinline__ f32x2 pk_mul_rounded(f32x2 a, f32x2 b) {   // (mul_rounded of common.h, two at a time)
    f32x2 p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ f32x2 pk_field_lerp(f32x2 hy, f32x2 ly, f32x2 hx, f32x2 lx, f32x2 a, f32x2 b, f32x2 c, f32x2 d) {
    const f32x2 top = pk_fma(lx, b, hx * a), bot = pk_fma(lx, d, hx * c);   // field_lerp above, per component
    return pk_fma(ly, bot, hy * top);
}
// float -> byte B of `word`, truncating like astype(uint8) (the blend is in [0, 255.0001]; v_cvt_u32_f32 truncates and saturates at 0): one
// SDWA instruction converts and writes the byte in place.  (v_cvt_pk_u8_f32 rounds to NEAREST -- measured -- and would need a floor in front.)
__device__ __forceinline__ void cvt_u8_into(int B, unsigned &word, float v) {   // (B: a constant once the caller's loops are unrolled)
    if (B == 0) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(word) : "v"(v));
    if (B == 1) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(word) : "v"(v));
    if (B == 2) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(word) : "v"(v));
    if (B == 3) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(word) : "v"(v));
}
struct TapsU8 {   // one pixel's source: byte offsets of its two row pairs inside the sample, weights of (left, right) per row
    unsigned o0, o1;
    float wl, wr, r0, r1;
};
__device__ __forceinline__ int med3_i32(int x, int hi) {   // clamp to [0, hi], hi >= 0 and wave-uniform: one instruction (hipcc emits max + min)
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
// make_taps2's values (same operations on ix / iy), border cases decided on the integer column / row
__device__ __forceinline__ TapsU8 make_taps_u8(float wx0, float wx1, float fx, float wy0, float wy1, float fy, int H, int W, unsigned W3) {
    const int x0 = (int)fx, y0 = (int)fy, y1 = y0 + 1;
    const bool in = (unsigned)x0 <= (unsigned)(W - 2), m1 = x0 == -1, p1 = x0 == W - 1;
    TapsU8 t;
    t.wl = in ? wx0 : (m1 ? wx1 : 0.f), t.wr = in ? wx1 : (p1 ? wx0 : 0.f);
    t.r0 = (unsigned)y0 < (unsigned)H ? wy0 : 0.f, t.r1 = (unsigned)y1 < (unsigned)H ? wy1 : 0.f;
    const unsigned xs = (unsigned)med3_i32(x0, W - 2), xs3 = xs + 2u * xs;
    t.o0 = __umul24((unsigned)med3_i32(y0, H - 1), W3) + xs3, t.o1 = __umul24((unsigned)med3_i32(y1, H - 1), W3) + xs3;   // (rows, 3 W < 2^24)
    return t;
}
template <int FMODE, bool SWAP, bool PKU8>   // FMODE: how a lane gets its field values -- 0 four corners per pixel, 1 a 3-column window per lane, 2 the window through the wave
__global__ void __launch_bounds__(256) upsample_grid_sample_u8_kernel(const unsigned char *__restrict__ input,
                                                                      const float *__restrict__ field,
                                                                      unsigned char *__restrict__ out, int H, int W, int fh, int fw,
                                                                      float ry, float rx, unsigned nrg, float kx, float kx1, float ky, float ky1) {
    // grid = (row groups padded to a multiple of 8, 256-pixel chunks of a row, samples): the XCD of a workgroup is blockIdx.x % 8 whatever
    // y and z are, so xcd_remap on x alone keeps consecutive row groups (shared source rows) on one L2.  A wave = one output row's chunk:
    // sample, row, the field's row pair and its vertical weights are wave-uniform -- no division anywhere, scalar base addresses.
    // (Several rows per wave, one after the other or as a software pipeline over the rows, bought nothing: measured, docs/ROUNDS.md.)
    const unsigned rg = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned oy = __builtin_amdgcn_readfirstlane(rg * 4u + (threadIdx.x >> 6));
    const unsigned n = blockIdx.z;
    if (rg >= nrg || oy >= (unsigned)H) return;
    const int ox0 = (int)(blockIdx.y * 256u + (threadIdx.x & 63u) * 4u);
    if (FMODE != 2 && ox0 >= W) return;   // (FMODE 2: every lane fetches and serves field columns first)
    const unsigned row = n * (unsigned)H + oy;
    const float sy = mul_rounded(ry, (float)oy);
    const int y0 = __builtin_amdgcn_readfirstlane((int)sy), y1 = y0 + (y0 < fh - 1 ? 1 : 0);
    const float ly = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy - (float)y0))), hy = 1.f - ly;
    const float2 *f0 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y0) * fw;
    const float2 *f1 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y1) * fw;
    const f32x2 ly2 = pk_splat(ly), hy2 = pk_splat(hy), one2 = pk_splat(1.f);
    const float oxf = (float)ox0;   // (exact, and so are oxf + 1 .. 3: W < 2^24)
    f32x2 gx[2], gy[2];
    if constexpr (FMODE != 0) {
        f32x2 sx[2], fl[2];
        sx[0] = pk_mul_rounded(pk_splat(rx), f32x2{oxf, oxf + 1.f}), sx[1] = pk_mul_rounded(pk_splat(rx), f32x2{oxf + 2.f, oxf + 3.f});
        fl[0] = pk_floor(sx[0]), fl[1] = pk_floor(sx[1]);   // sx >= 0: floor == the float kernel's (float)(int)sx
        const unsigned cb = (unsigned)(int)fl[0].x;         // first field column this lane can touch; it needs cb .. cb + 2 at most
        if constexpr (FMODE == 2) {
            // the wave's 256 pixels span <= 62 field columns: lane L fetches column c0 + L of the row pair (2 memory instructions instead
            // of 6 -- the launch is bound by their number, see the gathers below) and every pixel takes its four corners from the lanes that
            // hold them (lane (x0 - c0) holds column x0, lane (x0 - c0 + 1) column min(x0 + 1, fw - 1): the values the per-lane loads return)
            const unsigned c0 = __builtin_amdgcn_readfirstlane(cb);   // (lane 0 has the wave's smallest column)
            const unsigned mine = min(c0 + (threadIdx.x & 63u), (unsigned)(fw - 1));
            const float2 w0 = f0[mine], w1 = f1[mine];
            const int w0x = __builtin_bit_cast(int, w0.x), w0y = __builtin_bit_cast(int, w0.y), w1x = __builtin_bit_cast(int, w1.x), w1y = __builtin_bit_cast(int, w1.y);
            auto take = [](int addr, int v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, v)); };
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x2 lx = sx[q] - fl[q], hx = one2 - lx;
                const int j0 = ((int)fl[q].x - (int)c0) * 4, j1 = ((int)fl[q].y - (int)c0) * 4;
                gx[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{take(j0, w0x), take(j1, w0x)}, f32x2{take(j0 + 4, w0x), take(j1 + 4, w0x)},
                                      f32x2{take(j0, w1x), take(j1, w1x)}, f32x2{take(j0 + 4, w1x), take(j1 + 4, w1x)});
                gy[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{take(j0, w0y), take(j1, w0y)}, f32x2{take(j0 + 4, w0y), take(j1 + 4, w0y)},
                                      f32x2{take(j0, w1y), take(j1, w1y)}, f32x2{take(j0 + 4, w1y), take(j1 + 4, w1y)});
            }
            if (ox0 >= W) return;
        } else {
            float2 r0[3], r1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned col = min(cb + (unsigned)k, (unsigned)(fw - 1));
                r0[k] = f0[col], r1[k] = f1[col];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x2 lx = sx[q] - fl[q], hx = one2 - lx;
                // a pixel whose cell is the lane's second one reads columns (1, 2) of the window (pixel 0 never is)
                const bool s0 = q == 0 ? false : fl[q].x > fl[0].x, s1 = fl[q].y > fl[0].x;
                const float2 a0 = s0 ? r0[1] : r0[0], b0 = s0 ? r0[2] : r0[1], c0 = s0 ? r1[1] : r1[0], d0 = s0 ? r1[2] : r1[1];
                const float2 a1 = s1 ? r0[1] : r0[0], b1 = s1 ? r0[2] : r0[1], c1 = s1 ? r1[1] : r1[0], d1 = s1 ? r1[2] : r1[1];
                gx[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{a0.x, a1.x}, f32x2{b0.x, b1.x}, f32x2{c0.x, c1.x}, f32x2{d0.x, d1.x});
                gy[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{a0.y, a1.y}, f32x2{b0.y, b1.y}, f32x2{c0.y, c1.y}, f32x2{d0.y, d1.y});
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 sx = pk_mul_rounded(pk_splat(rx), f32x2{oxf + (float)(2 * q), oxf + (float)(2 * q + 1)});
            const f32x2 fl = pk_floor(sx), lx = sx - fl, hx = one2 - lx;
            const unsigned xa = (unsigned)(int)fl.x, xb = (unsigned)(int)fl.y;
            const unsigned xa1 = xa + ((int)xa < fw - 1 ? 1u : 0u), xb1 = xb + ((int)xb < fw - 1 ? 1u : 0u);
            const float2 a0 = f0[xa], b0 = f0[xa1], c0 = f1[xa], d0 = f1[xa1];
            const float2 a1 = f0[xb], b1 = f0[xb1], c1 = f1[xb], d1 = f1[xb1];
            gx[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{a0.x, a1.x}, f32x2{b0.x, b1.x}, f32x2{c0.x, c1.x}, f32x2{d0.x, d1.x});
            gy[q] = pk_field_lerp(hy2, ly2, hx, lx, f32x2{a0.y, a1.y}, f32x2{b0.y, b1.y}, f32x2{c0.y, c1.y}, f32x2{d0.y, d1.y});
        }
    }
    // un-normalisation (unnormalize above: one fma) and the tap weights, a pixel pair at a time
    // (kx, kx1, ky, ky1: unnormalize()'s constants 0.5 * size or 0.5 * (size - 1), the same fp32 products formed by the launcher)
    const unsigned W3 = 3u * (unsigned)W;
    const unsigned char *ip = input + (size_t)n * H * W3;
    unsigned ulo[4], uhi[4], vlo[4], vhi[4], off[2][4];
    f32x2 wa0[2], wb0[2], wa1[2], wb1[2];
    f32x2 fx[2], fy[2], wx0[2], wx1[2], wy0[2], wy1[2];
    int x0[4], y0i[4];
    bool inner = true;   // this lane's 4 pixels have all four taps inside the frame
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 ix = pk_fma(gx[q], pk_splat(kx), pk_splat(kx1)), iy = pk_fma(gy[q], pk_splat(ky), pk_splat(ky1));
        fx[q] = pk_floor(ix), fy[q] = pk_floor(iy);
        wx1[q] = ix - fx[q], wx0[q] = one2 - wx1[q], wy1[q] = iy - fy[q], wy0[q] = one2 - wy1[q];
        x0[2 * q] = (int)fx[q].x, x0[2 * q + 1] = (int)fx[q].y, y0i[2 * q] = (int)fy[q].x, y0i[2 * q + 1] = (int)fy[q].y;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            inner = inner && (unsigned)x0[2 * q + e] <= (unsigned)(W - 2) && (unsigned)y0i[2 * q + e] <= (unsigned)(H - 2);
    }
    // Border logic only where a wave needs it (wave-uniform branch): with every tap of every lane inside the frame the weights are the plain
    // products and the offsets need no clamps -- the same values make_taps_u8 returns for such pixels, 12 instructions per pixel fewer.
    if (H >= 2 && __builtin_amdgcn_ballot_w64(!inner) == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wa0[q] = wx0[q] * wy0[q], wb0[q] = wx1[q] * wy0[q], wa1[q] = wx0[q] * wy1[q], wb1[q] = wx1[q] * wy1[q];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned x = (unsigned)x0[2 * q + e];
                off[0][2 * q + e] = __umul24((unsigned)y0i[2 * q + e], W3) + (x + 2u * x);
                off[1][2 * q + e] = off[0][2 * q + e] + W3;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const TapsU8 ta = make_taps_u8(wx0[q].x, wx1[q].x, fx[q].x, wy0[q].x, wy1[q].x, fy[q].x, H, W, W3);
            const TapsU8 tb = make_taps_u8(wx0[q].y, wx1[q].y, fx[q].y, wy0[q].y, wy1[q].y, fy[q].y, H, W, W3);
            off[0][2 * q] = ta.o0, off[1][2 * q] = ta.o1, off[0][2 * q + 1] = tb.o0, off[1][2 * q + 1] = tb.o1;
            const f32x2 wl = {ta.wl, tb.wl}, wr = {ta.wr, tb.wr}, r0 = {ta.r0, tb.r0}, r1 = {ta.r1, tb.r1};
            wa0[q] = wl * r0, wb0[q] = wr * r0, wa1[q] = wl * r1, wb1[q] = wr * r1;
        }
    }
    // The two horizontally adjacent source pixels of a row are 6 consecutive bytes at any byte offset: ONE aligned 12-byte load (the dword the
    // pair starts in and the two behind it) + two v_alignbyte.  The launch is bound by its vector MEMORY instructions, not by bytes: a
    // 4-byte + a 2-byte unaligned load per pair (23 memory instructions per lane) ran 35 us whether the arithmetic took 613 or 309
    // instructions; one unaligned 8-byte load per pair 27 us; aligned loads cost about half an unaligned one (timing-only ablation: -5 us).
    // Up to 8 bytes behind the pair are read and not used: only for the last THREE pixel pairs of the last row of the LAST sample are they
    // outside the caller's buffer, so a wave of that sample in which some lane reads there takes exact 4 + 2-byte loads instead
    // (wave-uniform branch; the other samples' waves do not even test).
    bool exact = false;
    if (n + 1 == gridDim.z) {
        unsigned m = off[1][0];
#pragma unroll
        for (int i = 1; i < 4; ++i) m = max(m, off[1][i]);   // (row 1 of a pixel is never above its row 0)
        exact = __builtin_amdgcn_ballot_w64((m & ~3u) + 12u > (unsigned)H * W3) != 0;
    }
    if (!exact) {
        unsigned a[2][4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const U32x3 t = *reinterpret_cast<const U32x3 *>(ip + (off[r][i] & ~3u));
                a[r][i][0] = t.x, a[r][i][1] = t.y, a[r][i][2] = t.z;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (v_alignbyte shifts by the low two bits of its third operand)
            ulo[i] = __builtin_amdgcn_alignbyte(a[0][i][1], a[0][i][0], off[0][i]), uhi[i] = __builtin_amdgcn_alignbyte(a[0][i][2], a[0][i][1], off[0][i]);
            vlo[i] = __builtin_amdgcn_alignbyte(a[1][i][1], a[1][i][0], off[1][i]), vhi[i] = __builtin_amdgcn_alignbyte(a[1][i][2], a[1][i][1], off[1][i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ulo[i] = reinterpret_cast<const U8x4 *>(ip + off[0][i])->v, uhi[i] = reinterpret_cast<const U8x2 *>(ip + off[0][i] + 4)->v;
            vlo[i] = reinterpret_cast<const U8x4 *>(ip + off[1][i])->v, vhi[i] = reinterpret_cast<const U8x2 *>(ip + off[1][i] + 4)->v;
        }
    }
    // blend: (ulo, uhi) = bytes 0..3 and 4..5 of the pixel pair of row 0, (vlo, vhi) of row 1; source pixel x = bytes 0..2, x + 1 = bytes 3..5.
    // One explicit fma chain per value: the same rounding in every instantiation and in the float kernel (a byte flips where the blend lands
    // on an integer).
    unsigned ow[3] = {0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int ci = SWAP ? 2 - c : c;   // compile-time
            const int i = 2 * q, j = 2 * q + 1;
            const f32x2 ux = {(float)((ulo[i] >> (8 * ci)) & 0xffu), (float)((ulo[j] >> (8 * ci)) & 0xffu)};
            const f32x2 vx = {(float)((vlo[i] >> (8 * ci)) & 0xffu), (float)((vlo[j] >> (8 * ci)) & 0xffu)};
            const f32x2 uy = {ci == 0 ? (float)(ulo[i] >> 24) : (float)((uhi[i] >> (8 * (ci - 1))) & 0xffu),
                              ci == 0 ? (float)(ulo[j] >> 24) : (float)((uhi[j] >> (8 * (ci - 1))) & 0xffu)};
            const f32x2 vy = {ci == 0 ? (float)(vlo[i] >> 24) : (float)((vhi[i] >> (8 * (ci - 1))) & 0xffu),
                              ci == 0 ? (float)(vlo[j] >> 24) : (float)((vhi[j] >> (8 * (ci - 1))) & 0xffu)};
            const f32x2 r = pk_fma(vy, wb1[q], pk_fma(vx, wa1[q], pk_fma(uy, wb0[q], ux * wa0[q])));
            const int bi = i * 3 + c, bj = j * 3 + c;   // byte positions inside the lane's 12 output bytes (compile-time)
            if constexpr (PKU8) {
                cvt_u8_into(bi & 3, ow[bi >> 2], r.x), cvt_u8_into(bj & 3, ow[bj >> 2], r.y);
            } else {
                ow[bi >> 2] |= (unsigned)min(max((int)r.x, 0), 255) << (8 * (bi & 3));
                ow[bj >> 2] |= (unsigned)min(max((int)r.y, 0), 255) << (8 * (bj & 3));
            }
        }
    }
    unsigned *op = reinterpret_cast<unsigned *>(out + ((size_t)row * W) * 3) + 3u * ((unsigned)ox0 >> 2);  // 12 bytes, 4-byte aligned (W % 4 == 0)
    op[0] = ow[0], op[1] = ow[1], op[2] = ow[2];
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace pws

using namespace pws;

extern "C" int pws_grid_sample_fwd(const float *input, const float *grid, float *out, int n, int c, int h, int w, int ho,
                                   int wo, int align_corners, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "pws_grid_sample_fwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(input && grid && out, "pws_grid_sample_fwd: NULL pointer");
    PWS_REQUIRE((size_t)h * w < (1u << 31) && (size_t)ho * wo < (1u << 31), "pws_grid_sample_fwd: plane too large");
    const size_t total = (size_t)n * ho * wo;
    const int howo = ho * wo;
    // algorithmic traffic: field 8 B + out 4*C B per output pixel, frame 4*C B per input pixel; ~30 flop/px/channel
    ProfScope prof(KID_GRID_SAMPLE_FWD, (double)total * (14.0 + 8.0 * c),
                   (double)total * (8.0 + 4.0 * c) + 4.0 * c * (double)n * h * w, as_stream(stream));
    if (howo % 4 == 0 && aligned16(grid) && aligned16(out) && w >= 2) {
        const size_t groups = total / 4;
        const unsigned nb = (unsigned)((groups + 255) / 256);
        // streaming hints once the launch no longer fits the 256 MB Infinity Cache (see the kernel's comment)
        // (experiments 2..4 are this file's A/B switches: 2 / 3 = never non-temporal; every OTHER value -- other kernels' A/B numbers -- keeps the product rule)
        const bool nt = g_experiment == 1 || (!(g_experiment >= 2 && g_experiment <= 4) && (double)total * (8.0 + 8.0 * c) > 256e6);
        if (g_experiment == 5 && nt)   // row window + wave shuffle (A/B only, tools/gs_shuffle_ab.py)
            hipLaunchKernelGGL((grid_sample_fwd2_kernel<4, true, true>), dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h,
                               w, howo, groups, nb, align_corners);
        else if (g_experiment == 5)
            hipLaunchKernelGGL((grid_sample_fwd2_kernel<4, false, true>), dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h,
                               w, howo, groups, nb, align_corners);
        else if (nt)
            hipLaunchKernelGGL((grid_sample_fwd2_kernel<4, true>), dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h,
                               w, howo, groups, nb, align_corners);
        else
            hipLaunchKernelGGL((grid_sample_fwd2_kernel<4, false>), dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h,
                               w, howo, groups, nb, align_corners);
    } else if (w >= 2) {
        const unsigned nb = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL((grid_sample_fwd2_kernel<1, false>), dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h, w,
                           howo, total, nb, align_corners);
    } else if (howo % 4 == 0 && aligned16(grid) && aligned16(out)) {
        const size_t groups = total / 4;
        const unsigned nb = (unsigned)((groups + 255) / 256);
        hipLaunchKernelGGL(grid_sample_fwd_kernel<4>, dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h, w,
                           howo, groups, nb, align_corners);
    } else {
        const unsigned nb = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(grid_sample_fwd_kernel<1>, dim3(nb), dim3(256), 0, as_stream(stream), input, grid, out, c, h, w,
                           howo, total, nb, align_corners);
    }
    return check_launch("grid_sample_fwd_kernel");
}

extern "C" int pws_grid_sample_bwd(const float *gout, const float *input, const float *grid, float *ginput, float *ggrid,
                                   int n, int c, int h, int w, int ho, int wo, int align_corners, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "pws_grid_sample_bwd: bad shape");
    if (n == 0 || (!ginput && !ggrid)) return PWS_OK;
    PWS_REQUIRE(gout && input && grid, "pws_grid_sample_bwd: NULL pointer");
    PWS_REQUIRE((size_t)h * w < (1u << 31) && (size_t)ho * wo < (1u << 31), "pws_grid_sample_bwd: plane too large");
    if (ginput) {
        hipError_t e = hipMemsetAsync(ginput, 0, sizeof(float) * (size_t)n * c * h * w, as_stream(stream));
        if (e != hipSuccess) {
            set_error("pws_grid_sample_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
    }
    const size_t total = (size_t)n * ho * wo;
    const unsigned nb = (unsigned)((total + 255) / 256);
    ProfScope prof(KID_GRID_SAMPLE_BWD, (double)total * (20.0 + 16.0 * c),
                   (double)total * (8.0 + 4.0 * c + (ggrid ? 8.0 : 0.0)) + (ginput ? 8.0 : 4.0) * c * (double)n * h * w,
                   as_stream(stream));
    if (!ginput && (ho * wo) % 4 == 0 && w >= 2 && aligned16(grid) && aligned16(gout) && aligned16(ggrid) && g_experiment != 2) {
        // the field gradient alone: 4 pixels per lane, vector loads, paired gathers (see the kernel)
        const size_t groups = total / 4;
        const unsigned nbg = (unsigned)((groups + 255) / 256);
        if ((double)total * (16.0 + 8.0 * c) > 256e6)   // beyond the Infinity Cache: streaming hints, as the forward
            hipLaunchKernelGGL(grid_sample_bwd_field_kernel<true>, dim3(nbg), dim3(256), 0, as_stream(stream), gout, input, grid, ggrid, c,
                               h, w, ho * wo, groups, nbg, align_corners);
        else
            hipLaunchKernelGGL(grid_sample_bwd_field_kernel<false>, dim3(nbg), dim3(256), 0, as_stream(stream), gout, input, grid, ggrid,
                               c, h, w, ho * wo, groups, nbg, align_corners);
        return check_launch("grid_sample_bwd_field_kernel");
    }
    hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(nb), dim3(256), 0, as_stream(stream), gout, input, grid, ginput, ggrid, c,
                       h, w, ho * wo, total, nb, align_corners);
    return check_launch("grid_sample_bwd_kernel");
}

extern "C" int pws_affine_grid(const float *theta, float *grid, int n, int h, int w, int align_corners,
                               pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0, "pws_affine_grid: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(theta && grid, "pws_affine_grid: NULL pointer");
    const size_t total = (size_t)n * h * w;
    ProfScope prof(KID_AFFINE_GRID, 0.0, 8.0 * (double)total, as_stream(stream));
    hipLaunchKernelGGL(affine_grid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), theta, grid,
                       h, w, total, align_corners);
    return check_launch("affine_grid_kernel");
}

extern "C" int pws_upsample_bilinear_ac(const float *in, float *out, int n, int c, int h, int w, int ho, int wo,
                                        pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "pws_upsample_bilinear_ac: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(in && out, "pws_upsample_bilinear_ac: NULL pointer");
    const float ry = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    const size_t total = (size_t)n * c * ho * wo;
    ProfScope prof(KID_UPSAMPLE, 0.0, 4.0 * (double)total + 4.0 * (double)n * c * h * w, as_stream(stream));
    hipLaunchKernelGGL(upsample_bilinear_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in,
                       out, h, w, ho, wo, ry, rx, total);
    return check_launch("upsample_bilinear_ac_kernel");
}

extern "C" int pws_upsample_bilinear_ac_bwd(const float *gout, float *gin, int n, int c, int h, int w, int ho, int wo,
                                            pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "pws_upsample_bilinear_ac_bwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(gout && gin, "pws_upsample_bilinear_ac_bwd: NULL pointer");
    const float ry = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f;
    const float rx = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    const size_t total = (size_t)n * c * h * w;
    ProfScope prof(KID_UPSAMPLE, 0.0, 4.0 * (double)total + 4.0 * (double)n * c * ho * wo, as_stream(stream));
    hipLaunchKernelGGL(upsample_bilinear_ac_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       gout, gin, h, w, ho, wo, ry, rx, total);
    return check_launch("upsample_bilinear_ac_bwd_kernel");
}

extern "C" int pws_affine_grid_bwd(const float *ggrid, float *gtheta, int n, int h, int w, int align_corners,
                                   pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0, "pws_affine_grid_bwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(ggrid && gtheta, "pws_affine_grid_bwd: NULL pointer");
    ProfScope prof(KID_AFFINE_GRID, 0.0, 8.0 * (double)n * h * w, as_stream(stream));
    hipLaunchKernelGGL(affine_grid_bwd_kernel, dim3((unsigned)n), dim3(256), 0, as_stream(stream), ggrid, gtheta, h, w,
                       align_corners);
    return check_launch("affine_grid_bwd_kernel");
}

extern "C" int pws_upsample_grid_sample_fwd(const float *input, const float *field, float *out, int n, int c, int h, int w,
                                            int fh, int fw, int align_corners, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && fh > 0 && fw > 0, "pws_upsample_grid_sample_fwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(input && field && out, "pws_upsample_grid_sample_fwd: NULL pointer");
    PWS_REQUIRE((size_t)h * w < (1u << 31) && w >= 2, "pws_upsample_grid_sample_fwd: plane too large or w < 2");
    const float ry = h > 1 ? (float)(fh - 1) / (float)(h - 1) : 0.f;
    const float rx = w > 1 ? (float)(fw - 1) / (float)(w - 1) : 0.f;
    const size_t total = (size_t)n * h * w;
    // fused 720p path: field read once (fh*fw*8 B), frame in + out 4*C B per pixel each
    ProfScope prof(KID_UPSAMPLE_GRID_SAMPLE_FWD, (double)total * (40.0 + 8.0 * c),
                   (double)total * 8.0 * c + 8.0 * (double)n * fh * fw, as_stream(stream));
    if (w % 4 == 0 && aligned16(out)) {
        const size_t groups = total / 4;
        const unsigned nb = (unsigned)((groups + 255) / 256);
        const bool nt = g_experiment == 1 || (!(g_experiment >= 2 && g_experiment <= 5) && (double)total * 8.0 * c > 256e6);
        if (3.f * rx < 0.999f && nt)
            hipLaunchKernelGGL((upsample_grid_sample_fwd_kernel<4, true, true>), dim3(nb), dim3(256), 0, as_stream(stream), input,
                               field, out, c, h, w, fh, fw, ry, rx, groups, nb, align_corners);
        else if (3.f * rx < 0.999f)
            hipLaunchKernelGGL((upsample_grid_sample_fwd_kernel<4, true, false>), dim3(nb), dim3(256), 0, as_stream(stream), input,
                               field, out, c, h, w, fh, fw, ry, rx, groups, nb, align_corners);
        else
            hipLaunchKernelGGL((upsample_grid_sample_fwd_kernel<4, false, false>), dim3(nb), dim3(256), 0, as_stream(stream), input,
                               field, out, c, h, w, fh, fw, ry, rx, groups, nb, align_corners);
    } else {
        const unsigned nb = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL((upsample_grid_sample_fwd_kernel<1, false, false>), dim3(nb), dim3(256), 0, as_stream(stream), input, field,
                           out, c, h, w, fh, fw, ry, rx, total, nb, align_corners);
    }
    return check_launch("upsample_grid_sample_fwd_kernel");
}

extern "C" int pws_upsample_grid_sample_u8(const unsigned char *frame_hwc, const float *field, unsigned char *out_hwc, int n, int h,
                                           int w, int fh, int fw, int swap_rb, int align_corners, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && fh > 0 && fw > 0, "pws_upsample_grid_sample_u8: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(frame_hwc && field && out_hwc, "pws_upsample_grid_sample_u8: NULL pointer");
    PWS_REQUIRE((size_t)h * w < (1u << 29) && w >= 2 && w % 4 == 0, "pws_upsample_grid_sample_u8: w must be a multiple of 4 (got %d)", w);
    PWS_REQUIRE((reinterpret_cast<size_t>(out_hwc) & 3) == 0, "pws_upsample_grid_sample_u8: out must be 4-byte aligned");
    const float ry = h > 1 ? (float)(fh - 1) / (float)(h - 1) : 0.f;
    const float rx = w > 1 ? (float)(fw - 1) / (float)(w - 1) : 0.f;
    const size_t total = (size_t)n * h * w;
    const unsigned cpr = (unsigned)((w + 255) / 256);                      // 256-pixel chunks per output row: one wave each
    const unsigned nrg = (unsigned)((h + 3) / 4), gx = (nrg + 7u) & ~7u;   // row groups of 4 (one row per wave), padded: see the kernel
    PWS_REQUIRE(n < 65536 && cpr < 65536 && (size_t)w * 3 < (1u << 24) && h < (1 << 24), "pws_upsample_grid_sample_u8: frame or batch too large");
    ProfScope prof(KID_UPSAMPLE_GRID_SAMPLE_U8, (double)total * (40.0 + 8.0 * 3), (double)total * 6.0 + 8.0 * (double)n * fh * fw,
                   as_stream(stream));
    // PWS_OPT_EXPERIMENT 4: float -> byte through (int) + clamp + shift/or instead of v_cvt_pk_u8_f32 (A/B and the equality test).
    // (Rounds 2-5 kept a row-window + wave-shuffle variant here, north_star's "wavefront shuffles for the bilinear gather": 16 -> 2 memory
    // instructions per lane bought nothing, 33.6 vs 27.2 us on a pure translation -- the kernel is bound by its vector instructions;
    // docs/ROUNDS.md.  The float kernel's variant stays: grid_sample_fwd2_kernel<.., ROWWIN>, tools/gs_shuffle_ab.py.)
    const bool pku8 = g_experiment != 4;
    const float kx = align_corners ? 0.5f * (float)(w - 1) : 0.5f * (float)w, kx1 = 0.5f * (float)(w - 1);
    const float ky = align_corners ? 0.5f * (float)(h - 1) : 0.5f * (float)h, ky1 = 0.5f * (float)(h - 1);
    const bool narrow = 3.f * rx < 0.999f;
    // field access (the kernel's FMODE): 3 * rx < 1 -> a lane's 4 pixels touch <= 3 field columns; 255 * rx + 3 <= 63 -> a wave's 256 pixels touch
    // <= 63: the wave fetches them once (1280 from 256 columns: rx = 0.199).  PWS_OPT_EXPERIMENT 49: per-lane windows (A/B)
    const int fmode = !narrow ? 0 : (255.f * rx + 3.f <= 63.f && g_experiment != 49 ? 2 : 1);
#define PWS_U8_LAUNCH_F(FMODE_, SWAP_, PK_)                                                                                             \
    hipLaunchKernelGGL((upsample_grid_sample_u8_kernel<FMODE_, SWAP_, PK_>), dim3(gx, cpr, (unsigned)n), dim3(256), 0, as_stream(stream), \
                       frame_hwc, field, out_hwc, h, w, fh, fw, ry, rx, nrg, kx, kx1, ky, ky1)
#define PWS_U8_LAUNCH(SWAP_, PK_)                                                                                                       \
    do {                                                                                                                                \
        if (fmode == 2) PWS_U8_LAUNCH_F(2, SWAP_, PK_);                                                                                 \
        else if (fmode == 1) PWS_U8_LAUNCH_F(1, SWAP_, PK_);                                                                            \
        else PWS_U8_LAUNCH_F(0, SWAP_, PK_);                                                                                            \
    } while (0)
    if (pku8) {
        if (swap_rb) PWS_U8_LAUNCH(true, true);
        else PWS_U8_LAUNCH(false, true);
    } else {
        if (swap_rb) PWS_U8_LAUNCH(true, false);
        else PWS_U8_LAUNCH(false, false);
    }
#undef PWS_U8_LAUNCH_F
#undef PWS_U8_LAUNCH
    return check_launch("upsample_grid_sample_u8_kernel");
}

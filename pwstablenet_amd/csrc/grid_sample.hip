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
struct __attribute__((packed, aligned(1))) U8x4 {
    unsigned v;
};
struct __attribute__((packed, aligned(1))) U8x2 {
    unsigned short v;
};
struct __attribute__((packed, aligned(1))) U8x16 {
    unsigned w[4];
};
// ROWWIN (the "wavefront shuffle" variant, measured against the per-tap gathers: DESIGN.md): where a stabiliser's field maps the 4
// consecutive output pixels of a lane onto at most 5 consecutive pixels of ONE source row pair, the lane fetches each row as ONE
// unaligned 16-byte window (5.33 RGB pixels) -- plus, for the sixth pixel, the second dword of the NEXT lane's window through a
// wave shuffle when that lane's window starts exactly 4 pixels further -- instead of 4 x (4 + 2)-byte gathers per row: 2 vector
// memory instructions per lane instead of 16.  Lanes whose pixels straddle a row or spread wider keep the per-tap gathers.
// SWAP (R <-> B) is a template parameter and a pixel pair travels as two 32-bit words: every source byte is then a CONSTANT byte of a
// dword and converts with one v_cvt_f32_ubyteN.  (Round 2 carried the six bytes as a 64-bit value shifted by a run-time amount:
// 330 of the kernel's 897 vector instructions per lane were 64-bit shifts, masks and u64 -> float conversions.)
template <bool NARROW, bool ROWWIN, bool SWAP>
__global__ void __launch_bounds__(256) upsample_grid_sample_u8_kernel(const unsigned char *__restrict__ input,
                                                                      const float *__restrict__ field,
                                                                      unsigned char *__restrict__ out, int H, int W, int fh, int fw,
                                                                      float ry, float rx, size_t total_groups, unsigned nblocks,
                                                                      int ac) {
    constexpr int PPT = 4;
    const unsigned blk = xcd_remap(blockIdx.x, nblocks);
    const size_t gidx_raw = (size_t)blk * 256 + threadIdx.x;
    const bool live = gidx_raw < total_groups;   // lanes past the end keep running (on the last group): the wave shuffles below want every lane
    const size_t gidx = live ? gidx_raw : total_groups - 1;
    const int HW = H * W;
    const size_t p0 = gidx * PPT;
    const int n = (int)(p0 / HW);
    const int hw = (int)(p0 % HW);
    const int oy = hw / W, ox0 = hw % W;  // W % 4 == 0: the group stays inside one row
    const float sy = mul_rounded(ry, (float)oy);
    const int y0 = (int)sy, y1 = y0 + (y0 < fh - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly;
    const float2 *f0 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y0) * fw;
    const float2 *f1 = reinterpret_cast<const float2 *>(field) + ((size_t)n * fh + y1) * fw;
    Taps2 t[PPT];
    if constexpr (NARROW) {
        const int cb = (int)mul_rounded(rx, (float)ox0);
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
            const bool second = x0 > cb;
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
    const unsigned char *ip = input + (size_t)n * HW * 3;
    unsigned char res[PPT * 3];
    // (ulo, uhi): bytes 0..3 and 4..5 of the pixel pair of row 0; (vlo, vhi): row 1.  Pixel x = bytes 0..2, pixel x + 1 = bytes 3..5.
    auto blend = [&](int i, unsigned ulo, unsigned uhi, unsigned vlo, unsigned vhi) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int ci = SWAP ? 2 - c : c;   // compile-time
            const float ux = (float)((ulo >> (8 * ci)) & 0xffu), vx = (float)((vlo >> (8 * ci)) & 0xffu);
            const float uy = ci == 0 ? (float)(ulo >> 24) : (float)((uhi >> (8 * (ci - 1))) & 0xffu);
            const float vy = ci == 0 ? (float)(vlo >> 24) : (float)((vhi >> (8 * (ci - 1))) & 0xffu);
            // one explicit fma chain: the same rounding in every instantiation and on both paths (left to the compiler, the
            // contraction of this sum differed between them -- a byte flips where the blend lands on an integer)
            const float r = fmaf(vy, t[i].b1, fmaf(vx, t[i].a1, fmaf(uy, t[i].b0, ux * t[i].a0)));
            res[i * 3 + c] = (unsigned char)min(max((int)r, 0), 255);
        }
    };
    bool fast = false;
    if constexpr (ROWWIN) {
        int lo = t[0].xs, hi_ = t[0].xs;
        bool same = true;
#pragma unroll
        for (int i = 1; i < PPT; ++i) {
            lo = min(lo, t[i].xs), hi_ = max(hi_, t[i].xs);
            same = same && t[i].r0 == t[0].r0 && t[i].r1 == t[0].r1;
        }
        // the window must not run past the frame buffer: 16 (+4 shuffled) bytes from the window start
        const size_t row_end = (size_t)HW * 3;
        const bool in_buf = ((size_t)t[0].r0 * W + lo) * 3 + 16 <= row_end && ((size_t)t[0].r1 * W + lo) * 3 + 16 <= row_end;
        const int span = hi_ - lo;   // the pair of the last pixel ends at column lo + span + 1: bytes up to 3 (span + 2)
        // neighbour lane's window: its second dword = bytes 16..19 of this lane's rows when it starts exactly 4 pixels further
        const int nb_lo = __shfl_down(lo, 1, 64), nb_r0 = __shfl_down(t[0].r0, 1, 64), nb_r1 = __shfl_down(t[0].r1, 1, 64);
        const bool nb_same = __shfl_down((int)(same && in_buf), 1, 64) != 0;
        const bool nb_ok = (threadIdx.x & 63) != 63 && nb_same && nb_lo == lo + 4 && nb_r0 == t[0].r0 && nb_r1 == t[0].r1;
        const bool want = same && in_buf && (span <= 3 || (span == 4 && nb_ok));
        // every lane that could be a window lane loads (the shuffle needs the neighbour's dwords whatever its own verdict)
        unsigned w0[5] = {0u, 0u, 0u, 0u, 0u}, w1[5] = {0u, 0u, 0u, 0u, 0u};
        if (same && in_buf) {
            const U8x16 a = *reinterpret_cast<const U8x16 *>(ip + ((size_t)t[0].r0 * W + lo) * 3);
            const U8x16 b = *reinterpret_cast<const U8x16 *>(ip + ((size_t)t[0].r1 * W + lo) * 3);
#pragma unroll
            for (int k = 0; k < 4; ++k) w0[k] = a.w[k], w1[k] = b.w[k];
        }
        w0[4] = __shfl_down(w0[1], 1, 64), w1[4] = __shfl_down(w1[1], 1, 64);
        fast = want;
        if (fast) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int d = (t[i].xs - lo) * 3;        // byte offset of the pixel pair inside the window: 0, 3, .. 12
                const int k = d >> 2, sh = d & 3;
                // dwords k, k + 1, k + 2 of the (extended) window, selected without dynamic register indexing
                const unsigned a0 = k == 0 ? w0[0] : (k == 1 ? w0[1] : (k == 2 ? w0[2] : w0[3]));
                const unsigned a1 = k == 0 ? w0[1] : (k == 1 ? w0[2] : (k == 2 ? w0[3] : w0[4]));
                const unsigned a2 = k == 0 ? w0[2] : (k == 1 ? w0[3] : w0[4]);
                const unsigned b0 = k == 0 ? w1[0] : (k == 1 ? w1[1] : (k == 2 ? w1[2] : w1[3]));
                const unsigned b1 = k == 0 ? w1[1] : (k == 1 ? w1[2] : (k == 2 ? w1[3] : w1[4]));
                const unsigned b2 = k == 0 ? w1[2] : (k == 1 ? w1[3] : w1[4]);
                blend(i, __builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh), __builtin_amdgcn_alignbyte(b1, b0, sh),
                      __builtin_amdgcn_alignbyte(b2, b1, sh));
            }
        }
    }
    if (!fast) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            // the two horizontally adjacent source pixels of a row are 6 consecutive bytes: one 4-byte + one 2-byte unaligned load
            const unsigned char *q0 = ip + (size_t)t[i].o0 * 3, *q1 = ip + (size_t)t[i].o1 * 3;
            blend(i, reinterpret_cast<const U8x4 *>(q0)->v, reinterpret_cast<const U8x2 *>(q0 + 4)->v, reinterpret_cast<const U8x4 *>(q1)->v,
                  reinterpret_cast<const U8x2 *>(q1 + 4)->v);
        }
    }
    if (!live) return;
    unsigned *op = reinterpret_cast<unsigned *>(out + ((size_t)n * HW + hw) * 3);  // 12 bytes, 4-byte aligned (hw % 4 == 0)
#pragma unroll
    for (int k = 0; k < 3; ++k)
        op[k] = (unsigned)res[4 * k] | ((unsigned)res[4 * k + 1] << 8) | ((unsigned)res[4 * k + 2] << 16) | ((unsigned)res[4 * k + 3] << 24);
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
    const size_t total = (size_t)n * h * w, groups = total / 4;
    const unsigned nb = (unsigned)((groups + 255) / 256);
    ProfScope prof(KID_UPSAMPLE_GRID_SAMPLE_U8, (double)total * (40.0 + 8.0 * 3), (double)total * 6.0 + 8.0 * (double)n * fh * fw,
                   as_stream(stream));
    // PWS_OPT_EXPERIMENT 4: the row-window + wave-shuffle variant, kept for the A/B in DESIGN.md (tools/warp_u8_ab.py).  Measured on 8
    // frames of 1280 x 720 (after the 32-bit byte handling above): 33.6 vs 27.2 us on a pure translation (every lane takes the
    // window), 39.2 vs 34.7 us on a stabiliser's field, 42.6 vs 38.9 us on the random-weight generator's: 16 -> 2 memory instructions
    // per lane buy nothing, the kernel is bound by its vector instructions (613 per lane), the variant's shuffles and selects add to
    // them, and the lanes whose 4 pixels straddle a source row pay both paths.  The per-tap gathers stay the product path.
    const bool rowwin = g_experiment == 4;
    const bool narrow = 3.f * rx < 0.999f;
#define PWS_U8_LAUNCH(NARROW_, ROWWIN_, SWAP_)                                                                                             \
    hipLaunchKernelGGL((upsample_grid_sample_u8_kernel<NARROW_, ROWWIN_, SWAP_>), dim3(nb), dim3(256), 0, as_stream(stream), frame_hwc, field, \
                       out_hwc, h, w, fh, fw, ry, rx, groups, nb, align_corners)
    if (narrow && !rowwin) {
        if (swap_rb) PWS_U8_LAUNCH(true, false, true);
        else PWS_U8_LAUNCH(true, false, false);
    } else if (narrow) {
        if (swap_rb) PWS_U8_LAUNCH(true, true, true);
        else PWS_U8_LAUNCH(true, true, false);
    } else if (!rowwin) {
        if (swap_rb) PWS_U8_LAUNCH(false, false, true);
        else PWS_U8_LAUNCH(false, false, false);
    } else {
        if (swap_rb) PWS_U8_LAUNCH(false, true, true);
        else PWS_U8_LAUNCH(false, true, false);
    }
#undef PWS_U8_LAUNCH
    return check_launch("upsample_grid_sample_u8_kernel");
}

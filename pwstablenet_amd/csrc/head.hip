// The two regression heads of the generator (fp32 VALU kernels; neither is GEMM-shaped enough for MFMA):
//
//  * theta head  (reference lib/networks_cascading.py:148-149,162-163): theta = LReLU(W2 LReLU(W1 vec(x)+b1)+b2),
//    a 4C->hidden GEMV and a hidden->6 GEMV per sample (`flatten` is a 2x2 conv on a 2x2 map, `linear` a 1x1 conv;
//    both are `down` blocks, so theta itself passes through LeakyReLU(0.2)).
//  * field head  (reference :128,174,198,219,235-237): conv3x3 (C->2) + bias, tanh, tanh again, NCHW->NHWC
//    permute, + F.affine_grid(theta) -- fused into one pass that reads the C-channel feature map once
//    (HBM-bound: C*4 B/pixel in, 8-16 B/pixel out) and writes the final N,H,W,2 field.
#include <type_traits>

#include "common.h"

namespace pws {

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// Phase 1: hidden pre-activations, K split over workgroups so that the 4C x hidden weight matrix (2 MB at ngf=64) is
// streamed by many CUs at once.  One wave per (64 hidden units, K slice); lane = hidden unit, so the weight read
// w_flat[k][j0..j0+63] is one coalesced 256-B row per k; up to TH_NB samples share each weight.
// partial[slice][n][hidden] is summed (in slice order: deterministic) by phase 2.
constexpr int TH_NB = 8;

__global__ void __launch_bounds__(64) theta_hidden_kernel(const float *__restrict__ x, int n, int k1, int hidden, int kchunk,
                                                          const float *__restrict__ w_flat, float *__restrict__ partial) {
    extern __shared__ float sx[];  // [TH_NB][kchunk]
    const int lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane;
    const int slice = blockIdx.y;
    const int k0 = slice * kchunk, kn = min(kchunk, k1 - k0);
    // groups of TH_NB samples are spread over blockIdx.z (a training batch of 64 was 8 serial passes of every wave over its weights:
    // 205 us per call in the configs[2] step)
    for (int n0 = blockIdx.z * TH_NB; n0 < n; n0 += gridDim.z * TH_NB) {
        const int nb = min(TH_NB, n - n0);
        __syncthreads();
        for (int i = lane; i < nb * kn; i += 64) sx[(i / kn) * kchunk + i % kn] = x[(size_t)(n0 + i / kn) * k1 + k0 + i % kn];
        __syncthreads();
        float acc[TH_NB];
#pragma unroll
        for (int s = 0; s < TH_NB; ++s) acc[s] = 0.f;
        if (j < hidden) {
            // 8 independent weight loads in flight per lane (one dependent load per k would be latency-bound)
            for (int kb = 0; kb < kn; kb += 8) {
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = w_flat[(size_t)(k0 + (kb + u < kn ? kb + u : kb)) * hidden + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (kb + u < kn) {
#pragma unroll
                        for (int s = 0; s < TH_NB; ++s) acc[s] = fmaf(s < nb ? sx[s * kchunk + kb + u] : 0.f, wv[u], acc[s]);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < TH_NB; ++s)
                if (s < nb) partial[((size_t)slice * n + n0 + s) * hidden + j] = acc[s];
        }
    }
}

// Phase 2: one workgroup per sample: h = LReLU(sum_slices + b1); theta = LReLU(W2 h + b2).
__global__ void __launch_bounds__(256) theta_final_kernel(const float *__restrict__ partial, int nslices, int n, int hidden,
                                                          const float *__restrict__ b_flat, const float *__restrict__ w_lin,
                                                          const float *__restrict__ b_lin, float *__restrict__ theta,
                                                          float *__restrict__ h_saved) {
    extern __shared__ float sh[];  // hidden floats
    const int s = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < hidden; j += 256) {
        float acc = b_flat ? b_flat[j] : 0.f;
        for (int sl = 0; sl < nslices; ++sl) acc += partial[((size_t)sl * n + s) * hidden + j];
        sh[j] = lrelu(acc);
        if (h_saved) h_saved[(size_t)s * hidden + j] = sh[j];
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    for (int o = wv; o < 6; o += 4) {
        float acc = 0.f;
        for (int j = lane; j < hidden; j += 64) acc = fmaf(sh[j], w_lin[(size_t)j * 6 + o], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) theta[(size_t)s * 6 + o] = lrelu(acc + (b_lin ? b_lin[o] : 0.f));
    }
}

// 16x16 output pixels per workgroup, one lane per pixel, channels staged through LDS CH at a time.
constexpr int FH_T = 16, FH_I = FH_T + 2, FH_CH = 16, FH_LDP = FH_CH + 4;

template <bool IO16>
__global__ void __launch_bounds__(256) field_head_kernel(const float *__restrict__ x, int ld, int N, int H, int W, int C,
                                                         const float *__restrict__ w_out, const float *__restrict__ b_out,
                                                         const float *__restrict__ theta, int ac, float *__restrict__ resid,
                                                         float *__restrict__ grid, int tiles_x, int tiles_y, unsigned ntiles, int raw) {
    __shared__ float s_in[FH_I * FH_I * FH_LDP];
    const int tid = threadIdx.x;
    const unsigned tile = xcd_remap(blockIdx.x, ntiles);
    const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int x0 = tx_i * FH_T, y0 = ty_i * FH_T;
    const int tx = tid & 15, ty = tid >> 4;
    const int ch = C < FH_CH ? C : FH_CH;  // channels per stage (C % ch == 0 checked on the host)
    const int c4n = ch / 4;
    float acc0 = b_out ? b_out[0] : 0.f, acc1 = b_out ? b_out[1] : 0.f;
    // the next channel stage is fetched into registers while the current one is consumed out of LDS
    constexpr int NITEM = (FH_I * FH_I * (FH_CH / 4) + 255) / 256;
    float4 pre[NITEM];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tid + it * 256;
            const int pix = item / c4n, c4 = item % c4n;
            const int iy = y0 - 1 + pix / FH_I, ix = x0 - 1 + pix % FH_I;
            const bool ok = item < FH_I * FH_I * c4n && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float4 v = ld4<IO16>(x, ok ? ((size_t)(n * H + iy) * W + ix) * ld + c0 + c4 * 4 : 0);  // unconditional load
            pre[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += ch) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tid + it * 256;
            if (item < FH_I * FH_I * c4n) *reinterpret_cast<float4 *>(s_in + (item / c4n) * FH_LDP + (item % c4n) * 4) = pre[it];
        }
        __syncthreads();
        if (c0 + ch < C) fetch(c0 + ch);
        // The weights are the same for every lane: read through the scalar unit (wave-uniform address -> s_load), not LDS.
        // With them in LDS the kernel was LDS-bandwidth-bound (48 B of ds_read per 8 FMAs; 259 us at N=32, 256x256x64).
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float *ip = s_in + ((ty + tap / 3) * FH_I + tx + tap % 3) * FH_LDP;
            const float *wp = w_out + ((size_t)tap * C + c0) * 2;
            for (int c4 = 0; c4 < c4n; ++c4) {
                const float4 v = *reinterpret_cast<const float4 *>(ip + c4 * 4);
                const float *wq = wp + c4 * 8;  // (c,0),(c,1),(c+1,0),(c+1,1),(c+2,*),(c+3,*)
                acc0 = fmaf(v.x, wq[0], acc0), acc1 = fmaf(v.x, wq[1], acc1);
                acc0 = fmaf(v.y, wq[2], acc0), acc1 = fmaf(v.y, wq[3], acc1);
                acc0 = fmaf(v.z, wq[4], acc0), acc1 = fmaf(v.z, wq[5], acc1);
                acc0 = fmaf(v.w, wq[6], acc0), acc1 = fmaf(v.w, wq[7], acc1);
            }
        }
    }
    const int y = y0 + ty, xq = x0 + tx;
    if (y < H && xq < W && raw) {
        // use_BN training: the pre-normalisation conv output; BatchNorm, tanh(tanh(.)) and the affine add follow in
        // field_bn_finish_kernel once the batch statistics are known
        *reinterpret_cast<float2 *>(resid + (((size_t)n * H + y) * W + xq) * 2) = make_float2(acc0, acc1);
    } else if (y < H && xq < W) {
        const float r0 = tanhf(tanhf(acc0)), r1 = tanhf(tanhf(acc1));
        const size_t p = ((size_t)n * H + y) * W + xq;
        if (resid) *reinterpret_cast<float2 *>(resid + p * 2) = make_float2(r0, r1);
        if (grid) {
            float a0 = 0.f, a1 = 0.f;
            if (theta) {
                const float *t = theta + (size_t)n * 6;
                const float bx = ac ? (W > 1 ? (2.f * xq) / (float)(W - 1) - 1.f : 0.f) : (2.f * xq + 1.f) / (float)W - 1.f;
                const float by = ac ? (H > 1 ? (2.f * y) / (float)(H - 1) - 1.f : 0.f) : (2.f * y + 1.f) / (float)H - 1.f;
                a0 = t[0] * bx + t[1] * by + t[2];
                a1 = t[3] * bx + t[4] * by + t[5];
            }
            *reinterpret_cast<float2 *>(grid + p * 2) = make_float2(r0 + a0, r1 + a1);
        }
    }
}


// bf16 storage, 64 input channels (BASELINE configs[2] / [3]): the same layer on the bf16 matrix cores.  A conv with TWO output
// channels wastes 15/16 of a matrix tile as a 3x3 implicit GEMM -- but  z[p][o] = sum_tap Y[p + tap][tap, o]  with the POINTWISE
// product  Y[q][(tap, o)] = sum_c x[q][c] W[tap][c][o]:  one 64 -> 18 (of 32) GEMM per halo pixel, then a 9-point stencil sum over
// Y.  A operands come straight from global memory (a lane's 8 consecutive channels of its pixel = one 16-byte load: x is read once,
// never staged), the weights sit in registers as three bf16 terms (w = hi + mid + lo to 2^-26: the products are as exact as the fp32
// kernel's; a (hi, lo) pair left 1.4e-5 on the field), Y goes through 23 KB of LDS.  The VALU kernel above needs 1152 FMAs per pixel (341 us at 64 x 256^2 x 64, 3x its HBM
// time); this one is bound by the read of x.
__global__ void __launch_bounds__(256) field_head16_mfma_kernel(const __bf16 *__restrict__ x, int ld, int N, int H, int W,
                                                                const float *__restrict__ w_out, const float *__restrict__ b_out,
                                                                const float *__restrict__ theta, int ac, float *__restrict__ resid,
                                                                float *__restrict__ grid, int tiles_x, int tiles_y, unsigned ntiles, int raw) {
    constexpr int C = 64, NPIX = FH_I * FH_I, YP = 18;   // Y row = 18 floats: the stencil's float2 reads of 16 neighbouring pixels hit 32 different banks
    __shared__ float ys[NPIX * YP];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // B operand: column l31 = (tap, o), this lane's 8 channels of each of the four 16-channel k-steps, split hi + lo
    bf16x8 bh[4], bl[4], bl2[4];
    {
        const int tap = l31 >> 1, o = l31 & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float w = l31 < 18 ? w_out[((size_t)tap * C + ks * 16 + hi * 8 + i) * 2 + o] : 0.f;
                const __bf16 h = (__bf16)w;
                const float r1 = w - (float)h;
                const __bf16 m = (__bf16)r1;
                bh[ks][i] = h, bl[ks][i] = m, bl2[ks][i] = (__bf16)(r1 - (float)m);
            }
    }
    constexpr int MT = (NPIX + 31) / 32;   // 11 tiles of 32 halo pixels: wave w takes the tiles w, w + 4, w + 8
    constexpr int MTW = (MT + 3) / 4;
    // Persistent: a workgroup walks the tiles t = blockIdx.x, + gridDim.x, ... (the grid is a multiple of the XCD count, so all of
    // them map to its XCD's contiguous chunk), keeps the 96 weight terms of its lanes in registers for all of them and has the next
    // tile's 12 x 16 bytes per lane in flight during the stencil and the stores of the current one.
    bf16x8 a[MTW][4];
    auto load_tile = [&](unsigned t) {
        const unsigned tile = xcd_remap(t, ntiles);
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * FH_T, y0 = ty_i * FH_T;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int q = (wv + 4 * i) * 32 + l31;
            const int iy = y0 - 1 + q / FH_I, ix = x0 - 1 + q % FH_I;
            const bool ok = q < NPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const __bf16 *px = x + (ok ? ((size_t)(n * H + iy) * W + ix) * ld + hi * 8 : 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(px + ks * 16);   // unconditional load, masked below
                a[i][ks] = __builtin_bit_cast(bf16x8, ok ? v : (u32x4){0u, 0u, 0u, 0u});
            }
        }
    };
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    load_tile(t);
    const float bias0 = b_out ? b_out[0] : 0.f, bias1 = b_out ? b_out[1] : 0.f;
    for (;;) {
        const unsigned tile = xcd_remap(t, ntiles);
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * FH_T, y0 = ty_i * FH_T;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = wv + 4 * i;
            if (mt >= MT) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], bl2[ks], acc, 0, 0, 0);
            }
            if (l31 < 18) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (qq < NPIX) ys[qq * YP + l31] = acc[r];
                }
            }
        }
        const unsigned tn = t + gridDim.x;
        const bool more = tn < ntiles;
        if (more) load_tile(tn);
        __syncthreads();
        const int tx = tid & 15, ty = tid >> 4;
        float acc0 = bias0, acc1 = bias1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float2 v = *reinterpret_cast<const float2 *>(ys + ((ty + tap / 3) * FH_I + tx + tap % 3) * YP + tap * 2);
            acc0 += v.x, acc1 += v.y;
        }
        const int y = y0 + ty, xq = x0 + tx;
        if (y < H && xq < W && raw) {
            *reinterpret_cast<float2 *>(resid + (((size_t)n * H + y) * W + xq) * 2) = make_float2(acc0, acc1);
        } else if (y < H && xq < W) {
            const float r0 = tanhf(tanhf(acc0)), r1 = tanhf(tanhf(acc1));
            const size_t p = ((size_t)n * H + y) * W + xq;
            if (resid) *reinterpret_cast<float2 *>(resid + p * 2) = make_float2(r0, r1);
            if (grid) {
                float a0 = 0.f, a1 = 0.f;
                if (theta) {
                    const float *th = theta + (size_t)n * 6;
                    const float bx = ac ? (W > 1 ? (2.f * xq) / (float)(W - 1) - 1.f : 0.f) : (2.f * xq + 1.f) / (float)W - 1.f;
                    const float by = ac ? (H > 1 ? (2.f * y) / (float)(H - 1) - 1.f : 0.f) : (2.f * y + 1.f) / (float)H - 1.f;
                    a0 = th[0] * bx + th[1] * by + th[2];
                    a1 = th[3] * bx + th[4] * by + th[5];
                }
                *reinterpret_cast<float2 *>(grid + p * 2) = make_float2(r0 + a0, r1 + a1);
            }
        }
        if (!more) break;
        __syncthreads();   // everybody is done with this tile's Y before the next tile's products are written
        t = tn;
    }
}

// Second cut of the kernel above (round 5).  profiles/r05_pmc_train_bf16_table.log: 189 us per dispatch of 64 samples = 3.3 TB/s on 630 MB, 15 %
// matrix-pipe busy, 35 % of the cycles spent issuing vector instructions -- its main loop is 430 vector + 490 scalar instructions per 256-pixel
// tile and lane, the memory side waits for them.  Same arithmetic (the three-term weights, fp32 sums, tanh(tanh(.)), the stencil's order of
// additions), fewer instructions per pixel:
//   * tile 16 x 32 (halo 18 x 34 = 612 pixels = 20 blocks of 32, five per wave): 1.20 halo pixels per output pixel instead of 1.27, two output
//     pixels per lane in the stencil phase;
//   * the matrix instruction transposed: the weights are the A operand, a block of 32 halo pixels the B operand, so a lane ends up with
//     (tap, o) rows m = 4 hi + {0..3, 8..11, 16..19} of ITS pixel -- three (two) 16-byte LDS writes per block instead of sixteen 4-byte ones;
//   * the pixels of block g + 2 (fp32: g + 1) are requested while block g is multiplied, through three (two) register sets that rotate with the
//     block index (the tile loop is unrolled three (two) times so that every index is a constant): 48 KB in flight per workgroup where the
//     first cut had one tile's loads issued behind its last matrix instruction;
//   * a block's tile offsets are computed once per kernel, an interior tile adds one scalar to them.
namespace {
constexpr int F2_TH = 16, F2_TW = 32, F2_IH = F2_TH + 2, F2_IW = F2_TW + 2, F2_NPIX = F2_IH * F2_IW, F2_BPW = 5, F2_YP = 20;
static_assert(F2_BPW * 4 * 32 >= F2_NPIX, "five blocks of 32 halo pixels per wave");
constexpr unsigned kF2Oob = 0x7ffffff0u;   // a byte offset no sample reaches (the launch rejects samples of 2^30 bytes and more)
struct F2Tile {
    int n, y0, x0;
};
}  // namespace
// F32: fp32 storage on v_mfma_f32_32x32x2_f32 (k-step j of lane half `hi` = channel 32 hi + j: a lane's 32 consecutive channels, exact fp32
// weights), else bf16 storage on v_mfma_f32_32x32x16_bf16 with the three-term weights.  NSET register sets of one block of pixels each.
template <bool F32, int NSET>
__global__ void __launch_bounds__(256, 3) field_head_v2_kernel(const void *__restrict__ x_, int ld, int N, int H, int W,
                                                               const float *__restrict__ w_out, const float *__restrict__ b_out,
                                                               const float *__restrict__ theta, int ac, float *__restrict__ resid,
                                                               float *__restrict__ grid, int tiles_x, int tiles_y, unsigned ntiles, int raw) {
    constexpr int C = 64;
    constexpr int EB = F32 ? 4 : 2;                // bytes per element
    constexpr int NV = F32 ? 8 : 4;                // 16-byte loads per lane and block: its 32 (fp32) / 4 x 8 (bf16) channels
    __shared__ __attribute__((aligned(16))) float ys[F2_NPIX * F2_YP];   // Y[halo pixel][(tap, o) 0 .. 17 (+2)]: 48 KB
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const unsigned char *x = static_cast<const unsigned char *>(x_);
    const unsigned img_bytes = (unsigned)H * W * ld * EB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // A operand: row l31 = (tap, o).  bf16: this lane's 8 channels of each of the four 16-channel k-steps, as three bf16 terms; fp32: its 32 channels
    bf16x8 wh[F32 ? 1 : 4], wm[F32 ? 1 : 4], wl[F32 ? 1 : 4];
    float w32[F32 ? 32 : 1];
    {
        const int tap = l31 >> 1, o = l31 & 1;
        if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 32; ++j) w32[j] = l31 < 18 ? w_out[((size_t)tap * C + hi * 32 + j) * 2 + o] : 0.f;
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float w = l31 < 18 ? w_out[((size_t)tap * C + ks * 16 + hi * 8 + i) * 2 + o] : 0.f;
                    const __bf16 h = (__bf16)w;
                    const float r1 = w - (float)h;
                    const __bf16 m = (__bf16)r1;
                    wh[ks][i] = h, wm[ks][i] = m, wl[ks][i] = (__bf16)(r1 - (float)m);
                }
        }
    }
    // this lane's halo pixel of each of its wave's five blocks: row | column << 8, and its byte offset from the halo's first pixel
    int geo[F2_BPW];
    unsigned rel[F2_BPW];
#pragma unroll
    for (int j = 0; j < F2_BPW; ++j) {
        const int q = (wv + 4 * j) * 32 + l31;
        const int row = q / F2_IW, col = q - row * F2_IW;
        geo[j] = q < F2_NPIX ? (row | col << 8) : -1;
        rel[j] = q < F2_NPIX ? (unsigned)(((row * W + col) * ld + hi * (F32 ? 32 : 8)) * EB) : kF2Oob;
    }
    auto decode = [&](unsigned t) {
        const unsigned tile = xcd_remap(t, ntiles);
        F2Tile r;
        r.x0 = (int)(tile % (unsigned)tiles_x) * F2_TW;
        r.y0 = (int)((tile / (unsigned)tiles_x) % (unsigned)tiles_y) * F2_TH;
        r.n = (int)(tile / (unsigned)(tiles_x * tiles_y));
        return r;
    };
    // Block j of tile T into a register set, through a buffer descriptor of T's sample: a lane outside the image (or past the halo) carries an
    // offset beyond the descriptor's records and is handed zeros by the address unit -- no select on the loaded words, no 64-bit lane addresses.
    auto issue = [&](const F2Tile &T, int j, u32x4 (&dst)[NV]) {
        const int oy = T.y0 - 1, ox = T.x0 - 1;
        const bool interior = oy >= 0 && oy + F2_IH <= H && ox >= 0 && ox + F2_IW <= W;   // scalar
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(x) + (size_t)T.n * img_bytes, 0, (int)img_bytes, 0x00020000);
        const int toff = (oy * W + ox) * ld * EB;   // scalar; negative at the top / left border (only for lanes the test below rejects)
        unsigned voff = rel[j];   // kF2Oob already where the lane is past the halo
        int soff = toff;
        if (!interior) {
            const int iy = oy + (geo[j] & 0xff), ix = ox + (geo[j] >> 8);
            const bool ok = geo[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            voff = ok ? rel[j] + (unsigned)toff : kF2Oob;
            soff = 0;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)   // (bf16: k-step v = channels 16 v + 8 hi ..)
            dst[v] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff + (F32 ? v * 16 : v * 32), soff, 0));
    };
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    F2Tile T = decode(t);
    unsigned tn = t + gridDim.x;
    bool more = tn < ntiles;
    F2Tile TN = more ? decode(tn) : T;
    const float bias0 = b_out ? b_out[0] : 0.f, bias1 = b_out ? b_out[1] : 0.f;
    float *const yq0 = ys + (wv * 32 + l31) * F2_YP + 4 * hi;            // where this lane's products of block 0 go
    const float *const ysr = ys + ((tid >> 5) * F2_IW + (tid & 31)) * F2_YP;   // this lane's first stencil pixel (the second is 8 rows down)
    u32x4 xb[NSET][NV];
    constexpr int D = NSET - 1;   // blocks requested ahead
#pragma unroll
    for (int j = 0; j < D; ++j) issue(T, j, xb[j]);
    auto do_tile = [&](auto R_) {
        constexpr int R = decltype(R_)::value;   // register set of this tile's block 0
#pragma unroll
        for (int j = 0; j < F2_BPW; ++j) {
            if (j + D < F2_BPW) issue(T, j + D, xb[(j + D + R) % NSET]);
            else if (more) issue(TN, j + D - F2_BPW, xb[(j + D + R) % NSET]);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (F32) {
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 xv = __builtin_bit_cast(f32x4, xb[(j + R) % NSET][k4]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w32[4 * k4], xv[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w32[4 * k4 + 1], xv[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w32[4 * k4 + 2], xv[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w32[4 * k4 + 3], xv[3], acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 xv = __builtin_bit_cast(bf16x8, xb[(j + R) % NSET][ks]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks], xv, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ks], xv, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks], xv, acc, 0, 0, 0);
                }
            }
            // this lane's pixel (column l31 of the block), rows m = 4 hi + {0..3, 8..11, 16..19}
            const int q = (wv + 4 * j) * 32 + l31;
            if (q < F2_NPIX) {
                float *yq = yq0 + j * (128 * F2_YP);   // one lane address, the block a constant offset
                *reinterpret_cast<f32x4 *>(yq) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f32x4 *>(yq + 8) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
                if (hi == 0) *reinterpret_cast<f32x4 *>(yq + 16) = (f32x4){acc[8], acc[9], acc[10], acc[11]};
            }
        }
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int idx = tid + h2 * 256;
            const int ty = idx >> 5, tx = idx & 31;
            float acc0 = bias0, acc1 = bias1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float2 v = *reinterpret_cast<const float2 *>(ysr + (h2 * 8 * F2_IW + (tap / 3) * F2_IW + tap % 3) * F2_YP + tap * 2);
                acc0 += v.x, acc1 += v.y;
            }
            const int y = T.y0 + ty, xq = T.x0 + tx;
            if (y < H && xq < W && raw) {
                *reinterpret_cast<float2 *>(resid + (((size_t)T.n * H + y) * W + xq) * 2) = make_float2(acc0, acc1);
            } else if (y < H && xq < W) {
                const float r0 = tanhf(tanhf(acc0)), r1 = tanhf(tanhf(acc1));
                const size_t pp = ((size_t)T.n * H + y) * W + xq;
                if (resid) *reinterpret_cast<float2 *>(resid + pp * 2) = make_float2(r0, r1);
                if (grid) {
                    float a0 = 0.f, a1 = 0.f;
                    if (theta) {
                        const float *th = theta + (size_t)T.n * 6;
                        // (2 x + 1) / W - 1, or 2 x / (W - 1) - 1 with align_corners: the numerators are exact integers either way, one quotient each
                        const float bx = (ac && W <= 1) ? 0.f : (float)(2 * xq + 1 - ac) / (float)(W - ac) - 1.f;
                        const float by = (ac && H <= 1) ? 0.f : (float)(2 * y + 1 - ac) / (float)(H - ac) - 1.f;
                        a0 = th[0] * bx + th[1] * by + th[2];
                        a1 = th[3] * bx + th[4] * by + th[5];
                    }
                    *reinterpret_cast<float2 *>(grid + pp * 2) = make_float2(r0 + a0, r1 + a1);
                }
            }
        }
        __syncthreads();   // everybody is done with this tile's Y before the next tile's products are written
    };
    auto advance = [&]() {
        T = TN, t = tn, tn += gridDim.x;
        more = tn < ntiles;
        if (more) TN = decode(tn);
    };
    // the register set of a tile's block 0 moves by F2_BPW % NSET from tile to tile: NSET tiles unrolled, every index a constant
    for (;;) {
        do_tile(std::integral_constant<int, 0>{});
        if (!more) break;
        advance();
        do_tile(std::integral_constant<int, F2_BPW % NSET>{});
        if (!more) break;
        advance();
        if constexpr (NSET == 3) {
            do_tile(std::integral_constant<int, (2 * F2_BPW) % NSET>{});
            if (!more) break;
            advance();
        }
    }
}

// The same formulation in exact fp32 (BASELINE configs[1]: fp32 storage, 64 channels) on v_mfma_f32_32x32x2_f32: k-step j of lane
// half `hi` is channel 32 hi + j, so a lane reads the 32 consecutive channels of its pixel (8 x 16 bytes) and holds the matching 32
// weights of its column (tap, o).  Products and sums are fp32 as in the VALU kernel (the summation order differs).
__global__ void __launch_bounds__(256) field_head32_mfma_kernel(const float *__restrict__ x, int ld, int N, int H, int W,
                                                                const float *__restrict__ w_out, const float *__restrict__ b_out,
                                                                const float *__restrict__ theta, int ac, float *__restrict__ resid,
                                                                float *__restrict__ grid, int tiles_x, int tiles_y, unsigned ntiles, int raw) {
    constexpr int C = 64, NPIX = FH_I * FH_I, YP = 18;
    __shared__ float ys[NPIX * YP];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    float bw[32];
    {
        const int tap = l31 >> 1, o = l31 & 1;
#pragma unroll
        for (int j = 0; j < 32; ++j) bw[j] = l31 < 18 ? w_out[((size_t)tap * C + hi * 32 + j) * 2 + o] : 0.f;
    }
    constexpr int MT = (NPIX + 31) / 32, MTW = (MT + 3) / 4;
    // persistent, as field_head16_mfma_kernel: weights loaded once per workgroup, the next tile's loads under the stencil and the stores
    f32x4 a[MTW][8];
    auto load_tile = [&](unsigned t) {
        const unsigned tile = xcd_remap(t, ntiles);
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * FH_T, y0 = ty_i * FH_T;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int q = (wv + 4 * i) * 32 + l31;
            const int iy = y0 - 1 + q / FH_I, ix = x0 - 1 + q % FH_I;
            const bool ok = q < NPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float *px = x + (ok ? ((size_t)(n * H + iy) * W + ix) * ld + hi * 32 : 0);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(px + k4 * 4);   // unconditional load, masked below
                a[i][k4] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    load_tile(t);
    const float bias0 = b_out ? b_out[0] : 0.f, bias1 = b_out ? b_out[1] : 0.f;
    for (;;) {
        const unsigned tile = xcd_remap(t, ntiles);
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * FH_T, y0 = ty_i * FH_T;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = wv + 4 * i;
            if (mt >= MT) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j >> 2][j & 3], bw[j], acc, 0, 0, 0);
            if (l31 < 18) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (qq < NPIX) ys[qq * YP + l31] = acc[r];
                }
            }
        }
        const unsigned tn = t + gridDim.x;
        const bool more = tn < ntiles;
        if (more) load_tile(tn);
        __syncthreads();
        const int tx = tid & 15, ty = tid >> 4;
        float acc0 = bias0, acc1 = bias1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float2 v = *reinterpret_cast<const float2 *>(ys + ((ty + tap / 3) * FH_I + tx + tap % 3) * YP + tap * 2);
            acc0 += v.x, acc1 += v.y;
        }
        const int y = y0 + ty, xq = x0 + tx;
        if (y < H && xq < W && raw) {
            *reinterpret_cast<float2 *>(resid + (((size_t)n * H + y) * W + xq) * 2) = make_float2(acc0, acc1);
        } else if (y < H && xq < W) {
            const float r0 = tanhf(tanhf(acc0)), r1 = tanhf(tanhf(acc1));
            const size_t p = ((size_t)n * H + y) * W + xq;
            if (resid) *reinterpret_cast<float2 *>(resid + p * 2) = make_float2(r0, r1);
            if (grid) {
                float a0 = 0.f, a1 = 0.f;
                if (theta) {
                    const float *th = theta + (size_t)n * 6;
                    const float bx = ac ? (W > 1 ? (2.f * xq) / (float)(W - 1) - 1.f : 0.f) : (2.f * xq + 1.f) / (float)W - 1.f;
                    const float by = ac ? (H > 1 ? (2.f * y) / (float)(H - 1) - 1.f : 0.f) : (2.f * y + 1.f) / (float)H - 1.f;
                    a0 = th[0] * bx + th[1] * by + th[2];
                    a1 = th[3] * bx + th[4] * by + th[5];
                }
                *reinterpret_cast<float2 *>(grid + p * 2) = make_float2(r0 + a0, r1 + a1);
            }
        }
        if (!more) break;
        __syncthreads();
        t = tn;
    }
}

// ---- use_BN training-mode pieces of the two heads (netg.cpp sequences them with pws_bn_train_fwd in between)
// z1[n][j] = sum over K slices of the hidden pre-activations + b1[j]   (theta_hidden_kernel's partials)
__global__ void __launch_bounds__(256) theta_sum_kernel(const float *__restrict__ partial, int nslices, int n, int hidden,
                                                        const float *__restrict__ b_flat, float *__restrict__ z1) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * hidden) return;
    const int j = (int)(e % hidden);
    float acc = b_flat ? b_flat[j] : 0.f;
    for (int sl = 0; sl < nslices; ++sl) acc += partial[(size_t)sl * n * hidden + e];
    z1[e] = acc;
}
// z2[n][o] = sum_j h[n][j] W2[j][o] + b2[o]: one wave per (sample, output)
__global__ void __launch_bounds__(256) theta_z2_kernel(const float *__restrict__ h, int n, int hidden, const float *__restrict__ w_lin,
                                                       const float *__restrict__ b_lin, float *__restrict__ z2) {
    const int wave = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= n * 6) return;
    const int s = wave / 6, o = wave % 6;
    float acc = 0.f;
    for (int j = lane; j < hidden; j += 64) acc = fmaf(h[(size_t)s * hidden + j], w_lin[(size_t)j * 6 + o], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) z2[(size_t)s * 6 + o] = acc + (b_lin ? b_lin[o] : 0.f);
}
// r = tanh(tanh(yhat)); resid = r; grid = r + affine_grid(theta)   (yhat: the BatchNorm output of the `out` conv)
__global__ void __launch_bounds__(256) field_bn_finish_kernel(const float *__restrict__ yhat, const float *__restrict__ theta, int H, int W,
                                                              size_t total, int ac, float *__restrict__ resid, float *__restrict__ grid) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const float2 v = *reinterpret_cast<const float2 *>(yhat + p * 2);
    const float r0 = tanhf(tanhf(v.x)), r1 = tanhf(tanhf(v.y));
    if (resid) *reinterpret_cast<float2 *>(resid + p * 2) = make_float2(r0, r1);
    if (grid) {
        const int xq = (int)(p % W), y = (int)((p / W) % H);
        const size_t n = p / ((size_t)W * H);
        float a0 = 0.f, a1 = 0.f;
        if (theta) {
            const float *t = theta + n * 6;
            const float bx = ac ? (W > 1 ? (2.f * xq) / (float)(W - 1) - 1.f : 0.f) : (2.f * xq + 1.f) / (float)W - 1.f;
            const float by = ac ? (H > 1 ? (2.f * y) / (float)(H - 1) - 1.f : 0.f) : (2.f * y + 1.f) / (float)H - 1.f;
            a0 = t[0] * bx + t[1] * by + t[2], a1 = t[3] * bx + t[4] * by + t[5];
        }
        *reinterpret_cast<float2 *>(grid + p * 2) = make_float2(r0 + a0, r1 + a1);
    }
}

}  // namespace pws

using namespace pws;

static int theta_slices(int k1) {
    int kchunk = 64;
    return (k1 + kchunk - 1) / kchunk;
}

// ---- host helpers of the use_BN training path (declared in common.h, called by netg.cpp; not part of the C ABI)
namespace pws {
int theta_z1(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat, float *ws, float *z1, hipStream_t st) {
    const int k1 = 4 * c, kchunk = 64, nslices = theta_slices(k1);
    hipLaunchKernelGGL(theta_hidden_kernel, dim3((hidden + 63) / 64, nslices, (unsigned)((n + TH_NB - 1) / TH_NB)), dim3(64), sizeof(float) * TH_NB * kchunk, st, x, n, k1, hidden,
                       kchunk, w_flat, ws);
    const size_t e = (size_t)n * hidden;
    hipLaunchKernelGGL(theta_sum_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st, ws, nslices, n, hidden, b_flat, z1);
    return check_launch("theta_z1 kernels");
}
int theta_z2(const float *h, int n, int hidden, const float *w_lin, const float *b_lin, float *z2, hipStream_t st) {
    const size_t waves = (size_t)n * 6;
    hipLaunchKernelGGL(theta_z2_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, h, n, hidden, w_lin, b_lin, z2);
    return check_launch("theta_z2_kernel");
}
int field_head_raw(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *b_out, float *z, hipStream_t st) {
    const int tiles_x = (w + FH_T - 1) / FH_T, tiles_y = (h + FH_T - 1) / FH_T;
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * n;
    hipLaunchKernelGGL(field_head_kernel<false>, dim3(ntiles), dim3(256), 0, st, x, ld, n, h, w, c, w_out, b_out, (const float *)nullptr, 0, z,
                       (float *)nullptr, tiles_x, tiles_y, ntiles, 1);
    return check_launch("field_head_kernel<raw>");
}
int field_bn_finish(const float *yhat, const float *theta, int n, int h, int w, int ac, float *resid, float *grid, hipStream_t st) {
    const size_t total = (size_t)n * h * w;
    hipLaunchKernelGGL(field_bn_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, yhat, theta, h, w, total, ac, resid,
                       grid);
    return check_launch("field_bn_finish_kernel");
}
}  // namespace pws

extern "C" size_t pws_theta_head_ws_floats(int n, int c, int hidden) {
    if (n <= 0 || c <= 0 || hidden <= 0) return 0;
    return (size_t)theta_slices(4 * c) * n * hidden;
}

extern "C" int pws_theta_head_fwd(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat,
                                  const float *w_lin, const float *b_lin, float *ws, float *theta, pws_stream_t stream) {
    return pws_theta_head_fwd_save(x, n, c, hidden, w_flat, b_flat, w_lin, b_lin, ws, theta, nullptr, stream);
}

extern "C" int pws_theta_head_fwd_save(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat,
                                       const float *w_lin, const float *b_lin, float *ws, float *theta, float *h_saved,
                                       pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && hidden > 0, "pws_theta_head_fwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && w_flat && w_lin && theta && ws, "pws_theta_head_fwd: NULL pointer (ws needs pws_theta_head_ws_floats())");
    const int k1 = 4 * c, kchunk = 64, nslices = theta_slices(k1);
    PWS_REQUIRE(hidden * sizeof(float) <= 64 * 1024, "pws_theta_head_fwd: hidden = %d exceeds 64 KB of LDS", hidden);
    ProfScope prof(KID_THETA_HEAD, 2.0 * n * ((double)k1 * hidden + 6.0 * hidden), 4.0 * ((double)k1 * hidden + n * k1),
                   as_stream(stream));
    hipLaunchKernelGGL(theta_hidden_kernel, dim3((hidden + 63) / 64, nslices, (unsigned)((n + TH_NB - 1) / TH_NB)), dim3(64), sizeof(float) * TH_NB * kchunk,
                       as_stream(stream), x, n, k1, hidden, kchunk, w_flat, ws);
    hipLaunchKernelGGL(theta_final_kernel, dim3(n), dim3(256), sizeof(float) * hidden, as_stream(stream), ws, nslices, n, hidden,
                       b_flat, w_lin, b_lin, theta, h_saved);
    return check_launch("theta_head kernels");
}

extern "C" int pws_field_head_fwd(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *b_out,
                                  const float *theta, int align_corners, float *resid, float *grid, pws_stream_t stream) {
    return pws_field_head_fwd_s(x, ld, n, h, w, c, w_out, b_out, theta, align_corners, resid, grid, PWS_STORE_FP32, stream);
}

extern "C" int pws_field_head_fwd_s(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *b_out,
                                    const float *theta, int align_corners, float *resid, float *grid, int store,
                                    pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0, "pws_field_head_fwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && w_out && (resid || grid), "pws_field_head_fwd: NULL pointer");
    const int ch = c < FH_CH ? c : FH_CH;
    PWS_REQUIRE(c % 4 == 0 && c % ch == 0 && ld % 4 == 0 && ld >= c && (reinterpret_cast<size_t>(x) & 15) == 0,
                "pws_field_head_fwd: c=%d must be a multiple of 4 (and of 16 when > 16), ld %% 4 == 0, x 16-B aligned", c);
    const int tiles_x = (w + FH_T - 1) / FH_T, tiles_y = (h + FH_T - 1) / FH_T;
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * n;
    ProfScope prof(KID_FIELD_HEAD, 2.0 * n * h * w * 18.0 * c,
                   (double)n * h * w * (4.0 * c + (resid ? 8.0 : 0.0) + (grid ? 8.0 : 0.0)), as_stream(stream));
    // the matrix-core kernels are persistent: 3 workgroups per CU (registers), a multiple of the XCD count (see the kernels);
    // PWS_OPT_EXPERIMENT 92: one tile per workgroup
    static PerDeviceInt ncu_dev;
    int &ncu = ncu_dev.cur();
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    unsigned pgrid = ntiles;
    if (g_experiment != 92 && ntiles > (unsigned)(ncu * 3)) {   // equal shares: k tiles per workgroup, as few workgroups as that takes
        const unsigned k = (ntiles + (unsigned)(ncu * 3) - 1) / (unsigned)(ncu * 3);
        pgrid = ((ntiles + k - 1) / k + kXcds - 1) / kXcds * kXcds;
    }
    const bool v2_16 = store == PWS_STORE_BF16 && ld % 8 == 0, v2_32 = store != PWS_STORE_BF16;
    if (c == 64 && (v2_16 || v2_32) && g_experiment != 90 && g_experiment != 35 && w >= F2_TW && h >= F2_TH && (long long)h * w * ld * 4 < (1ll << 30)) {
        // the second cut (16 x 32 tiles; PWS_OPT_EXPERIMENT 35: the first cut below)
        const int tx2 = (w + F2_TW - 1) / F2_TW, ty2 = (h + F2_TH - 1) / F2_TH;
        const unsigned nt2 = (unsigned)tx2 * ty2 * n;
        unsigned pg2 = nt2;
        if (g_experiment != 92 && nt2 > (unsigned)(ncu * 3)) {
            const unsigned k = (nt2 + (unsigned)(ncu * 3) - 1) / (unsigned)(ncu * 3);
            pg2 = ((nt2 + k - 1) / k + kXcds - 1) / kXcds * kXcds;
        }
        if (v2_16)
            hipLaunchKernelGGL((field_head_v2_kernel<false, 3>), dim3(pg2), dim3(256), 0, as_stream(stream), x, ld, n, h, w, w_out, b_out, theta, align_corners, resid,
                               grid, tx2, ty2, nt2, 0);
        else
            hipLaunchKernelGGL((field_head_v2_kernel<true, 2>), dim3(pg2), dim3(256), 0, as_stream(stream), x, ld, n, h, w, w_out, b_out, theta, align_corners, resid,
                               grid, tx2, ty2, nt2, 0);
    } else if (store == PWS_STORE_BF16 && c == 64 && ld % 8 == 0 && g_experiment != 90)   // the matrix-core kernel (see its comment)
        hipLaunchKernelGGL(field_head16_mfma_kernel, dim3(pgrid), dim3(256), 0, as_stream(stream), reinterpret_cast<const __bf16 *>(x), ld, n, h, w,
                           w_out, b_out, theta, align_corners, resid, grid, tiles_x, tiles_y, ntiles, 0);
    else if (store == PWS_STORE_BF16)
        hipLaunchKernelGGL(field_head_kernel<true>, dim3(ntiles), dim3(256), 0, as_stream(stream), x, ld, n, h, w, c, w_out, b_out,
                           theta, align_corners, resid, grid, tiles_x, tiles_y, ntiles, 0);
    else if (c == 64 && g_experiment != 90)
        hipLaunchKernelGGL(field_head32_mfma_kernel, dim3(pgrid), dim3(256), 0, as_stream(stream), x, ld, n, h, w, w_out, b_out, theta, align_corners,
                           resid, grid, tiles_x, tiles_y, ntiles, 0);
    else
        hipLaunchKernelGGL(field_head_kernel<false>, dim3(ntiles), dim3(256), 0, as_stream(stream), x, ld, n, h, w, c, w_out, b_out,
                           theta, align_corners, resid, grid, tiles_x, tiles_y, ntiles, 0);
    return check_launch("field_head_kernel");
}

// Backward of the two regression heads (see head.hip): small fp32 VALU kernels, HBM/latency-bound.
//
// field head:  grid = r + A(theta),  r = tanh(t1),  t1 = tanh(z),  z = conv3x3(x; W_out) + b_out
//   g_r = g_grid + g_resid ;  g_z = g_r (1 - r^2)(1 - t1^2) with t1 = atanh(r) (|r| <= tanh(1) = 0.7616, well conditioned)
//   dtheta[n] = sum_pixels g_grid (x) [bx, by, 1] ;  db_out = sum g_z ;  dW_out[tap][c][o] = sum x[pix+tap-1][c] g_z[pix][o]
//   dx[pix][c] = sum_{tap,o} g_z[pix+1-tap][o] W_out[tap][c][o]
// theta head:  theta = L(z2), z2 = W2 h + b2, h = L(z1), z1 = W1 v + b1  (L = LeakyReLU 0.2, v = vec(x_s8))
#include "common.h"

namespace pws {

__device__ __forceinline__ float lrelu_grad(float y) { return y > 0.f ? 1.f : 0.2f; }

// ---- K1: g_z, db_out, dtheta.  Workgroup = (slice of one image, image): every lane walks its slice 256 pixels at a time
// and keeps the 8 partial sums in registers, so there is ONE round of 8 atomics per workgroup (one round per 256 pixels put
// 8192 x 8 atomics on a handful of addresses at N=32: 112 us for a 67 MB elementwise pass).
__global__ void __launch_bounds__(256) field_gz_kernel(const float *__restrict__ resid, const float *__restrict__ g_grid,
                                                       const float *__restrict__ g_resid, int H, int W, int slices, int ac,
                                                       float *__restrict__ gz, float *__restrict__ db_out,
                                                       float *__restrict__ dtheta) {
    __shared__ float red[8][4];
    const int n = blockIdx.y;
    const size_t HW = (size_t)H * W;
    const size_t per = (HW + slices - 1) / slices;
    const size_t q_lo = (size_t)blockIdx.x * per, q_hi = q_lo + per < HW ? q_lo + per : HW;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // dtheta[6], db[2]
    for (size_t q = q_lo + threadIdx.x; q < q_hi; q += 256) {
        const size_t p = (size_t)n * HW + q;
        const int xq = (int)(q % W), yq = (int)(q / W);
        const float2 r = *reinterpret_cast<const float2 *>(resid + p * 2);
        float2 gg = make_float2(0.f, 0.f), gr = make_float2(0.f, 0.f);
        if (g_grid) gg = *reinterpret_cast<const float2 *>(g_grid + p * 2);
        if (g_resid) gr = *reinterpret_cast<const float2 *>(g_resid + p * 2);
        const float t0 = atanhf(r.x), t1 = atanhf(r.y);
        const float z0 = (gg.x + gr.x) * (1.f - r.x * r.x) * (1.f - t0 * t0);
        const float z1 = (gg.y + gr.y) * (1.f - r.y * r.y) * (1.f - t1 * t1);
        *reinterpret_cast<float2 *>(gz + p * 2) = make_float2(z0, z1);
        const float bx = ac ? (W > 1 ? (2.f * xq) / (float)(W - 1) - 1.f : 0.f) : (2.f * xq + 1.f) / (float)W - 1.f;
        const float by = ac ? (H > 1 ? (2.f * yq) / (float)(H - 1) - 1.f : 0.f) : (2.f * yq + 1.f) / (float)H - 1.f;
        v[0] += gg.x * bx, v[1] += gg.x * by, v[2] += gg.x, v[3] += gg.y * bx, v[4] += gg.y * by, v[5] += gg.y, v[6] += z0, v[7] += z1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (threadIdx.x < 6) {
            if (dtheta) atomicAdd(dtheta + (size_t)n * 6 + threadIdx.x, s);
        } else if (db_out) {
            atomicAdd(db_out + threadIdx.x - 6, s);
        }
    }
}

// ---- K2: dx.  One lane per pixel: its 9 g_z taps sit in registers, the weights are wave-uniform and come through the
// scalar unit, so the 18 * C FMAs per pixel run at VALU rate; the result leaves in 16-byte pieces (8 bf16 / 4 fp32
// channels).  (One lane per (pixel, 4 channels) with the weights in LDS was LDS-bound with bank conflicts between the
// channel groups: 531 us at N=32, 256x256x64, against a VALU floor of ~65 us.)
template <bool IO16>
__global__ void __launch_bounds__(256) field_dx_kernel(const float *__restrict__ gz, const float *__restrict__ w_out, int N,
                                                       int H, int W, int C, float *__restrict__ dx, int dx_ld, int accumulate) {
    constexpr int VEC = IO16 ? 8 : 4;
    const size_t total = (size_t)N * H * W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= total) return;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
    float2 g[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        // forward: z[q] += x[q + tap - 1] * W[tap]  =>  dx[p] += gz[p + 1 - tap] * W[tap]
        const int yy = y + 1 - tap / 3, xx = x + 1 - tap % 3;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float2 v = *reinterpret_cast<const float2 *>(gz + (((size_t)n * H + (ok ? yy : y)) * W + (ok ? xx : x)) * 2);
        g[tap] = ok ? v : make_float2(0.f, 0.f);
    }
    for (int c = 0; c < C; c += VEC) {
        float a[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) a[k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float *wp = w_out + ((size_t)tap * C + c) * 2;  // wave-uniform -> scalar loads
#pragma unroll
            for (int k = 0; k < VEC; ++k) a[k] = fmaf(g[tap].x, wp[2 * k], fmaf(g[tap].y, wp[2 * k + 1], a[k]));
        }
        if constexpr (!IO16) {
            float4 o = make_float4(a[0], a[1], a[2], a[3]);
            if (accumulate) {
                const float4 old = ld4<false>(dx, pix * dx_ld + c);
                o.x += old.x, o.y += old.y, o.z += old.z, o.w += old.w;
            }
            st4<false>(dx, pix * dx_ld + c, o);
        }
        if constexpr (IO16) {
            __bf16 *d = reinterpret_cast<__bf16 *>(dx) + pix * dx_ld + c;
            if (accumulate) {
                const uint4 u = *reinterpret_cast<const uint4 *>(d);
                a[0] += __builtin_bit_cast(float, u.x << 16), a[1] += __builtin_bit_cast(float, u.x & 0xffff0000u);
                a[2] += __builtin_bit_cast(float, u.y << 16), a[3] += __builtin_bit_cast(float, u.y & 0xffff0000u);
                a[4] += __builtin_bit_cast(float, u.z << 16), a[5] += __builtin_bit_cast(float, u.z & 0xffff0000u);
                a[6] += __builtin_bit_cast(float, u.w << 16), a[7] += __builtin_bit_cast(float, u.w & 0xffff0000u);
            }
            *reinterpret_cast<uint4 *>(d) = make_uint4(cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(a[4], a[5]),
                                                       cvt_pk_bf16(a[6], a[7]));
        }
    }
}

// bf16 storage: the same arithmetic (one lane per pixel, weights through the scalar unit), but the results of a workgroup's 256
// pixels x FDX_CH channels meet in LDS (fp32, rows of FDX_CH + 4 floats) and leave as 16-byte accesses that are
// contiguous over the 8 lanes of a pixel -- with a lane per pixel the read-modify-write of dx was 64 separate 16-byte pieces 128
// bytes apart per wave instruction.  dx_act: this call completes the gradient of x, itself the output of an activation: the
// (accumulated) sum is multiplied by act'(x) (what pws_dst.act_y does in the data-gradient epilogues).
constexpr int FDX_CH = 32;               // channels per pass through the tile (64: 70 KB of LDS = 8 waves per CU, too few to hide the
                                        // read-modify-write's latency; 32: 37 KB = 16 waves: 497 -> 340 us at 64 x 256 x 256; 16: half-line accesses, 500 us)
constexpr int FDX_PITCH = FDX_CH + 4;   // floats per pixel row of the LDS tile
__global__ void __launch_bounds__(256) field_dx16_kernel(const float *__restrict__ gz, const float *__restrict__ w_out, int N, int H,
                                                         int W, int C, __bf16 *__restrict__ dx, int dx_ld, int accumulate,
                                                         const __bf16 *__restrict__ xact, int x_ld, int dx_act) {
    extern __shared__ float s_tile[];   // [256][FDX_PITCH]
    const size_t total = (size_t)N * H * W;
    const size_t pix0 = (size_t)blockIdx.x * 256, pix = pix0 + threadIdx.x;
    const bool live = pix < total;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), n = live ? (int)(pix / ((size_t)W * H)) : 0;
    float2 g[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + 1 - tap / 3, xx = x + 1 - tap % 3;
        const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float2 v = *reinterpret_cast<const float2 *>(gz + (((size_t)n * H + (ok ? yy : 0)) * W + (ok ? xx : 0)) * 2);
        g[tap] = ok ? v : make_float2(0.f, 0.f);
    }
    for (int cb = 0; cb < C; cb += FDX_CH) {   // FDX_CH channels per pass through the tile
        const int cw = C - cb < FDX_CH ? C - cb : FDX_CH;
        for (int c = 0; c < cw; c += 8) {
            float a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float *wp = w_out + ((size_t)tap * C + cb + c) * 2;  // wave-uniform -> scalar loads
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fmaf(g[tap].x, wp[2 * k], fmaf(g[tap].y, wp[2 * k + 1], a[k]));
            }
            float *t = s_tile + threadIdx.x * FDX_PITCH + c;
            *reinterpret_cast<float4 *>(t) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4 *>(t + 4) = make_float4(a[4], a[5], a[6], a[7]);
        }
        __syncthreads();
        constexpr int LPP = FDX_CH / 8;   // lanes per pixel
        const int q = threadIdx.x % LPP, groups = cw / 8;
        const float sl = dx_act == PWS_ACT_LRELU ? 0.2f : 0.f;
#pragma unroll 2
        for (int it = 0; it < LPP; ++it) {
            const int p = it * (256 / LPP) + threadIdx.x / LPP;
            const size_t gp = pix0 + p;
            if (gp < total && q < groups) {
                const float4 lo = *reinterpret_cast<const float4 *>(s_tile + p * FDX_PITCH + q * 8);
                const float4 hi = *reinterpret_cast<const float4 *>(s_tile + p * FDX_PITCH + q * 8 + 4);
                float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                __bf16 *d = dx + gp * dx_ld + cb + q * 8;
                if (accumulate) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(d);
                    a[0] += __builtin_bit_cast(float, u.x << 16), a[1] += __builtin_bit_cast(float, u.x & 0xffff0000u);
                    a[2] += __builtin_bit_cast(float, u.y << 16), a[3] += __builtin_bit_cast(float, u.y & 0xffff0000u);
                    a[4] += __builtin_bit_cast(float, u.z << 16), a[5] += __builtin_bit_cast(float, u.z & 0xffff0000u);
                    a[6] += __builtin_bit_cast(float, u.w << 16), a[7] += __builtin_bit_cast(float, u.w & 0xffff0000u);
                }
                if (dx_act != PWS_ACT_NONE) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(xact + gp * x_ld + cb + q * 8);
                    a[0] *= __builtin_bit_cast(float, u.x << 16) > 0.f ? 1.f : sl, a[1] *= __builtin_bit_cast(float, u.x & 0xffff0000u) > 0.f ? 1.f : sl;
                    a[2] *= __builtin_bit_cast(float, u.y << 16) > 0.f ? 1.f : sl, a[3] *= __builtin_bit_cast(float, u.y & 0xffff0000u) > 0.f ? 1.f : sl;
                    a[4] *= __builtin_bit_cast(float, u.z << 16) > 0.f ? 1.f : sl, a[5] *= __builtin_bit_cast(float, u.z & 0xffff0000u) > 0.f ? 1.f : sl;
                    a[6] *= __builtin_bit_cast(float, u.w << 16) > 0.f ? 1.f : sl, a[7] *= __builtin_bit_cast(float, u.w & 0xffff0000u) > 0.f ? 1.f : sl;
                }
                *reinterpret_cast<uint4 *>(d) = make_uint4(cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(a[4], a[5]),
                                                           cvt_pk_bf16(a[6], a[7]));
            }
        }
        __syncthreads();   // the tile is rewritten by the next 64 channels
    }
}

// ---- K3: dW_out.  Workgroup = 16x16 pixel tile x 32 channels; lane = (channel, pixel-row group).
constexpr int FB_T = 16, FB_I = FB_T + 2, FB_CH = 32, FB_LDP = FB_CH + 1;

template <bool IO16>
__global__ void __launch_bounds__(256) field_dw_kernel(const float *__restrict__ x, int ld, const float *__restrict__ gz, int N,
                                                       int H, int W, int C, float *__restrict__ dw, int tiles_x, int tiles_y,
                                                       int ntiles) {
    __shared__ float s_in[FB_I * FB_I * FB_LDP];   // the final cross-group reduction re-uses it (s_red)
    __shared__ float s_g[FB_T * FB_T * 2];
    static_assert(FB_I * FB_I * FB_LDP >= 8 * 32 * 18, "s_red fits in s_in");
    float *s_red = s_in;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * FB_CH;
    const int c = tid & 31, grp = tid >> 5;  // 8 groups x 2 tile rows each
    float acc[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) acc[i] = 0.f;
    // a workgroup walks a strided subset of the tiles and keeps its partial sums in registers: one round of atomics per
    // workgroup at the end (one per TILE put 2048 atomics on each of the 1152 addresses: 260 us per call, L2-atomic-bound).
    // The next tile is fetched into registers while the current one is consumed out of LDS (with 2-3 resident workgroups per
    // CU the load -> LDS -> barrier -> compute chain otherwise exposes the full global latency per tile).
    constexpr int NITEM = (FB_I * FB_I * (FB_CH / 4) + 255) / 256;
    float4 pre[NITEM];
    float2 pre_g;
    auto fetch = [&](int tile) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * FB_T, y0 = ty_i * FB_T;
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tid + it * 256;
            const int pix = item / (FB_CH / 4), c4 = (item % (FB_CH / 4)) * 4;
            const int iy = y0 - 1 + pix / FB_I, ix = x0 - 1 + pix % FB_I;
            const bool ok = item < FB_I * FB_I * (FB_CH / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W && c0 + c4 < C;
            // unconditional load from a valid address, zeroed by a select: a conditional load would serialise on vmcnt(0)
            const float4 v = ld4<IO16>(x, ok ? ((size_t)(n * H + iy) * W + ix) * ld + c0 + c4 : 0);
            pre[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int ty = tid >> 4, tx = tid & 15, y = y0 + ty, xq = x0 + tx;
        const bool okg = y < H && xq < W;
        const float2 g = *reinterpret_cast<const float2 *>(gz + (okg ? (((size_t)n * H + y) * W + xq) * 2 : 0));
        pre_g = okg ? g : make_float2(0.f, 0.f);
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();  // previous tile consumed
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tid + it * 256;
            if (item < FB_I * FB_I * (FB_CH / 4)) {
                float *d = s_in + (item / (FB_CH / 4)) * FB_LDP + (item % (FB_CH / 4)) * 4;
                d[0] = pre[it].x, d[1] = pre[it].y, d[2] = pre[it].z, d[3] = pre[it].w;
            }
        }
        s_g[tid * 2] = pre_g.x, s_g[tid * 2 + 1] = pre_g.y;
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);  // in flight during the FMAs below
        for (int ty = grp * 2; ty < grp * 2 + 2; ++ty)
            for (int tx = 0; tx < FB_T; ++tx) {
                const float g0 = s_g[(ty * FB_T + tx) * 2], g1 = s_g[(ty * FB_T + tx) * 2 + 1];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float xv = s_in[((ty + tap / 3) * FB_I + tx + tap % 3) * FB_LDP + c];
                    acc[tap * 2] = fmaf(xv, g0, acc[tap * 2]), acc[tap * 2 + 1] = fmaf(xv, g1, acc[tap * 2 + 1]);
                }
            }
    }
    __syncthreads();  // every lane is done reading the last tile out of s_in
#pragma unroll
    for (int i = 0; i < 18; ++i) s_red[(grp * 32 + c) * 18 + i] = acc[i];
    __syncthreads();
    for (int e = tid; e < 32 * 18; e += 256) {
        const int cc = e / 18, i = e % 18;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += s_red[(g * 32 + cc) * 18 + i];
        if (c0 + cc < C) atomicAdd(dw + ((size_t)(i / 2) * C + c0 + cc) * 2 + (i & 1), s);
    }
}

// ---- K2+K3 fused on the bf16 matrix cores (bf16 storage, C == 64): dx AND dW_out from ONE read of the activation.
// Both are products with the same small matrix G'[q][g] = gz[q - (tap - 1)][o], g = 2 tap + o (18 columns, padded to 32): the 3x3
// shift lives in the 2-channel gz (a 2.6 KB halo tile), so the 64-channel x tile needs NO halo and every x element is read once:
//     dx[q][c]       = sum_g G'[q][g] Wm[c][g]                 (K = 32: two k-steps of v_mfma_f32_32x32x16_bf16, weights as A)
//     dW[tap][c][o] += sum_q x[q][c] G'[q][g]                  (K = pixels: x^T and G' through the transposing LDS read)
// and act'(x) of the dx epilogue comes from the x tile already in LDS.  The VALU kernels above read the 537 MB activation twice
// and spend 1152 FMAs per pixel in each of them (343 + 304 us per call at 64 x 256 x 256: 1.57x / 2.2x their algorithmic bytes,
// 60 % LDS bank conflicts); here the matrix work is 2 MFLOP per 256 pixels and the kernel is a stream of x in, dx out.
// Precision: gz and W_out enter the products rounded to bf16 (as dy and the weights do in every other bf16 data / weight gradient);
// fp32 accumulation; dx leaves as bf16.
// Layouts (copied from wgrad_bf16_kernel, which pins them with tools/probes/tr16_read_probe.hip): x tile = two 32-channel planes of
// [256 px][64 B] rows, G' = one plane of [256 px][32 g]; an operand with K = pixels is two ds_read_b64_tr_b16.
constexpr int FQ_T = 16, FQ_PIX = FQ_T * FQ_T, FQ_ROW = 64;
constexpr int FQ_PLANE = FQ_PIX * FQ_ROW + 128;   // + 128 B: the two channel planes of a pixel (staged by lanes c8 and c8 + 4) on different banks
constexpr int FQ_SX = 0, FQ_SG = 2 * FQ_PLANE, FQ_SGZ = FQ_SG + FQ_PIX * FQ_ROW;
constexpr int FQ_HALO = FQ_T + 2;
constexpr int FQ_LDS = FQ_SGZ + FQ_HALO * FQ_HALO * 8;

__device__ __forceinline__ float fq_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float fq_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ bf16x8 fq_tr_pair(const unsigned char *lds, int off0, int off1) {
    typedef short s16x4_ __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4_ *lptr;
    const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off0));
    const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off1));
    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const s16x8_ v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(256, 2) field_bwd16_mfma_kernel(const __bf16 *__restrict__ x, int x_ld, const float *__restrict__ gz,
                                                                   const float *__restrict__ w_out, int N, int H, int W,
                                                                   __bf16 *__restrict__ dx, int dx_ld, int accumulate, int dx_act,
                                                                   float *__restrict__ dw, int tiles_x, int tiles_y, int ntiles) {
    typedef float f32x16_ __attribute__((ext_vector_type(16)));
    constexpr int C = 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FQ_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lg = lane >> 4;

    // ---- the weights as the A operand of the dx product: matrix row i of row tile rt stands for channel
    // rt * 32 + (r >> 3) * 16 + h * 8 + (r & 7), with h = (i >> 2) & 1 and r = 4 (i >> 3) + (i & 3) -- the result register r of lane
    // half h -- so that a lane ends up with two groups of 8 consecutive channels per row tile, and the two lane halves' 16-byte
    // stores of one instruction form whole 32-byte sectors
    bf16x8 wa[2][2];
    {
        const int ih = (l31 >> 2) & 1, ir = 4 * (l31 >> 3) + (l31 & 3);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int ch = rt * 32 + (ir >> 3) * 16 + ih * 8 + (ir & 7);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned wd[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g0 = ks * 16 + hi * 8 + 2 * j;   // g = 2 tap + o: (g0, g0 + 1) = both outputs of tap g0 / 2
                    const float2 v = g0 < 18 ? *reinterpret_cast<const float2 *>(w_out + ((size_t)(g0 >> 1) * C + ch) * 2) : make_float2(0.f, 0.f);
                    wd[j] = cvt_pk_bf16(v.x, v.y);
                }
                const u32x4 u = {wd[0], wd[1], wd[2], wd[3]};
                wa[rt][ks] = __builtin_bit_cast(bf16x8, u);
            }
        }
    }
    // ---- dW accumulators: wave = (channel half wv & 1, pixel half wv >> 1); rows = channels, columns = g
    f32x16_ accw;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    const int wci = wv & 1, wph = wv >> 1;
    // The four 16-byte slots of a 64-byte row are permuted by XOR with bits 2..3 of the row (pixel) index: the row-strided 16-byte
    // accesses (G' rows written / read per pixel, act'(x) reads of the dx epilogue) then spread over all banks, and a transposing read
    // still covers them exactly once (its 4 consecutive rows share the XOR).  Unpermuted: 55 % of the LDS cycles were bank conflicts.
    const int colb = (16 * (lg & 1) + 4 * (li & 3)) * 2;
    int a_lane[2], b_lane[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = 8 * (lg >> 1) + 4 * q + (li >> 2);
        const int cs = colb ^ (((c >> 2) & 3) << 4);
        a_lane[q] = FQ_SX + wci * FQ_PLANE + (wph * 128 + c) * FQ_ROW + cs;
        b_lane[q] = FQ_SG + (wph * 128 + c) * FQ_ROW + cs;
    }

    // ---- staging: lane = (pixel tid >> 3 (+ 32 per item), 8-channel piece tid & 7)
    const int c8 = tid & 7, p0 = tid >> 3;
    const int xl0 = FQ_SX + (c8 >> 2) * FQ_PLANE + p0 * FQ_ROW + (((c8 & 3) ^ ((p0 >> 2) & 3)) << 4);   // (pixel p0 + 32 it: same XOR)
    u32x4 rx[8];
    float2 rg[2];
    unsigned okx = 0;
    int n0 = 0, y0 = 0, x0 = 0;
    auto locate = [&](int tile) {
        x0 = (tile % tiles_x) * FQ_T, y0 = ((tile / tiles_x) % tiles_y) * FQ_T, n0 = tile / (tiles_x * tiles_y);
    };
    auto issue = [&]() {
        okx = 0;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pix = p0 + 32 * it;
            const int yy = y0 + (pix >> 4), xx = x0 + (pix & 15);
            const bool ok = yy < H && xx < W;
            const size_t e = ok ? (((size_t)n0 * H + yy) * W + xx) * x_ld + c8 * 8 : 0;
            rx[it] = *reinterpret_cast<const u32x4 *>(x + e);
            okx |= ok ? (1u << it) : 0u;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int h = tid + 256 * it;
            const int yy = y0 - 1 + h / FQ_HALO, xx = x0 - 1 + h % FQ_HALO;
            const bool ok = h < FQ_HALO * FQ_HALO && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float2 v = *reinterpret_cast<const float2 *>(gz + (ok ? (((size_t)n0 * H + yy) * W + xx) * 2 : 0));
            rg[it] = ok ? v : make_float2(0.f, 0.f);
        }
    };
    const float sl = dx_act == PWS_ACT_LRELU ? 0.2f : 0.f;

    int tile = blockIdx.x;
    if (tile < ntiles) locate(tile), issue();
    for (; tile < ntiles; tile += gridDim.x) {
        const int ty0 = y0, tx0 = x0, tn0 = n0;
        __syncthreads();   // the previous tile is consumed
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const bool ok = (okx >> it) & 1u;
            u32x4 v;
            v.x = ok ? rx[it].x : 0u, v.y = ok ? rx[it].y : 0u, v.z = ok ? rx[it].z : 0u, v.w = ok ? rx[it].w : 0u;
            *reinterpret_cast<u32x4 *>(lds + xl0 + it * 32 * FQ_ROW) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int h = tid + 256 * it;
            if (h < FQ_HALO * FQ_HALO) *reinterpret_cast<float2 *>(lds + FQ_SGZ + h * 8) = rg[it];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) locate(tile + gridDim.x), issue();   // in flight during everything below
        {   // ---- G' row of pixel tid: 9 taps x 2 outputs as bf16, 14 zero columns
            const int py = tid >> 4, px = tid & 15;
            unsigned wd[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float2 v = *reinterpret_cast<const float2 *>(lds + FQ_SGZ + ((py + 2 - tap / 3) * FQ_HALO + (px + 2 - tap % 3)) * 8);
                wd[tap] = cvt_pk_bf16(v.x, v.y);
            }
            u32x4 *row = reinterpret_cast<u32x4 *>(lds + FQ_SG + tid * FQ_ROW);
            const int sw = (tid >> 2) & 3;
            row[0 ^ sw] = u32x4{wd[0], wd[1], wd[2], wd[3]};
            row[1 ^ sw] = u32x4{wd[4], wd[5], wd[6], wd[7]};
            row[2 ^ sw] = u32x4{wd[8], 0u, 0u, 0u};
            row[3 ^ sw] = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        // ---- dx of this wave's 64 pixels: two column tiles of 32 pixels x two row tiles of 32 channels x two k-steps
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int pix = wv * 64 + ct * 32 + l31;
            bf16x8 bq[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bq[ks] = *reinterpret_cast<const bf16x8 *>(lds + FQ_SG + pix * FQ_ROW + (((ks * 2 + hi) ^ ((pix >> 2) & 3)) << 4));
            const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
            const bool live = yy < H && xx < W;
            const size_t gp = live ? ((size_t)tn0 * H + yy) * W + xx : 0;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16_ d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rt][0], bq[0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[rt][1], bq[1], d, 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ch = rt * 32 + h * 16 + hi * 8;   // 8 consecutive channels: registers 8 h .. 8 h + 7
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = d[h * 8 + k];
                    __bf16 *dptr = dx + gp * dx_ld + ch;
                    if (accumulate) {
                        const u32x4 o = *reinterpret_cast<const u32x4 *>(live ? dptr : dx);
                        v[0] += fq_lo(o.x), v[1] += fq_hi(o.x), v[2] += fq_lo(o.y), v[3] += fq_hi(o.y);
                        v[4] += fq_lo(o.z), v[5] += fq_hi(o.z), v[6] += fq_lo(o.w), v[7] += fq_hi(o.w);
                    }
                    if (dx_act != PWS_ACT_NONE) {   // act'(x) from the tile in LDS
                        const u32x4 yv = *reinterpret_cast<const u32x4 *>(lds + FQ_SX + (ch >> 5) * FQ_PLANE + pix * FQ_ROW +
                                                                          ((((ch & 31) >> 3) ^ ((pix >> 2) & 3)) << 4));
                        v[0] *= fq_lo(yv.x) > 0.f ? 1.f : sl, v[1] *= fq_hi(yv.x) > 0.f ? 1.f : sl;
                        v[2] *= fq_lo(yv.y) > 0.f ? 1.f : sl, v[3] *= fq_hi(yv.y) > 0.f ? 1.f : sl;
                        v[4] *= fq_lo(yv.z) > 0.f ? 1.f : sl, v[5] *= fq_hi(yv.z) > 0.f ? 1.f : sl;
                        v[6] *= fq_lo(yv.w) > 0.f ? 1.f : sl, v[7] *= fq_hi(yv.w) > 0.f ? 1.f : sl;
                    }
                    const u32x4 wq = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    if (live) *reinterpret_cast<u32x4 *>(dptr) = wq;
                }
            }
        }
        // ---- dW: this wave's channel half over its 128 pixels (8 k-steps of 16)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 a = fq_tr_pair(lds, a_lane[0] + j * 16 * FQ_ROW, a_lane[1] + j * 16 * FQ_ROW);
            const bf16x8 b = fq_tr_pair(lds, b_lane[0] + j * 16 * FQ_ROW, b_lane[1] + j * 16 * FQ_ROW);
            accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accw, 0, 0, 0);
        }
    }
    // ---- the two pixel halves of a channel half meet in LDS (one atomic per address and workgroup: with ONE workgroup --
    // deterministic mode -- every address gets a single add per launch), then one atomic per (channel, g): lane column = g = 2 tap + o,
    // register r = channel (r & 3) + 8 (r >> 2) + 4 hi of the half
    __syncthreads();   // every wave is done with the tiles in LDS
    float *red = reinterpret_cast<float *>(lds);
    if (wph == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wci * 16 + r) * 64 + lane] = accw[r];
    }
    __syncthreads();
    if (wph == 0 && l31 < 18) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[r] += red[(wci * 16 + r) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            atomicAdd(dw + ((size_t)(l31 >> 1) * C + ch) * 2 + (l31 & 1), accw[r]);
        }
    }
}

// Deterministic mode: db_out[k] += sum over all pixels of gz[.][k], ONE workgroup, fixed order (strided partial sums per lane, then
// a fixed tree) -- the per-sample workgroups of field_gz_kernel would add to db_out in arrival order.
__global__ void __launch_bounds__(256) ordered_sum2_kernel(const float *__restrict__ gz, size_t pixels, float *__restrict__ db_out) {
    __shared__ float red[2][256];
    float s0 = 0.f, s1 = 0.f;
    for (size_t p = threadIdx.x; p < pixels; p += 256) {
        const float2 v = *reinterpret_cast<const float2 *>(gz + p * 2);
        s0 += v.x, s1 += v.y;
    }
    red[0][threadIdx.x] = s0, red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) red[0][threadIdx.x] += red[0][threadIdx.x + step], red[1][threadIdx.x] += red[1][threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x < 2) db_out[threadIdx.x] += red[threadIdx.x][0];
}

// ---- theta head backward
// T1: dz2 = dtheta * L'(theta), dW2 += h (x) dz2, db2 += dz2, dz1 = (W2^T dz2) * L'(h) -> ws, db1 += dz1.
// A lane owns ONE hidden unit j and walks the samples of its workgroup's slice in order, so its seven sums (dW2[j][0..5], db1[j]) stay in registers and reach
// memory once.  (Rounds 1-5: one workgroup per SAMPLE, seven atomics per hidden unit each -- at 64 samples 64-way contention on 3 584 addresses, 100 us for a
// 512 x 6 matrix, three times per training step.)  nslices > 1: the samples are dealt to nslices workgroups per block of 256 hidden units and the sums meet by
// atomics (<= 8 per address); nslices == 1 (deterministic mode, small batches): plain stream-ordered adds in sample order -- every address gets its adds from one
// lane in a fixed order.
__global__ void __launch_bounds__(256) theta_bwd1_kernel(const float *__restrict__ theta, const float *__restrict__ dtheta,
                                                         const float *__restrict__ h, int hidden, const float *__restrict__ w_lin,
                                                         float *__restrict__ dw_lin, float *__restrict__ db_lin,
                                                         float *__restrict__ db_flat, float *__restrict__ dz1, int bn, int n, int nslices) {
    // bn (use_BN training): `dtheta` already IS dz2 (BatchNorm backward ran on it) and the output is dh, the gradient wrt the
    // BatchNorm output of the hidden layer (its own BatchNorm backward follows); the conv biases get no gradient (BN removes it)
    extern __shared__ float dz2s[];   // [samples of this slice][6]
    const int tid = threadIdx.x;
    const int jblocks = (hidden + 255) / 256;
    const int jb = (int)blockIdx.x % jblocks, sl = (int)blockIdx.x / jblocks;
    const int cnt = sl < n ? (n - sl + nslices - 1) / nslices : 0;   // samples sl, sl + nslices, ...
    for (int t = tid; t < cnt * 6; t += 256) {
        const size_t e = (size_t)(sl + (t / 6) * nslices) * 6 + (t % 6);
        dz2s[t] = bn ? dtheta[e] : dtheta[e] * lrelu_grad(theta[e]);
    }
    __syncthreads();
    const bool atomic = nslices > 1;
    const int j = jb * 256 + tid;
    if (j < hidden) {
        float w[6], acc[6], bsum = 0.f;
#pragma unroll
        for (int o = 0; o < 6; ++o) w[o] = w_lin[(size_t)j * 6 + o], acc[o] = 0.f;
        for (int c = 0; c < cnt; ++c) {
            const size_t i = (size_t)(sl + c * nslices);
            const float hj = h[i * hidden + j];
            float dh = 0.f;
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const float z = dz2s[c * 6 + o];
                acc[o] = fmaf(hj, z, acc[o]);
                dh = fmaf(w[o], z, dh);
            }
            const float d1 = bn ? dh : dh * lrelu_grad(hj);
            dz1[i * hidden + j] = d1;
            bsum += d1;
        }
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            if (atomic) atomicAdd(dw_lin + (size_t)j * 6 + o, acc[o]);
            else dw_lin[(size_t)j * 6 + o] += acc[o];
        }
        if (!bn) {
            if (atomic) atomicAdd(db_flat + j, bsum);
            else db_flat[j] += bsum;
        }
    }
    if (!bn && jb == 0 && tid < 6) {
        float sz = 0.f;
        for (int c = 0; c < cnt; ++c) sz += dz2s[c * 6 + tid];
        if (atomic) atomicAdd(db_lin + tid, sz);
        else db_lin[tid] += sz;
    }
}

// T2: dW1[k][j] += sum_n v[n][k] dz1[n][j]   (stream-ordered read-modify-write: no atomics needed)
__global__ void __launch_bounds__(256) theta_bwd2_kernel(const float *__restrict__ v, const float *__restrict__ dz1, int n, int k1,
                                                         int hidden, float *__restrict__ dw_flat) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)k1 * hidden) return;
    const int j = (int)(e % hidden), k = (int)(e / hidden);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = fmaf(v[(size_t)i * k1 + k], dz1[(size_t)i * hidden + j], s);
    dw_flat[e] += s;
}

// T3: dv[n][k] = sum_j W1[k][j] dz1[n][j]: one wave per (n, k), lanes over j
__global__ void __launch_bounds__(256) theta_bwd3_kernel(const float *__restrict__ w_flat, const float *__restrict__ dz1, int n,
                                                         int k1, int hidden, float *__restrict__ dv, int accumulate) {
    const int wave = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= n * k1) return;
    const int i = wave / k1, k = wave % k1;
    float s = 0.f;
    for (int j = lane; j < hidden; j += 64) s = fmaf(w_flat[(size_t)k * hidden + j], dz1[(size_t)i * hidden + j], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        float *d = dv + (size_t)i * k1 + k;
        *d = accumulate ? *d + s : s;
    }
}

}  // namespace pws

using namespace pws;

extern "C" int pws_field_head_bwd(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                                  const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld,
                                  int dx_accumulate, float *dw_out, float *db_out, float *dtheta, float *ws, pws_stream_t stream) {
    return pws_field_head_bwd_s(x, ld, n, h, w, c, w_out, resid, g_grid, g_resid, align_corners, dx, dx_ld, dx_accumulate, dw_out,
                                db_out, dtheta, ws, PWS_STORE_FP32, stream);
}

// The field head's backward in two halves (netg.cpp puts the BatchNorm backward of the `out` layer between them when use_BN):
// gz: gz = d loss / d (input of tanh(tanh(.))), dtheta, [db_out];  dx_dw: data and weight gradient of the 3x3 conv from gz.
namespace pws {
int field_bwd_gz(const float *resid, const float *g_grid, const float *g_resid, int n, int h, int w, int ac, float *gz, float *db_out,
                 float *dtheta, hipStream_t st) {
    if (dtheta) {
        hipError_t e = hipMemsetAsync(dtheta, 0, sizeof(float) * (size_t)n * 6, st);
        if (e != hipSuccess) {
            set_error("pws_field_head_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
    }
    // ~2048 workgroups over the batch, at least 1024 pixels each
    int slices = (2048 + n - 1) / n;
    const int max_slices = (int)(((size_t)h * w + 1023) / 1024);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1 || t_deterministic) slices = 1;   // deterministic: one workgroup (one add to dtheta[n]) per sample
    PWS_REQUIRE(n <= 65535, "pws_field_head_bwd: more than 65535 samples");
    hipLaunchKernelGGL(field_gz_kernel, dim3((unsigned)slices, (unsigned)n), dim3(256), 0, st, resid, g_grid, g_resid, h, w, slices, ac, gz,
                       t_deterministic ? (float *)nullptr : db_out, dtheta);
    if (t_deterministic && db_out) hipLaunchKernelGGL(ordered_sum2_kernel, dim3(1), dim3(256), 0, st, gz, (size_t)n * h * w, db_out);
    return check_launch("field_gz_kernel");
}

int field_bwd_dx_dw(const float *x, int ld, const float *gz, int n, int h, int w, int c, const float *w_out, float *dx, int dx_ld,
                    int dx_accumulate, float *dw_out, int store, hipStream_t st, int dx_act) {
    const bool io16 = store == PWS_STORE_BF16;
    PWS_REQUIRE(dx_act == PWS_ACT_NONE || (io16 && dx && (dx_act == PWS_ACT_LRELU || dx_act == PWS_ACT_RELU) && ld % 8 == 0),
                "pws_field_head_bwd: dx_act needs bf16 storage, dx, PWS_ACT_LRELU / PWS_ACT_RELU and ld %% 8 == 0");
    const size_t total = (size_t)n * h * w;
    if (io16 && dx && dw_out && c == 64 && ld % 8 == 0 && dx_ld % 8 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 &&
        (reinterpret_cast<size_t>(dx) & 15) == 0 && g_experiment != 91) {
        // dx and dW_out from one read of the activation, on the matrix cores (field_bwd16_mfma_kernel)
        const int tiles_x = (w + FQ_T - 1) / FQ_T, tiles_y = (h + FQ_T - 1) / FQ_T;
        const long ntiles = (long)tiles_x * tiles_y * n;
        PWS_REQUIRE(ntiles < (1L << 30), "pws_field_head_bwd: too many tiles");
        static PerDeviceInt ncu_dev;
        int &ncu = ncu_dev.cur();
        if (ncu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        long gx = 3L * ncu;   // three resident workgroups per CU (51 KB of LDS each), each walks ntiles / gx tiles
        if (gx > ntiles) gx = ntiles;
        if (t_deterministic) gx = 1;   // one adding workgroup
        hipLaunchKernelGGL(field_bwd16_mfma_kernel, dim3((unsigned)gx), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(x), ld, gz, w_out, n, h,
                           w, reinterpret_cast<__bf16 *>(dx), dx_ld, dx_accumulate, dx_act, dw_out, tiles_x, tiles_y, (int)ntiles);
        return check_launch("field_bwd16_mfma_kernel");
    }
    if (dx) {
        PWS_REQUIRE(!io16 || (c % 8 == 0 && dx_ld % 8 == 0), "pws_field_head_bwd: bf16 storage needs c and dx_ld to be multiples of 8");
        if (io16) {
            static PerDeviceFlag attr_set_dev;
            bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
            constexpr int lds_bytes = 256 * FDX_PITCH * (int)sizeof(float);
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&field_dx16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   lds_bytes);
                if (e != hipSuccess) {
                    set_error("hipFuncSetAttribute(field_dx16_kernel): %s", hipGetErrorString(e));
                    return PWS_EHIP;
                }
                attr_set = true;
            }
            hipLaunchKernelGGL(field_dx16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), lds_bytes, st, gz, w_out, n, h, w, c,
                               reinterpret_cast<__bf16 *>(dx), dx_ld, dx_accumulate, reinterpret_cast<const __bf16 *>(x), ld, dx_act);
        } else
            hipLaunchKernelGGL(field_dx_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gz, w_out, n, h, w, c, dx,
                               dx_ld, dx_accumulate);
    }
    if (dw_out) {
        const int tiles_x = (w + FB_T - 1) / FB_T, tiles_y = (h + FB_T - 1) / FB_T;
        const int ntiles = tiles_x * tiles_y * n, cblocks = (c + FB_CH - 1) / FB_CH;
        int gx = (1024 + cblocks - 1) / cblocks;  // ~4 workgroups per CU in total; each walks ntiles / gx tiles
        if (gx > ntiles) gx = ntiles;
        if (t_deterministic) gx = 1;   // one adding workgroup per channel block
        if (io16)
            hipLaunchKernelGGL(field_dw_kernel<true>, dim3((unsigned)gx, (unsigned)cblocks), dim3(256), 0, st, x, ld, gz, n, h, w, c, dw_out,
                               tiles_x, tiles_y, ntiles);
        else
            hipLaunchKernelGGL(field_dw_kernel<false>, dim3((unsigned)gx, (unsigned)cblocks), dim3(256), 0, st, x, ld, gz, n, h, w, c, dw_out,
                               tiles_x, tiles_y, ntiles);
    }
    return check_launch("field_head_bwd kernels");
}

// theta head, use_BN: dz2 given -> dW2 += h (x) dz2, dh = W2^T dz2 (no activation derivative: the BatchNorm backward of the hidden
// layer follows);  then, with dz1:  dW1 += v (x) dz1, dv = W1 dz1
// sample slices of theta_bwd1_kernel: one in deterministic mode (a fixed order of additions), else up to 8 of at least 8 samples each
static int theta_bwd1_slices(int n) {
    if (t_deterministic || n < 16) return 1;
    return n / 8 < 8 ? n / 8 : 8;
}
static size_t theta_bwd1_lds(int n, int nslices) { return (size_t)((n + nslices - 1) / nslices) * 6 * sizeof(float); }
int theta_bwd_bn_lin(const float *dz2, const float *h, int n, int hidden, const float *w_lin, float *dw_lin, float *dh, hipStream_t st) {
    const int nslices = theta_bwd1_slices(n);
    hipLaunchKernelGGL(theta_bwd1_kernel, dim3((unsigned)(((hidden + 255) / 256) * nslices)), dim3(256), theta_bwd1_lds(n, nslices), st, (const float *)nullptr, dz2, h,
                       hidden, w_lin, dw_lin, (float *)nullptr, (float *)nullptr, dh, 1, n, nslices);
    return check_launch("theta_bwd1_kernel<bn>");
}
int theta_bwd_flat(const float *x, int n, int c, int hidden, const float *w_flat, const float *dz1, float *dw_flat, float *dx,
                   int dx_accumulate, hipStream_t st) {
    const int k1 = 4 * c;
    const size_t e = (size_t)k1 * hidden;
    hipLaunchKernelGGL(theta_bwd2_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st, x, dz1, n, k1, hidden, dw_flat);
    if (dx) {
        const size_t waves = (size_t)n * k1;
        hipLaunchKernelGGL(theta_bwd3_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, w_flat, dz1, n, k1, hidden, dx,
                           dx_accumulate);
    }
    return check_launch("theta_bwd2/3 kernels");
}
}  // namespace pws

extern "C" int pws_field_head_bwd_s(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                                    const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld,
                                    int dx_accumulate, float *dw_out, float *db_out, float *dtheta, float *ws, int store,
                                    pws_stream_t stream) {
    return pws_field_head_bwd_act(x, ld, n, h, w, c, w_out, resid, g_grid, g_resid, align_corners, dx, dx_ld, dx_accumulate, dw_out, db_out,
                                  dtheta, ws, store, PWS_ACT_NONE, stream);
}

extern "C" int pws_field_head_bwd_act(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                                      const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld,
                                      int dx_accumulate, float *dw_out, float *db_out, float *dtheta, float *ws, int store, int dx_act,
                                      pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "pws_field_head_bwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && w_out && resid && (g_grid || g_resid) && ws, "pws_field_head_bwd: NULL pointer");
    PWS_REQUIRE(9 * c * 2 * sizeof(float) <= 64 * 1024, "pws_field_head_bwd: c=%d too large", c);
    hipStream_t st = as_stream(stream);
    const size_t total = (size_t)n * h * w;
    ProfScope prof(KID_FIELD_HEAD_BWD, 4.0 * total * 18.0 * c, (double)total * (8.0 * c + 32.0), st);
    int rc = field_bwd_gz(resid, g_grid, g_resid, n, h, w, align_corners, ws, db_out, dtheta, st);
    if (rc == PWS_OK) rc = field_bwd_dx_dw(x, ld, ws, n, h, w, c, w_out, dx, dx_ld, dx_accumulate, dw_out, store, st, dx_act);
    return rc;
}

extern "C" int pws_theta_head_bwd(const float *x, int n, int c, int hidden, const float *w_flat, const float *w_lin,
                                  const float *h_saved, const float *theta, const float *dtheta, float *dw_flat, float *db_flat,
                                  float *dw_lin, float *db_lin, float *dx, int dx_accumulate, float *ws, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && c > 0 && hidden > 0, "pws_theta_head_bwd: bad shape");
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(x && w_flat && w_lin && h_saved && theta && dtheta && dw_flat && db_flat && dw_lin && db_lin && ws,
                "pws_theta_head_bwd: NULL pointer");
    hipStream_t st = as_stream(stream);
    const int k1 = 4 * c;
    ProfScope prof(KID_THETA_HEAD_BWD, 4.0 * n * (double)k1 * hidden, 4.0 * 3.0 * (double)k1 * hidden, st);
    const int nslices = theta_bwd1_slices(n);
    PWS_REQUIRE(theta_bwd1_lds(n, nslices) <= 64 * 1024, "pws_theta_head_bwd: n = %d samples exceed the kernel's LDS staging", n);
    hipLaunchKernelGGL(theta_bwd1_kernel, dim3((unsigned)(((hidden + 255) / 256) * nslices)), dim3(256), theta_bwd1_lds(n, nslices), st, theta, dtheta, h_saved, hidden,
                       w_lin, dw_lin, db_lin, db_flat, ws, 0, n, nslices);
    const size_t e = (size_t)k1 * hidden;
    hipLaunchKernelGGL(theta_bwd2_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st, x, ws, n, k1, hidden, dw_flat);
    if (dx) {
        const size_t waves = (size_t)n * k1;
        hipLaunchKernelGGL(theta_bwd3_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, w_flat, ws, n, k1, hidden, dx,
                           dx_accumulate);
    }
    return check_launch("theta_head_bwd kernels");
}

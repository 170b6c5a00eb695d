// Winograd F(2x2, 3x3) convolution for the 3x3 stride-1 layers (Conv2d k3 s1 p1 and ConvTranspose2d k3 s1 p1 = flipped
// correlation) on the fp32 matrix cores of gfx950.  2.25x fewer multiplies than the direct form:
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray), d = 4x4 input patch, Y = 2x2 outputs
// i.e. 16 independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co],  xi = 0..15.
//
// Workgroup = 512 threads (8 waves) = one 16x16-pixel output region (8x8 = 64 Winograd tiles) x 64 output channels.
//   * wave w owns components xi = 2w, 2w+1: 2 comps x (2 tile blocks x 2 channel blocks) of 32x32 accumulators = 128 VGPRs;
//   * input channels are walked in chunks of 8.  Per chunk the raw halo tile [18x18][8] goes to LDS, one thread per
//     (tile, channel) applies B^T d B and writes V[xi][tile][8] into the OTHER of two V buffers while the MFMAs of the
//     current chunk run on the first -- the transform's VALU/LDS work hides under the 64-cycle MFMAs (one barrier pair
//     per chunk);
//   * U (the transformed weights, prepared once at pack time as [xi][cin][cout]) is not shared between waves (each wave
//     has its own components), so the B fragments are read straight from L2 into registers, one chunk ahead;
//   * epilogue: the 16 components of a (tile, channel) live in 8 different waves, so they meet in LDS (16 channels at a
//     time), then A^T M A + bias + activation, NHWC store.
// fp32 throughout; the result differs from the direct kernel by Winograd's usual ~1e-6 relative rounding.
#include <cstdlib>

#include "common.h"

namespace pws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoParams {
    const float *src_ptr[4];
    int src_c[4];
    int src_ld[4];
    int nsrc;
    int N, H, W;
    int cin_pad, cout;
    const float *uw;    // [16][cin_pad][cout]
    const float *bias;
    float *out;
    int out_ld, act;
    int tiles_x, tiles_y;
    unsigned ntiles;
    int ablate;  // measurement knob (PWS_WINO_ABLATE): 1 skip epilogue, 2 skip transform, 4 skip MFMA, 8 skip B loads
};

constexpr int WN_CK = 8, WN_CKP = 9;
constexpr int WN_RAW = 18 * 18 * WN_CKP;            // floats
constexpr int WN_V = 16 * 64 * WN_CKP;              // floats per V buffer
constexpr int WN_LDS_MAIN = WN_RAW + 2 * WN_V;      // 21348 floats = 85.4 KB
constexpr int WN_MP = 17;                           // epilogue row pad: [xi][tile][16 + 1]
constexpr int WN_LDS_EPI = 16 * 64 * WN_MP;         // 17408 floats
constexpr int WN_LDS_BYTES = (WN_LDS_MAIN > WN_LDS_EPI ? WN_LDS_MAIN : WN_LDS_EPI) * 4;

__global__ void __launch_bounds__(512, 2) wino_k3s1_kernel(const WinoParams p) {
    extern __shared__ float lds[];
    float *raw = lds;
    float *vbuf = lds + WN_RAW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;

    const unsigned tile = xcd_remap(blockIdx.x, p.ntiles);
    const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty_i * 16, x0 = tx_i * 16, co0 = blockIdx.y * 64;

    // ---- raw-tile staging descriptors: 324 pixels x 2 float4 = 648 items over 512 threads (2 per thread)
    int g_pix[2], l_off[2], g_c4[2];
    bool g_ok[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 512;
        const int pix = item >> 1, c4 = (item & 1) * 4;
        const int ly = pix / 18, lx = pix % 18;
        const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
        const bool in = item < 648;
        g_ok[it] = in && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = g_ok[it] ? (n * p.H + iy) * p.W + ix : 0;
        l_off[it] = in ? pix * WN_CKP + c4 : -1;
        g_c4[it] = c4;
    }
    float4 r_raw[2];
    auto load_raw = [&](int s, int c0) {
        const float *sp = p.src_ptr[s] + c0;
        const size_t ld = p.src_ld[s];
#pragma unroll
        for (int it = 0; it < 2; ++it) r_raw[it] = *reinterpret_cast<const float4 *>(sp + (size_t)g_pix[it] * ld + g_c4[it]);
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (l_off[it] >= 0) {
                float *d = raw + l_off[it];
                d[0] = g_ok[it] ? r_raw[it].x : 0.f, d[1] = g_ok[it] ? r_raw[it].y : 0.f;
                d[2] = g_ok[it] ? r_raw[it].z : 0.f, d[3] = g_ok[it] ? r_raw[it].w : 0.f;
            }
        }
    };
    // ---- input transform: thread = (tile t, channel c)
    const int t_tile = tid >> 3, t_c = tid & 7;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const int t_src = ((2 * t_ty) * 18 + 2 * t_tx) * WN_CKP + t_c;
    const int t_dst = t_tile * WN_CKP + t_c;
    auto transform = [&](float *vdst) {
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = raw[t_src + (i * 18 + j) * WN_CKP];
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vdst[t_dst + (i * 4 + 0) * 64 * WN_CKP] = t[i][0] - t[i][2];
            vdst[t_dst + (i * 4 + 1) * 64 * WN_CKP] = t[i][1] + t[i][2];
            vdst[t_dst + (i * 4 + 2) * 64 * WN_CKP] = t[i][2] - t[i][1];
            vdst[t_dst + (i * 4 + 3) * 64 * WN_CKP] = t[i][1] - t[i][3];
        }
    };
    // ---- B fragments (transformed weights) straight from L2: [comp 2][kk 4][nsub 2]
    float bcur[2][4][2], bnxt[2][4][2];
    const bool co_ok0 = co0 + l31 < p.cout, co_ok1 = co0 + 32 + l31 < p.cout;
    auto load_b = [&](float (&b)[2][4][2], int wrow) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float *row = p.uw + ((size_t)(2 * wv + c) * p.cin_pad + wrow + 2 * kk + hi) * p.cout + co0;
                const float v0 = row[co_ok0 ? l31 : 0], v1 = row[co_ok1 ? 32 + l31 : 0];
                b[c][kk][0] = co_ok0 ? v0 : 0.f, b[c][kk][1] = co_ok1 ? v1 : 0.f;
            }
    };

    f32x16 acc[2][2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][m][nn][r] = 0.f;
    const int a_base = (2 * wv * 64 + l31) * WN_CKP + hi;  // + comp*64*CKP + m*32*CKP + 2*kk

    int total_chunks = 0;
    for (int s = 0; s < p.nsrc; ++s) total_chunks += p.src_c[s] / WN_CK;
    // ---- prologue: chunk 0 raw -> LDS -> V[0]; chunk 1 raw and chunk 0 B fragments in flight
    int s = 0, c0 = 0;
    load_raw(s, c0);
    load_b(bcur, 0);
    store_raw();
    __syncthreads();
    transform(vbuf);
    c0 += WN_CK;
    if (c0 >= p.src_c[s] && s < p.nsrc - 1) ++s, c0 = 0;
    if (total_chunks > 1) load_raw(s, c0);
    __syncthreads();
    for (int ch = 0; ch < total_chunks; ++ch) {
        const bool more = ch + 1 < total_chunks;
        if (more) store_raw();  // chunk ch+1 (raw of chunk ch was consumed by its transform before the last barrier)
        __syncthreads();
        // prefetch: raw of chunk ch+2, B fragments of chunk ch+1
        c0 += WN_CK;
        if (c0 >= p.src_c[s] && s < p.nsrc - 1) ++s, c0 = 0;
        if (ch + 2 < total_chunks) load_raw(s, c0);
        if (more && !(p.ablate & 8)) load_b(bnxt, (ch + 1) * WN_CK);
        // transform of chunk ch+1 into the other V buffer, interleaved by the scheduler with the MFMAs of chunk ch
        if (more && !(p.ablate & 2)) transform(vbuf + ((ch + 1) & 1) * WN_V);
        const float *v = vbuf + (ch & 1) * WN_V;
        if (!(p.ablate & 4))
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float a0 = v[a_base + c * 64 * WN_CKP + 2 * kk];
                const float a1 = v[a_base + c * 64 * WN_CKP + 32 * WN_CKP + 2 * kk];
                acc[c][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bcur[c][kk][0], acc[c][0][0], 0, 0, 0);
                acc[c][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bcur[c][kk][1], acc[c][0][1], 0, 0, 0);
                acc[c][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bcur[c][kk][0], acc[c][1][0], 0, 0, 0);
                acc[c][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bcur[c][kk][1], acc[c][1][1], 0, 0, 0);
            }
        if (more) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bcur[c][kk][0] = bnxt[c][kk][0], bcur[c][kk][1] = bnxt[c][kk][1];
        }
        __syncthreads();
    }

    // ---- epilogue: 4 rounds of 16 output channels through LDS [xi][tile][16+1]
    if (p.ablate & 1) {
        if (acc[0][0][0][0] == 12345.678f) p.out[0] = acc[1][1][1][3];  // keep the accumulators live
        return;
    }
    float *mb = lds;
    const int e_tile = tid >> 3;                 // 512 threads = 64 tiles x 8 channel pairs
    const int e_co = (tid & 7) * 2;
    const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if ((l31 >> 4) == (q & 1)) {
            const int col = l31 & 15;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int trow = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        mb[((2 * wv + c) * 64 + trow) * WN_MP + col] = acc[c][m][q >> 1][r];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = e_co + u;
            const int co = co0 + q * 16 + col;
            float m_[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m_[i][j] = mb[((i * 4 + j) * 64 + e_tile) * WN_MP + col];
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t0[j] = m_[0][j] + m_[1][j] + m_[2][j], t1[j] = m_[1][j] - m_[2][j] - m_[3][j];
            const float bias = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
            const float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
            const float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
            const int oy = y0 + 2 * e_ty, ox = x0 + 2 * e_tx;
            if (co < p.cout) {
                float *o = p.out + ((size_t)(n * p.H + oy) * p.W + ox) * p.out_ld + co;
                const size_t rs = (size_t)p.W * p.out_ld;
                if (oy < p.H && ox < p.W) o[0] = act_apply(y00 + bias, p.act);
                if (oy < p.H && ox + 1 < p.W) o[p.out_ld] = act_apply(y01 + bias, p.act);
                if (oy + 1 < p.H && ox < p.W) o[rs] = act_apply(y10 + bias, p.act);
                if (oy + 1 < p.H && ox + 1 < p.W) o[rs + p.out_ld] = act_apply(y11 + bias, p.act);
            }
        }
        __syncthreads();
    }
}

// U = G g G^T from the packed correlation kernel P[tap][cin_pad][cout] (so conv and flipped convT are both covered)
__global__ void wino_pack_kernel(const float *__restrict__ pk, float *__restrict__ uw, size_t plane /* cin_pad*cout */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
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
        uw[(size_t)(r * 4 + 0) * plane + i] = u[r][0];
        uw[(size_t)(r * 4 + 1) * plane + i] = 0.5f * (u[r][0] + u[r][1] + u[r][2]);
        uw[(size_t)(r * 4 + 2) * plane + i] = 0.5f * (u[r][0] - u[r][1] + u[r][2]);
        uw[(size_t)(r * 4 + 3) * plane + i] = u[r][2];
    }
}

// Called by conv2d_fwd_impl for eligible launches; returns PWS_OK after launching, or a negative error.
int wino_k3s1_launch(const pws_conv_args *a, const ProfHint &ph, hipStream_t st) {
    WinoParams p{};
    p.nsrc = a->nsrc;
    int cin = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        p.src_ptr[s] = a->src[s].ptr, p.src_c[s] = a->src[s].channels, p.src_ld[s] = a->src[s].ld;
        cin += a->src[s].channels;
    }
    p.N = a->n, p.H = a->h, p.W = a->w, p.cin_pad = (cin + 15) / 16 * 16, p.cout = a->cout;
    p.uw = a->w_wino, p.bias = a->bias, p.out = a->out, p.out_ld = a->out_ld, p.act = a->act;
    p.tiles_x = (a->w + 15) / 16, p.tiles_y = (a->h + 15) / 16;
    p.ntiles = (unsigned)(p.tiles_x * p.tiles_y * a->n);
    static const int ablate = getenv("PWS_WINO_ABLATE") ? atoi(getenv("PWS_WINO_ABLATE")) : 0;
    p.ablate = ablate;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_k3s1_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wino_k3s1_kernel): %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    ProfScope prof(KID_CONV_WINO, ph.flops, ph.bytes, st);
    hipLaunchKernelGGL(wino_k3s1_kernel, dim3(p.ntiles, (a->cout + 63) / 64), dim3(512), WN_LDS_BYTES, st, p);
    return check_launch("wino_k3s1_kernel");
}

}  // namespace pws

extern "C" size_t pws_packed_wino_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return (size_t)16 * ((cin + 15) / 16 * 16) * cout;
}

extern "C" int pws_pack_conv_weight_wino(const float *w_packed, float *w_wino, int cin, int cout, pws_stream_t stream) {
    PWS_REQUIRE(w_packed && w_wino && cin > 0 && cout > 0, "pws_pack_conv_weight_wino: bad arguments");
    const size_t plane = (size_t)((cin + 15) / 16 * 16) * cout;
    hipLaunchKernelGGL(pws::wino_pack_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, pws::as_stream(stream), w_packed,
                       w_wino, plane);
    return pws::check_launch("wino_pack_kernel");
}

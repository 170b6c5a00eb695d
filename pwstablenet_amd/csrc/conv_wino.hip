// Winograd F(2x2, 3x3) convolution for the 3x3 stride-1 layers (Conv2d k3 s1 p1 and ConvTranspose2d k3 s1 p1 = flipped
// correlation) on the fp32 matrix cores of gfx950.  2.25x fewer multiplies than the direct form:
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray), d = 4x4 input patch, Y = 2x2 outputs
// i.e. 16 independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co],  xi = 0..15.
//
// Workgroup = 256 threads (4 waves) = one 8x16-pixel output region (4x8 = 32 Winograd tiles) x 64 output channels;
// 76 KB of LDS -> two workgroups per CU, which run out of phase and hide each other's staging (a single 8-wave
// workgroup per CU measured 35-50 % MFMA utilisation: staging, barriers and the epilogue were fully exposed).
//   * wave w owns the Winograd ROW i = w, i.e. components xi = 4w .. 4w+3: 4 comps x 2 channel blocks of 32x32
//     accumulators = 128 VGPRs;
//   * input channels are walked in chunks of 8.  Per chunk the raw halo tile [10x18][8] and the transformed-weight chunk
//     U[16][8][64] (prepared once at pack time) go to LDS; one thread per (tile, channel) applies B^T d B and writes
//     V[xi][tile][8] into the OTHER of two V buffers while the MFMAs of the current chunk run on the first, so the
//     transform's VALU/LDS work hides under the 64-cycle MFMAs; raw and U of the next chunk are prefetched into registers;
//   * epilogue: a wave holds all four column components of its row, so the column half of A^T M A is done in registers;
//     the four rows meet in LDS once (2 partial values per tile and channel), then bias + activation and NHWC stores with
//     lane = channel (256-B coalesced per pixel).
// fp32 throughout; the result differs from the direct kernel by Winograd's usual ~1e-6 relative rounding.
#include <cstdlib>
#include <type_traits>

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
    int ablate;  // measurement knob (env PWS_WINO_ABLATE, tools/conv_bench.py): 1 skip epilogue, 2 skip transform,
                 // 4 skip MFMA, 8 skip U loads -- results are then wrong on purpose; 0 in normal operation
};

constexpr int WN_CK = 8, WN_CKP = 9;
constexpr int WN_TH = 8, WN_TW = 16;                        // output pixels per workgroup: 4 x 8 = 32 Winograd tiles
constexpr int WN_RH = WN_TH + 2, WN_RW = WN_TW + 2;         // raw halo tile
constexpr int WN_RAW = (WN_RH * WN_RW * WN_CKP + 3) / 4 * 4;  // floats (16-B multiple: U below is float4-accessed)
constexpr int WN_V = 16 * 32 * WN_CKP;                      // floats per V buffer  [xi][tile][8+1]
constexpr int WN_U = 16 * WN_CK * 64;                       // floats: U chunk     [xi][k][64 cout]
constexpr int WN_LDS_MAIN = WN_RAW + 2 * WN_V + WN_U;       // 19028 floats = 76.1 KB  -> 2 workgroups per CU
constexpr int WN_EP = 65;                                   // epilogue row pad: [i 4][b 2][tile 32][64 + 1]
constexpr int WN_LDS_EPI = 4 * 2 * 32 * WN_EP;              // 16640 floats
constexpr int WN_LDS_BYTES = (WN_LDS_MAIN > WN_LDS_EPI ? WN_LDS_MAIN : WN_LDS_EPI) * 4;
constexpr int WN_RAW_ITEMS = WN_RH * WN_RW * 2;             // float4 items of the raw tile (360)

__global__ void __launch_bounds__(256, 2) wino_k3s1_kernel(const WinoParams p) {
    extern __shared__ float lds[];
    float *raw = lds;
    float *vbuf = lds + WN_RAW;
    float *ubuf = lds + WN_RAW + 2 * WN_V;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;

    const unsigned tile = xcd_remap(blockIdx.x, p.ntiles);
    const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty_i * WN_TH, x0 = tx_i * WN_TW, co0 = blockIdx.y * 64;

    // ---- raw-tile staging descriptors: 180 pixels x 2 float4 over 256 threads (2 per thread, second one partial)
    int g_pix[2], l_off[2], g_c4[2];
    bool g_ok[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 256;
        const int pix = item >> 1, c4 = (item & 1) * 4;
        const int ly = pix / WN_RW, lx = pix % WN_RW;
        const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
        const bool in = item < WN_RAW_ITEMS;
        g_ok[it] = in && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = g_ok[it] ? (n * p.H + iy) * p.W + ix : 0;
        l_off[it] = in ? pix * WN_CKP + c4 : -1;
        g_c4[it] = c4;
    }
    // Global loads run TWO chunks ahead of their use (a chunk is only ~1-2 us of matrix work, about one HBM latency):
    // two register sets each for the raw tile and for U, indexed statically (the chunk loop is unrolled by 2).
    float4 r_raw[2][2];
    auto load_raw = [&](float4 (&r)[2], int s, int c0) {
        const float *sp = p.src_ptr[s] + c0;
        const size_t ld = p.src_ld[s];
#pragma unroll
        for (int it = 0; it < 2; ++it) r[it] = *reinterpret_cast<const float4 *>(sp + (size_t)g_pix[it] * ld + g_c4[it]);
    };
    auto store_raw = [&](const float4 (&r)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (l_off[it] >= 0) {
                float *d = raw + l_off[it];
                d[0] = g_ok[it] ? r[it].x : 0.f, d[1] = g_ok[it] ? r[it].y : 0.f;
                d[2] = g_ok[it] ? r[it].z : 0.f, d[3] = g_ok[it] ? r[it].w : 0.f;
            }
        }
    };
    // ---- U chunk [16][8][64]: 2048 float4 over 256 threads = 8 per thread, all unconditional loads
    float4 r_u[2][8];
    const int u_q = (tid & 15) * 4, u_k = (tid >> 4) & 7;      // item = tid + it*256 -> xi = it*2 + (tid >> 7)
    const bool u_ok = co0 + u_q < p.cout;
    const float *u_base = p.uw + ((size_t)(tid >> 7) * p.cin_pad + u_k) * p.cout + (u_ok ? co0 + u_q : 0);
    auto load_u = [&](float4 (&r)[8], int wrow) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            r[it] = *reinterpret_cast<const float4 *>(u_base + ((size_t)(2 * it) * p.cin_pad + wrow) * p.cout);
    };
    auto store_u = [&](const float4 (&r)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            *reinterpret_cast<float4 *>(ubuf + (tid + it * 256) * 4) = u_ok ? r[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // ---- input transform: thread = (tile t, channel c)
    const int t_tile = tid >> 3, t_c = tid & 7;
    const int t_src = ((2 * (t_tile >> 3)) * WN_RW + 2 * (t_tile & 7)) * WN_CKP + t_c;
    const int t_dst = t_tile * WN_CKP + t_c;
    auto transform = [&](float *vdst) {
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = raw[t_src + (i * WN_RW + j) * WN_CKP];
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vdst[t_dst + (i * 4 + 0) * 32 * WN_CKP] = t[i][0] - t[i][2];
            vdst[t_dst + (i * 4 + 1) * 32 * WN_CKP] = t[i][1] + t[i][2];
            vdst[t_dst + (i * 4 + 2) * 32 * WN_CKP] = t[i][2] - t[i][1];
            vdst[t_dst + (i * 4 + 3) * 32 * WN_CKP] = t[i][1] - t[i][3];
        }
    };

    // wave w owns the Winograd row i = w: components xi = 4w + j, j = 0..3, all 32 tiles x 64 channels
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nn][r] = 0.f;
    const int a_base = (4 * wv * 32 + l31) * WN_CKP + hi;   // + j*32*CKP + 2*kk
    const int b_base = (4 * wv * WN_CK + hi) * 64 + l31;     // + (j*CK + 2*kk)*64 + nt*32

    int total_chunks = 0;
    for (int s = 0; s < p.nsrc; ++s) total_chunks += p.src_c[s] / WN_CK;
    // source cursor of the NEXT raw chunk to request
    int s = 0, c0 = 0;
    auto advance = [&]() {
        c0 += WN_CK;
        if (c0 >= p.src_c[s] && s < p.nsrc - 1) ++s, c0 = 0;
    };
    // ---- prologue: raw(0) -> LDS -> V[0]; raw(1), raw(2), U(0), U(1) in flight
    load_raw(r_raw[0], s, c0), advance();
    load_u(r_u[0], 0);
    store_raw(r_raw[0]);
    __syncthreads();
    transform(vbuf);
    if (total_chunks > 1) load_raw(r_raw[1], s, c0), advance();   // raw(1) -> set 1
    if (total_chunks > 2) load_raw(r_raw[0], s, c0), advance();   // raw(2) -> set 0
    if (total_chunks > 1) load_u(r_u[1], WN_CK);
    __syncthreads();

    // one chunk; PAR = ch & 1 is a compile-time constant so that the register sets are statically indexed
    auto chunk = [&](int ch, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const bool more = ch + 1 < total_chunks;
        if (more) store_raw(r_raw[PAR ^ 1]);  // raw(ch+1) lives in set (ch+1)&1; raw(ch) was consumed before the last barrier
        store_u(r_u[PAR]);                    // U(ch); the MFMAs of chunk ch-1 finished before the last barrier
        __syncthreads();
        if (ch + 3 < total_chunks) load_raw(r_raw[PAR ^ 1], s, c0), advance();  // raw(ch+3) into the set just stored
        if (ch + 2 < total_chunks && !(p.ablate & 8)) load_u(r_u[PAR], (ch + 2) * WN_CK);
        // transform of chunk ch+1 into the other V buffer; the scheduler interleaves it with the MFMAs of chunk ch
        if (more && !(p.ablate & 2)) transform(vbuf + (PAR ^ 1) * WN_V);
        const float *v = vbuf + PAR * WN_V;
        if (!(p.ablate & 4)) {
            // fragments of k-step kk+1 are requested from LDS before the 8 MFMAs of k-step kk are issued
            float fa[2][4], fb[2][4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fa[0][j] = v[a_base + j * 32 * WN_CKP];
                fb[0][j][0] = ubuf[b_base + (j * WN_CK) * 64], fb[0][j][1] = ubuf[b_base + (j * WN_CK) * 64 + 32];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        fa[nxt][j] = v[a_base + j * 32 * WN_CKP + 2 * (kk + 1)];
                        fb[nxt][j][0] = ubuf[b_base + (j * WN_CK + 2 * (kk + 1)) * 64];
                        fb[nxt][j][1] = ubuf[b_base + (j * WN_CK + 2 * (kk + 1)) * 64 + 32];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the requests above the MFMAs (hipcc otherwise sinks each read to its use)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j][0], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j][1], acc[j][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    for (int ch = 0; ch < total_chunks; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < total_chunks) chunk(ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue.  Each wave holds a whole Winograd row (4 column components), so the column half of A^T M A runs in
    // registers; only 2 partial values per (tile, channel) and row go through LDS, once.
    if (p.ablate & 1) {
        if (acc[0][0][0] == 12345.678f) p.out[0] = acc[3][1][3];  // keep the accumulators live
        return;
    }
    float *eb = lds;  // [i = wave][b][tile][64 + 1]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float t0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
            const float t1 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
            eb[((wv * 2 + 0) * 32 + trow) * WN_EP + nt * 32 + l31] = t0;
            eb[((wv * 2 + 1) * 32 + trow) * WN_EP + nt * 32 + l31] = t1;
        }
    __syncthreads();
    {
        const int col = tid & 63;           // lane = channel: 256-B coalesced stores per output pixel
        const int co = co0 + col;
        const float bias = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = (tid >> 6) + 4 * k;
            float e[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i][0] = eb[((i * 2 + 0) * 32 + t) * WN_EP + col], e[i][1] = eb[((i * 2 + 1) * 32 + t) * WN_EP + col];
            const float y00 = e[0][0] + e[1][0] + e[2][0], y01 = e[0][1] + e[1][1] + e[2][1];
            const float y10 = e[1][0] - e[2][0] - e[3][0], y11 = e[1][1] - e[2][1] - e[3][1];
            const int oy = y0 + 2 * (t >> 3), ox = x0 + 2 * (t & 7);
            if (co < p.cout) {
                float *o = p.out + ((size_t)(n * p.H + oy) * p.W + ox) * p.out_ld + co;
                const size_t rs = (size_t)p.W * p.out_ld;
                if (oy < p.H && ox < p.W) o[0] = act_apply(y00 + bias, p.act);
                if (oy < p.H && ox + 1 < p.W) o[p.out_ld] = act_apply(y01 + bias, p.act);
                if (oy + 1 < p.H && ox < p.W) o[rs] = act_apply(y10 + bias, p.act);
                if (oy + 1 < p.H && ox + 1 < p.W) o[rs + p.out_ld] = act_apply(y11 + bias, p.act);
            }
        }
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
    p.tiles_x = (a->w + WN_TW - 1) / WN_TW, p.tiles_y = (a->h + WN_TH - 1) / WN_TH;
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
    hipLaunchKernelGGL(wino_k3s1_kernel, dim3(p.ntiles, (a->cout + 63) / 64), dim3(256), WN_LDS_BYTES, st, p);
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

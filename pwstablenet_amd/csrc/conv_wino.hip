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
//
// MODE 1 -- the same machinery for ConvTranspose2d k4 s2 p1: each output parity class (py, px) is a 2x2 correlation
// (pack.hip), i.e. F(3x3, 2x2) on the SAME 4x4 input patches and the same B^T (the interpolation points 0, 1, -1, inf are
// unchanged), 16 multiplies for 9 outputs x 4 taps = 2.25x fewer as well:
//     G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]]      A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
// Tiles advance by 3 pixels (raw halo tile 13 x 25 for the 4 x 8 tiles of a workgroup = 12 x 24 class outputs = 24 x 48
// output pixels of that parity), blockIdx.z = class, epilogue = 3 x 3 outputs per tile in two 32-channel passes.
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
constexpr int WN_V = 16 * 32 * WN_CKP;                      // floats per V buffer  [xi][tile][8+1]
constexpr int WN_U = 16 * WN_CK * 64;                       // floats: U chunk     [xi][k][64 cout]

template <int MODE>
struct WinoGeo {
    static constexpr int OT = MODE == 0 ? 2 : 3;                 // outputs per tile and dimension
    static constexpr int TH = 4 * OT, TW = 8 * OT;               // (class) output pixels per workgroup: 4 x 8 tiles
    static constexpr int RH = TH + 4 - OT, RW = TW + 4 - OT;     // raw halo tile
    static constexpr int RAW = (RH * RW * WN_CKP + 3) / 4 * 4;   // floats (16-B multiple: U below is float4-accessed)
    static constexpr int LDS_MAIN = RAW + 2 * WN_V + WN_U;       // MODE 0: 19028 floats = 76.1 KB, MODE 1: 79.4 KB -> 2 per CU
    static constexpr int EPW = MODE == 0 ? 65 : 33;              // epilogue row pad
    static constexpr int LDS_EPI = MODE == 0 ? 4 * 2 * 32 * 65 : 4 * 3 * 32 * 33;
    static constexpr int LDS_BYTES = (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) * 4;
    static constexpr int RAW_ITEMS = RH * RW * 2;                // float4 items of the raw tile
    static constexpr int RITS = (RAW_ITEMS + 255) / 256;         // per thread
};

template <int MODE>
__global__ void __launch_bounds__(256, 2) wino_kernel(const WinoParams p) {
    using G = WinoGeo<MODE>;
    extern __shared__ float lds[];
    float *raw = lds;
    float *vbuf = lds + G::RAW;
    float *ubuf = lds + G::RAW + 2 * WN_V;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;

    const unsigned tile = xcd_remap(blockIdx.x, p.ntiles);
    const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty_i * G::TH, x0 = tx_i * G::TW, co0 = blockIdx.y * 64;
    const int cls = MODE == 1 ? (int)blockIdx.z : 0, py = cls >> 1, px = cls & 1;  // MODE 1: output parity class

    // ---- raw-tile staging descriptors: RH x RW pixels x 2 float4 over 256 threads (last item partial)
    constexpr int RITS = G::RITS;
    int g_pix[RITS], l_off[RITS], g_c4[RITS];
    bool g_ok[RITS];
#pragma unroll
    for (int it = 0; it < RITS; ++it) {
        const int item = tid + it * 256;
        const int pix = item >> 1, c4 = (item & 1) * 4;
        const int ly = pix / G::RW, lx = pix % G::RW;
        const int iy = y0 - 1 + ly + (MODE == 1 ? py : 0), ix = x0 - 1 + lx + (MODE == 1 ? px : 0);
        const bool in = item < G::RAW_ITEMS;
        g_ok[it] = in && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = g_ok[it] ? (n * p.H + iy) * p.W + ix : 0;
        l_off[it] = in ? pix * WN_CKP + c4 : -1;
        g_c4[it] = c4;
    }
    // Global loads run TWO chunks ahead of their use (a chunk is only ~1-2 us of matrix work, about one HBM latency):
    // two register sets each for the raw tile and for U, indexed statically (the chunk loop is unrolled by 2).
    float4 r_raw[2][RITS];
    auto load_raw = [&](float4 (&r)[RITS], int s, int c0) {
        const float *sp = p.src_ptr[s] + c0;
        const size_t ld = p.src_ld[s];
#pragma unroll
        for (int it = 0; it < RITS; ++it) r[it] = *reinterpret_cast<const float4 *>(sp + (size_t)g_pix[it] * ld + g_c4[it]);
    };
    auto store_raw = [&](const float4 (&r)[RITS]) {
#pragma unroll
        for (int it = 0; it < RITS; ++it) {
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
    const float *u_base = p.uw + ((size_t)(cls * 16 + (tid >> 7)) * p.cin_pad + u_k) * p.cout + (u_ok ? co0 + u_q : 0);
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
    const int t_src = ((G::OT * (t_tile >> 3)) * G::RW + G::OT * (t_tile & 7)) * WN_CKP + t_c;
    const int t_dst = t_tile * WN_CKP + t_c;
    auto transform = [&](float *vdst) {
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = raw[t_src + (i * G::RW + j) * WN_CKP];
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
    float *eb = lds;  // MODE 0: [i = wave][b 2][tile][64 + 1]   MODE 1: [i = wave][b 3][tile][32 + 1] per 32-channel pass
    if constexpr (MODE == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float t0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
                const float t1 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
                eb[((wv * 2 + 0) * 32 + trow) * G::EPW + nt * 32 + l31] = t0;
                eb[((wv * 2 + 1) * 32 + trow) * G::EPW + nt * 32 + l31] = t1;
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
                for (int i = 0; i < 4; ++i) e[i][0] = eb[((i * 2 + 0) * 32 + t) * G::EPW + col], e[i][1] = eb[((i * 2 + 1) * 32 + t) * G::EPW + col];
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
    } else {
        // F(3x3,2x2): three column outputs per row-wave, three rows out; LDS holds one 32-channel half at a time
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (nt) __syncthreads();  // the first half has been read
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float m1 = acc[1][nt][r], m2 = acc[2][nt][r];
                eb[((wv * 3 + 0) * 32 + trow) * G::EPW + l31] = acc[0][nt][r] + m1 + m2;
                eb[((wv * 3 + 1) * 32 + trow) * G::EPW + l31] = m1 - m2;
                eb[((wv * 3 + 2) * 32 + trow) * G::EPW + l31] = m1 + m2 - acc[3][nt][r];
            }
            __syncthreads();
            const int col = tid & 31;           // lane = channel: 128-B coalesced stores per output pixel
            const int co = co0 + nt * 32 + col;
            const float bias = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
            const int OH = 2 * p.H, OW = 2 * p.W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = (tid >> 5) + 8 * k;
                float e[4][3];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b) e[i][b] = eb[((i * 3 + b) * 32 + t) * G::EPW + col];
                const int yy = y0 + 3 * (t >> 3), xx = x0 + 3 * (t & 7);
                if (co < p.cout) {
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const float o0 = e[0][b] + e[1][b] + e[2][b], o1 = e[1][b] - e[2][b], o2 = e[1][b] + e[2][b] - e[3][b];
                        const int ox = 2 * (xx + b) + px;
                        if (xx + b < p.W) {
                            float *o = p.out + ((size_t)(n * OH + 2 * yy + py) * OW + ox) * p.out_ld + co;
                            const size_t rs = (size_t)2 * OW * p.out_ld;  // next class row = two output rows
                            if (yy < p.H) o[0] = act_apply(o0 + bias, p.act);
                            if (yy + 1 < p.H) o[rs] = act_apply(o1 + bias, p.act);
                            if (yy + 2 < p.H) o[2 * rs] = act_apply(o2 + bias, p.act);
                        }
                    }
                }
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

// MODE 1 weights: U[cls][16][cin_pad][cout] = G g G^T of the 2x2 sub-pixel kernels P[cls][dy*2+dx][cin_pad][cout] (pack.hip)
__global__ void wino_pack_ct4_kernel(const float *__restrict__ pk, float *__restrict__ uw, size_t plane /* cin_pad*cout */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cls = blockIdx.y;
    if (i >= plane) return;
    float g[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) g[r][s] = pk[(size_t)(cls * 4 + r * 2 + s) * plane + i];
    float u[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) u[0][s] = g[0][s], u[1][s] = 0.5f * (g[0][s] + g[1][s]), u[2][s] = 0.5f * (g[0][s] - g[1][s]), u[3][s] = g[1][s];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *o = uw + (size_t)(cls * 16 + r * 4) * plane + i;
        o[0] = u[r][0], o[plane] = 0.5f * (u[r][0] + u[r][1]), o[2 * plane] = 0.5f * (u[r][0] - u[r][1]), o[3 * plane] = u[r][1];
    }
}

template <int MODE>
static int wino_launch(WinoParams &p, int cout, const ProfHint &ph, hipStream_t st) {
    using G = WinoGeo<MODE>;
    p.tiles_x = (p.W + G::TW - 1) / G::TW, p.tiles_y = (p.H + G::TH - 1) / G::TH;
    p.ntiles = (unsigned)(p.tiles_x * p.tiles_y * p.N);
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wino_kernel<%d>): %s", MODE, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    ProfScope prof(MODE == 0 ? KID_CONV_WINO : KID_CONV_WINO_CT4, ph.flops, ph.bytes, st);
    hipLaunchKernelGGL(wino_kernel<MODE>, dim3(p.ntiles, (cout + 63) / 64, MODE == 1 ? 4 : 1), dim3(256), G::LDS_BYTES, st, p);
    return check_launch("wino_kernel");
}

// Called by conv2d_fwd_impl for eligible launches (K3S1 / CONVT_K3S1: F(2x2,3x3); CONVT_K4S2: F(3x3,2x2) per parity class);
// returns PWS_OK after launching, or a negative error.
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
    static const int ablate = getenv("PWS_WINO_ABLATE") ? atoi(getenv("PWS_WINO_ABLATE")) : 0;
    p.ablate = ablate;
    return a->kind == PWS_CONVT_K4S2 ? wino_launch<1>(p, a->cout, ph, st) : wino_launch<0>(p, a->cout, ph, st);
}

}  // namespace pws

extern "C" size_t pws_packed_wino_ct4_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return (size_t)4 * 16 * ((cin + 15) / 16 * 16) * cout;
}

extern "C" int pws_pack_conv_weight_wino_ct4(const float *w_packed, float *w_wino, int cin, int cout, pws_stream_t stream) {
    PWS_REQUIRE(w_packed && w_wino && cin > 0 && cout > 0, "pws_pack_conv_weight_wino_ct4: bad arguments");
    const size_t plane = (size_t)((cin + 15) / 16 * 16) * cout;
    hipLaunchKernelGGL(pws::wino_pack_ct4_kernel, dim3((unsigned)((plane + 255) / 256), 4), dim3(256), 0, pws::as_stream(stream),
                       w_packed, w_wino, plane);
    return pws::check_launch("wino_pack_ct4_kernel");
}

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

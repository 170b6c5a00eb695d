// Implicit-GEMM convolution / transposed convolution for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), NHWC activations, fused bias + activation, virtual concat on the read side.
//
// Replaces the ATen conv2d / conv_transpose2d + LeakyReLU / ReLU + torch.cat sequence of the reference
// blocks (reference lib/networks_cascading.py:245-350).
//
// GEMM view:  M = output pixels (n,y,x)   N = cout   K = taps x cin.
//   * A workgroup owns a TN x TH x TW block of output pixels (BM = TN*TH*TW) and BN = 64 output channels.
//   * K is walked in chunks of CK input channels.  Per chunk the (halo'd) input tile [pixels][CK] and the
//     weight slab [taps][CK][BN] are staged in LDS once and reused by all taps (an input pixel is read from
//     HBM/L2 ~1.3x instead of 9x).  LDS rows are padded to CK+1 floats so that the MFMA A-fragment read
//     (lane = pixel, one float per lane) is bank-conflict-free.
//   * Chunk c+1 is fetched global->registers while chunk c is being multiplied (the f32 MFMA is 64 cycles
//     per instruction, so a chunk is thousands of cycles of matrix work); two workgroups per CU cover the rest.
//   * A "virtual concat": the input channels come from up to 4 NHWC sources with their own pixel strides,
//     so torch.cat never copies.  The first layer reads the reference's NCHW input directly.
//   * ConvTranspose2d k4 s2 p1 runs as 4 independent sub-pixel 2x2 convolutions (one output parity class per
//     workgroup), ConvTranspose2d k3 s1 p1 as a 3x3 convolution with flipped taps (done at pack time).
//   * Tile shape is picked per launch so that the grid fills 256 CUs; the deep, small-spatial layers
//     (<= 16x16 outputs: M is tiny, K is 2304..16384, the weights are the traffic) additionally split K over
//     workgroups: every split streams its own slice of the weights and writes an fp32 partial tile, and a
//     small reduce kernel adds the partials, the bias and the activation (deterministic; no atomics).
//
// Numerics: fp32 in / fp32 accumulate; v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain in k order.
#include "common.h"

namespace pws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKParams {
    const float *src_ptr[4];
    int src_c[4];
    int src_ld[4];
    int nsrc;
    int N, H, W;   // input
    int LH, LW;    // logical output extent walked by the tiles (conv: OH,OW ; convT k4s2: H,W)
    int OH, OW;    // output tensor extent
    int cin_pad;   // rows per tap in the packed weights
    int cout;
    const float *w;
    const float *bias;
    float *out;     // final output, or the partial buffer when ksplit > 1
    int out_ld;
    int act;
    int tiles_x, tiles_y;
    unsigned ntiles;
    int nclasses;   // 4 for convT k4s2, else 1
    int ksplit;     // >= 1
    int chunks_per_split;
    size_t split_stride;  // floats between consecutive partial buffers
};

template <int KS_, int STRIDE_, int PAD_, bool CONVT_, int TH_, int TW_, int TN_, int CK_, int WM_, int WN_, int MT_,
          int NT_, bool NCHW_ = false>
struct ConvCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, PAD = PAD_, TH = TH_, TW = TW_, TN = TN_, CK = CK_;
    static constexpr bool CONVT = CONVT_, NCHW = NCHW_;
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int BM = TH * TW * TN, BN = 32 * NT * WN;
    static_assert(BM == 32 * MT * WM, "tile pixels must equal the M extent of the wave grid");
    static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    static constexpr int PIX = TN * IH * IW;
    static constexpr int CKP = CK + 1;
    static constexpr int TAPS = KS * KS;
    static constexpr int LDS_IN = (PIX * CKP + 3) / 4 * 4;  // floats, keeps the weight slab 16-B aligned
    static constexpr int LDS_W = TAPS * CK * BN;
    static constexpr int LDS_BYTES = (LDS_IN + LDS_W) * 4;
    static constexpr int C4 = CK / 4;
    static constexpr int ITEMS_IN = (PIX * C4 + THREADS - 1) / THREADS;
    static constexpr int ITEMS_W = (TAPS * CK * (BN / 4) + THREADS - 1) / THREADS;
    static constexpr int ITEMS_NCHW = (PIX * CK + THREADS - 1) / THREADS;
};

template <class C>
__global__ void __launch_bounds__(C::THREADS, 2) conv_mfma_kernel(const ConvKParams p) {
    extern __shared__ float lds[];
    float *lds_in = lds;
    float *lds_w = lds + C::LDS_IN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wv / C::WN, wn = wv % C::WN;

    // ---- which tile / output-channel group / parity class / K split
    const unsigned tile = xcd_remap(blockIdx.x, p.ntiles);
    const int tx_i = tile % p.tiles_x;
    const int ty_i = (tile / p.tiles_x) % p.tiles_y;
    const int tn_i = tile / (p.tiles_x * p.tiles_y);
    const int n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
    const int co0 = blockIdx.y * C::BN;
    const int cls = C::CONVT ? (int)(blockIdx.z & 3) : 0;
    const int split = C::CONVT ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    const int py = cls >> 1, px = cls & 1;
    const int pad_y = C::CONVT ? 1 - py : C::PAD;
    const int pad_x = C::CONVT ? 1 - px : C::PAD;
    const int iy0 = y0 * C::STRIDE - pad_y, ix0 = x0 * C::STRIDE - pad_x;

    // ---- hoisted per-thread staging descriptors (pixel decode is chunk independent)
    int g_pix[C::ITEMS_IN];   // global pixel index (n*H+iy)*W+ix or -1
    int l_off[C::ITEMS_IN];   // LDS float offset
    int g_c4[C::ITEMS_IN];
#pragma unroll
    for (int it = 0; it < C::ITEMS_IN; ++it) {
        const int item = tid + it * C::THREADS;
        const int pix = item / C::C4, c4 = item % C::C4;
        const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
        const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
        const bool ok = item < C::PIX * C::C4 && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = ok ? (n * p.H + iy) * p.W + ix : -1;
        l_off[it] = item < C::PIX * C::C4 ? pix * C::CKP + c4 * 4 : -1;
        g_c4[it] = c4 * 4;
    }
    float4 r_in[C::NCHW ? 1 : C::ITEMS_IN];
    float4 r_w[C::ITEMS_W];
    float r_nchw[C::NCHW ? C::ITEMS_NCHW : 1];

    auto load_chunk = [&](int s, int c0, int wrow) {
        if constexpr (C::NCHW) {
            const int Creal = p.src_c[0];
#pragma unroll
            for (int it = 0; it < C::ITEMS_NCHW; ++it) {
                const int item = tid + it * C::THREADS;
                const int c = item / C::PIX, pix = item % C::PIX;
                const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                const bool ok = item < C::PIX * C::CK && (c0 + c) < Creal && n < p.N && iy >= 0 && iy < p.H && ix >= 0 &&
                                ix < p.W;
                r_nchw[it] = ok ? p.src_ptr[0][((size_t)(n * Creal + c0 + c) * p.H + iy) * p.W + ix] : 0.f;
            }
        } else {
            const float *sp = p.src_ptr[s] + c0;
            const size_t ld = p.src_ld[s];
#pragma unroll
            for (int it = 0; it < C::ITEMS_IN; ++it) {
                r_in[it] = g_pix[it] >= 0 ? *reinterpret_cast<const float4 *>(sp + (size_t)g_pix[it] * ld + g_c4[it])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) {
            const int item = tid + it * C::THREADS;
            const int row = item / (C::BN / 4), q = item % (C::BN / 4);
            const int tap = row / C::CK, c = row % C::CK;
            const bool ok = item < C::TAPS * C::CK * (C::BN / 4) && (co0 + q * 4) < p.cout;
            r_w[it] = ok ? *reinterpret_cast<const float4 *>(
                               p.w + ((size_t)(cls * C::TAPS + tap) * p.cin_pad + wrow + c) * p.cout + co0 + q * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&]() {
        if constexpr (C::NCHW) {
#pragma unroll
            for (int it = 0; it < C::ITEMS_NCHW; ++it) {
                const int item = tid + it * C::THREADS;
                const int c = item / C::PIX, pix = item % C::PIX;
                if (item < C::PIX * C::CK) lds_in[pix * C::CKP + c] = r_nchw[it];
            }
        } else {
#pragma unroll
            for (int it = 0; it < C::ITEMS_IN; ++it) {
                if (l_off[it] >= 0) {
                    float *d = lds_in + l_off[it];
                    d[0] = r_in[it].x, d[1] = r_in[it].y, d[2] = r_in[it].z, d[3] = r_in[it].w;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) {
            const int item = tid + it * C::THREADS;
            if (item < C::TAPS * C::CK * (C::BN / 4)) *reinterpret_cast<float4 *>(lds_w + item * 4) = r_w[it];
        }
    };

    // ---- accumulators and fragment bases
    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    int a_base[C::MT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int m = (wm * C::MT + mt) * 32 + l31;
        const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
        a_base[mt] = ((tn * C::IH + ty * C::STRIDE) * C::IW + tx * C::STRIDE) * C::CKP + hi;
    }
    const int b_base = hi * C::BN + wn * C::NT * 32 + l31;

    // ---- K loop over this split's (source, channel chunk) range
    int total_chunks = 0;
    if constexpr (C::NCHW) {
        total_chunks = p.cin_pad / C::CK;
    } else {
        for (int s = 0; s < p.nsrc; ++s) total_chunks += p.src_c[s] / C::CK;
    }
    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(total_chunks, ch_begin + p.chunks_per_split);
    int s = 0, c0 = ch_begin * C::CK, wrow = ch_begin * C::CK;
    if constexpr (!C::NCHW) {
        while (s < p.nsrc - 1 && c0 >= p.src_c[s]) c0 -= p.src_c[s], ++s;
    }
    if (ch_begin < ch_end) load_chunk(s, c0, wrow);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();  // everyone finished reading the previous chunk from LDS
        store_chunk();
        __syncthreads();
        // advance and prefetch the next chunk into registers while this one is multiplied
        c0 += C::CK, wrow += C::CK;
        if (!C::NCHW && c0 >= p.src_c[s]) ++s, c0 = 0;
        if (ch + 1 < ch_end) load_chunk(s, c0, wrow);

#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            const int toff = ((tap / C::KS) * C::IW + (tap % C::KS)) * C::CKP;
#pragma unroll
            for (int kk = 0; kk < C::CK / 2; ++kk) {
                float a[C::MT], b[C::NT];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) a[mt] = lds_in[a_base[mt] + toff + 2 * kk];
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt) b[nt] = lds_w[b_base + (tap * C::CK + 2 * kk) * C::BN + nt * 32];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + activation (or raw partial sums when K is split), NHWC store
    //      (32 lanes = 32 consecutive channels = 128 B per pixel)
    const bool partial = p.ksplit > 1;
    float *outp = p.out + (partial ? (size_t)split * p.split_stride : 0);
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
        const int co = co0 + (wn * C::NT + nt) * 32 + l31;
        const bool co_ok = co < p.cout;
        const float bias = (co_ok && p.bias && !partial) ? p.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                if (co_ok && n < p.N && y < p.LH && x < p.LW) {
                    const int oy = C::CONVT ? 2 * y + py : y, ox = C::CONVT ? 2 * x + px : x;
                    const float v = partial ? acc[mt][nt][r] : act_apply(acc[mt][nt][r] + bias, p.act);
                    outp[((size_t)(n * p.OH + oy) * p.OW + ox) * p.out_ld + co] = v;
                }
            }
        }
    }
}

// out[i] = act(sum_s partial[s][i] + bias[i % cout]); `total` floats, dense (out_ld == cout), float4 per lane
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float *__restrict__ partial, int ksplit, size_t stride,
                                                            const float *__restrict__ bias, int cout, int act,
                                                            float *__restrict__ out, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 a = reinterpret_cast<const float4 *>(partial)[i];
    for (int s = 1; s < ksplit; ++s) {
        const float4 b = reinterpret_cast<const float4 *>(partial + (size_t)s * stride)[i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    const int co = (int)((i * 4) % cout);
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + co);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    a.x = act_apply(a.x, act), a.y = act_apply(a.y, act), a.z = act_apply(a.z, act), a.w = act_apply(a.w, act);
    reinterpret_cast<float4 *>(out)[i] = a;
}

// ------------------------------------------------------------------------------------------------ host side
struct ProfInfo {
    double flops, bytes;
};

struct TileChoice {  // one instantiation, described for the selector
    int th, tw, tn, ck, bn;
    int kid;
    int (*launch)(ConvKParams &, hipStream_t, const ProfInfo &);
};

template <class C, int KID>
static int launch_cfg(ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_mfma_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    dim3 grid(kp.ntiles, (kp.cout + C::BN - 1) / C::BN, kp.nclasses * kp.ksplit);
    hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(C::THREADS), C::LDS_BYTES, st, kp);
    return check_launch("conv_mfma_kernel");
}

template <class C, int KID>
static constexpr TileChoice choice() {
    return TileChoice{C::TH, C::TW, C::TN, C::CK, C::BN, KID, &launch_cfg<C, KID>};
}

//                        KS S  P  convT  TH  TW  TN  CK WM WN MT NT
using K3S1_T256 = ConvCfg<3, 1, 1, false, 16, 16, 1, 16, 4, 1, 2, 2>;
using K3S1_T128 = ConvCfg<3, 1, 1, false, 8, 16, 1, 16, 4, 1, 1, 2>;
using K3S1_T64 = ConvCfg<3, 1, 1, false, 8, 8, 1, 16, 2, 2, 1, 1>;
using K3S1_T64N4 = ConvCfg<3, 1, 1, false, 4, 4, 4, 16, 2, 2, 1, 1>;
using K3S1_T64N16 = ConvCfg<3, 1, 1, false, 2, 2, 16, 16, 2, 2, 1, 1>;
using K3S2_T256 = ConvCfg<3, 2, 1, false, 16, 16, 1, 8, 4, 1, 2, 2>;
using K3S2_T128 = ConvCfg<3, 2, 1, false, 8, 16, 1, 16, 4, 1, 1, 2>;
using K3S2_T64 = ConvCfg<3, 2, 1, false, 8, 8, 1, 16, 2, 2, 1, 1>;
using K3S2_T64N4 = ConvCfg<3, 2, 1, false, 4, 4, 4, 16, 2, 2, 1, 1>;
using K3S2_T64N16 = ConvCfg<3, 2, 1, false, 2, 2, 16, 16, 2, 2, 1, 1>;
using K5S1_T256 = ConvCfg<5, 1, 2, false, 16, 16, 1, 8, 4, 1, 2, 2>;
using K5S1_T256_NCHW = ConvCfg<5, 1, 2, false, 16, 16, 1, 8, 4, 1, 2, 2, true>;
using CT4_T256 = ConvCfg<2, 1, 0, true, 16, 16, 1, 16, 4, 1, 2, 2>;
using CT4_T128 = ConvCfg<2, 1, 0, true, 8, 16, 1, 16, 4, 1, 1, 2>;
using CT4_T64 = ConvCfg<2, 1, 0, true, 8, 8, 1, 16, 2, 2, 1, 1>;
using CT4_T64N4 = ConvCfg<2, 1, 0, true, 4, 4, 4, 16, 2, 2, 1, 1>;
using CT4_T64N16 = ConvCfg<2, 1, 0, true, 2, 2, 16, 16, 2, 2, 1, 1>;

// candidates ordered from the largest tile (best operand reuse) to the smallest
static const TileChoice kK3S1[] = {choice<K3S1_T256, KID_CONV_K3S1_BIG>(), choice<K3S1_T128, KID_CONV_K3S1_BIG>(),
                                   choice<K3S1_T64, KID_CONV_K3S1_SMALL>(), choice<K3S1_T64N4, KID_CONV_K3S1_SMALL>(),
                                   choice<K3S1_T64N16, KID_CONV_K3S1_SMALL>()};
static const TileChoice kK3S2[] = {choice<K3S2_T256, KID_CONV_K3S2_BIG>(), choice<K3S2_T128, KID_CONV_K3S2_BIG>(),
                                   choice<K3S2_T64, KID_CONV_K3S2_SMALL>(), choice<K3S2_T64N4, KID_CONV_K3S2_SMALL>(),
                                   choice<K3S2_T64N16, KID_CONV_K3S2_SMALL>()};
static const TileChoice kCT4[] = {choice<CT4_T256, KID_CONVT4_BIG>(), choice<CT4_T128, KID_CONVT4_BIG>(),
                                  choice<CT4_T64, KID_CONVT4_SMALL>(), choice<CT4_T64N4, KID_CONVT4_SMALL>(),
                                  choice<CT4_T64N16, KID_CONVT4_SMALL>()};
static const TileChoice kK5[] = {choice<K5S1_T256, KID_CONV_K5S1>()};
static const TileChoice kK5N[] = {choice<K5S1_T256_NCHW, KID_CONV_K5S1>()};

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

constexpr long kFillBlocks = 512;  // 256 CUs x 2 resident workgroups

// Pick the tile and the K split for one launch.
//  1. drop tiles that are mostly padding for this extent (a 16x16 tile on an 8x8 map);
//  2. take the largest remaining tile whose grid has >= kFillBlocks workgroups, else the one with the most workgroups;
//  3. if the grid is still < kFillBlocks and a workspace was given, split K (>= 2 chunks per split, <= 32 splits).
static int select_and_launch(const TileChoice *cands, int ncand, ConvKParams &kp, int cin_total, float *final_out,
                             float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi) {
    const TileChoice *best = nullptr;
    long best_blocks = -1;
    double best_score = -1.0;
    for (int i = 0; i < ncand; ++i) {
        const TileChoice &c = cands[i];
        const long tiles = cdiv(kp.LW, c.tw) * cdiv(kp.LH, c.th) * cdiv(kp.N, c.tn);
        const double useful = (double)kp.N * kp.LH * kp.LW / ((double)tiles * c.th * c.tw * c.tn);
        if (useful < 0.45 && i + 1 < ncand) continue;
        // sub-8 spatial tiles exist for maps that are themselves tiny; on a larger map their halo re-reads dominate
        if (c.th < 8 && c.th < kp.LH && i > 0 && best) continue;
        const long blocks = tiles * cdiv(kp.cout, c.bn) * kp.nclasses;
        if (blocks >= kFillBlocks) {
            best = &c, best_blocks = blocks;
            break;
        }
        if (blocks * useful > best_score) best = &c, best_blocks = blocks, best_score = blocks * useful;
    }
    const TileChoice &c = *best;
    kp.tiles_x = (int)cdiv(kp.LW, c.tw), kp.tiles_y = (int)cdiv(kp.LH, c.th);
    kp.ntiles = (unsigned)(kp.tiles_x * kp.tiles_y * cdiv(kp.N, c.tn));
    const int total_chunks = cin_total / c.ck;
    int ksplit = 1;
    const size_t out_floats = (size_t)kp.N * kp.OH * kp.OW * kp.cout;
    if (best_blocks < kFillBlocks && ws && kp.out_ld == kp.cout && kp.cout % 4 == 0 && total_chunks >= 4) {
        long want = cdiv(kFillBlocks, best_blocks);
        if (want > 32) want = 32;
        if (want > total_chunks / 2) want = total_chunks / 2;
        while (want > 1 && (size_t)want * out_floats > ws_floats) --want;
        ksplit = (int)want;
    }
    kp.ksplit = ksplit < 1 ? 1 : ksplit;
    kp.chunks_per_split = (int)cdiv(total_chunks, kp.ksplit);
    kp.ksplit = (int)cdiv(total_chunks, kp.chunks_per_split);  // no empty splits
    kp.split_stride = out_floats;
    kp.out = kp.ksplit > 1 ? ws : final_out;
    ProfScope prof(c.kid, pi.flops, pi.bytes, st);  // covers the split-K reduce as well
    int rc = c.launch(kp, st, pi);
    if (rc != PWS_OK || kp.ksplit == 1) return rc;
    const size_t total4 = out_floats / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv((long)total4, 256)), dim3(256), 0, st, ws, kp.ksplit, out_floats,
                       kp.bias, kp.cout, kp.act, final_out, total4);
    return check_launch("splitk_reduce_kernel");
}

int conv2d_fwd_impl(const pws_conv_args *a, hipStream_t st) {
    PWS_REQUIRE(a != nullptr, "pws_conv2d_fwd: args is NULL");
    PWS_REQUIRE(a->n >= 0 && a->h > 0 && a->w > 0, "pws_conv2d_fwd: bad n/h/w %d/%d/%d", a->n, a->h, a->w);
    PWS_REQUIRE(a->nsrc >= 1 && a->nsrc <= 4, "pws_conv2d_fwd: nsrc %d not in 1..4", a->nsrc);
    PWS_REQUIRE(a->cout > 0 && a->cout % 4 == 0, "pws_conv2d_fwd: cout %d must be a positive multiple of 4", a->cout);
    PWS_REQUIRE(a->out && a->w_packed, "pws_conv2d_fwd: NULL out / w_packed");
    PWS_REQUIRE(a->out_ld >= a->cout, "pws_conv2d_fwd: out_ld %d < cout %d", a->out_ld, a->cout);
    PWS_REQUIRE(a->act >= PWS_ACT_NONE && a->act <= PWS_ACT_RELU, "pws_conv2d_fwd: bad act %d", a->act);
    PWS_REQUIRE(a->ws == nullptr || (reinterpret_cast<size_t>(a->ws) & 15) == 0, "pws_conv2d_fwd: ws must be 16-B aligned");
    if (a->n == 0) return PWS_OK;

    ConvKParams kp{};
    int cin = 0;
    kp.nsrc = a->nsrc;
    const bool nchw = a->src_nchw != 0;
    if (nchw) {
        PWS_REQUIRE(a->kind == PWS_CONV_K5S1 && a->nsrc == 1, "pws_conv2d_fwd: NCHW source only for the k5 first layer");
        PWS_REQUIRE(a->src[0].ptr && a->src[0].channels > 0, "pws_conv2d_fwd: bad NCHW source");
        kp.src_ptr[0] = a->src[0].ptr, kp.src_c[0] = a->src[0].channels, kp.src_ld[0] = 0;
        cin = a->src[0].channels;
    } else {
        for (int s = 0; s < a->nsrc; ++s) {
            const pws_src &sr = a->src[s];
            PWS_REQUIRE(sr.ptr && sr.channels > 0 && sr.channels % 16 == 0 && sr.ld >= sr.channels && sr.ld % 4 == 0 &&
                            (reinterpret_cast<size_t>(sr.ptr) & 15) == 0,
                        "pws_conv2d_fwd: source %d (channels %d, ld %d) must be 16-B aligned, channels %% 16 == 0, ld %% 4 == 0",
                        s, sr.channels, sr.ld);
            kp.src_ptr[s] = sr.ptr, kp.src_c[s] = sr.channels, kp.src_ld[s] = sr.ld;
            cin += sr.channels;
        }
    }
    kp.cin_pad = (cin + 15) / 16 * 16;
    kp.N = a->n, kp.H = a->h, kp.W = a->w;
    kp.cout = a->cout, kp.w = a->w_packed, kp.bias = a->bias, kp.out_ld = a->out_ld, kp.act = a->act;
    kp.nclasses = 1;
    PWS_REQUIRE((size_t)a->n * a->h * a->w < (1u << 31), "pws_conv2d_fwd: n*h*w too large for 32-bit pixel indices");
    float *ws = static_cast<float *>(a->ws);
    const size_t ws_floats = a->ws_bytes / sizeof(float);

    // algorithmic work of this launch (real channels, each tensor touched once)
    auto info = [&](int k2, double out_pix) {
        ProfInfo pi;
        pi.flops = 2.0 * out_pix * a->cout * (double)cin * k2;
        pi.bytes = 4.0 * ((double)a->n * a->h * a->w * cin + out_pix * a->cout + (double)k2 * cin * a->cout);
        return pi;
    };
    switch (a->kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        kp.OH = kp.LH = a->h, kp.OW = kp.LW = a->w;
        return select_and_launch(kK3S1, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, info(9, (double)a->n * a->h * a->w));
    case PWS_CONV_K3S2:
        kp.OH = kp.LH = (a->h + 2 - 3) / 2 + 1, kp.OW = kp.LW = (a->w + 2 - 3) / 2 + 1;
        return select_and_launch(kK3S2, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, info(9, (double)a->n * kp.OH * kp.OW));
    case PWS_CONV_K5S1:
        kp.OH = kp.LH = a->h, kp.OW = kp.LW = a->w;
        return select_and_launch(nchw ? kK5N : kK5, 1, kp, kp.cin_pad, a->out, nullptr, 0, st,
                                 info(25, (double)a->n * a->h * a->w));
    case PWS_CONVT_K4S2: {
        kp.LH = a->h, kp.LW = a->w, kp.OH = 2 * a->h, kp.OW = 2 * a->w, kp.nclasses = 4;
        ProfInfo pi = info(4, (double)a->n * kp.OH * kp.OW);  // every output pixel sees 2x2 taps
        pi.bytes += 4.0 * 12.0 * cin * a->cout;              // all 16 taps of the weight are read
        return select_and_launch(kCT4, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, pi);
    }
    default:
        set_error("pws_conv2d_fwd: kind %d is not a runnable conv kind", a->kind);
        return PWS_EINVAL;
    }
}

}  // namespace pws

extern "C" int pws_conv2d_fwd(const pws_conv_args *args, pws_stream_t stream) {
    return pws::conv2d_fwd_impl(args, pws::as_stream(stream));
}

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
#include "conv_common.h"

namespace pws {

// SUBPIX: 0 = ordinary convolution; 1 = ConvTranspose2d k4 s2 p1 forward (output parity class (py,px) is a 2x2
// convolution whose window starts at (y+py-1, x+px-1)); 2 = data gradient of Conv2d k3 s2 p1 (class (py,px) is a 2x2
// window starting at (y, x) of which the even parity uses only the first tap).
template <int KS_, int STRIDE_, int PAD_, int SUBPIX_, int TH_, int TW_, int TN_, int CK_, int WM_, int WN_, int MT_,
          int NT_, bool NCHW_ = false>
struct ConvCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, PAD = PAD_, TH = TH_, TW = TW_, TN = TN_, CK = CK_;
    static constexpr int SUBPIX = SUBPIX_;
    static constexpr bool CONVT = SUBPIX_ != 0, NCHW = NCHW_;
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
    unsigned tile;
    int cls, split;
    conv_block_coords(p, C::CONVT, tile, cls, split);
    const int tx_i = tile % p.tiles_x;
    const int ty_i = (tile / p.tiles_x) % p.tiles_y;
    const int tn_i = tile / (p.tiles_x * p.tiles_y);
    const int n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
    const int co0 = blockIdx.y * C::BN;
    const int py = cls >> 1, px = cls & 1;
    const int pad_y = C::SUBPIX == 1 ? 1 - py : (C::SUBPIX == 2 ? 0 : C::PAD);
    const int pad_x = C::SUBPIX == 1 ? 1 - px : (C::SUBPIX == 2 ? 0 : C::PAD);
    const int iy0 = y0 * C::STRIDE - pad_y, ix0 = x0 * C::STRIDE - pad_x;

    // ---- hoisted per-thread staging descriptors (pixel decode is chunk independent).
    // Loads are UNCONDITIONAL (out-of-image / out-of-range items read a valid dummy address and are zeroed by a select
    // afterwards): a conditional load makes hipcc branch around it and wait vmcnt(0) before the next one, which
    // serialises the whole prefetch (cdna_hip_programming.md, "register or load" trap).
    int g_pix[C::ITEMS_IN];   // global pixel index (n*H+iy)*W+ix, 0 when masked
    int l_off[C::ITEMS_IN];   // LDS float offset, -1 for items past the tile
    int g_c4[C::ITEMS_IN];
    unsigned ok_mask = 0;
#pragma unroll
    for (int it = 0; it < C::ITEMS_IN; ++it) {
        const int item = tid + it * C::THREADS;
        const int pix = item / C::C4, c4 = item % C::C4;
        const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
        const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
        const bool ok = item < C::PIX * C::C4 && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = ok ? (n * p.H + iy) * p.W + ix : 0;
        ok_mask |= ok ? (1u << it) : 0u;
        l_off[it] = item < C::PIX * C::C4 ? pix * C::CKP + c4 * 4 : -1;
        g_c4[it] = c4 * 4;
    }
    static_assert(C::ITEMS_IN <= 32 && C::ITEMS_W <= 32, "mask width");
    // weight-slab items: offset inside one [tap][CK][cout] slab, 0 (a valid address) when masked
    int w_off[C::ITEMS_W];
    unsigned w_mask = 0;
#pragma unroll
    for (int it = 0; it < C::ITEMS_W; ++it) {
        const int item = tid + it * C::THREADS;
        const int row = item / (C::BN / 4), q = item % (C::BN / 4);
        const int tap = row / C::CK, c = row % C::CK;
        const bool ok = item < C::TAPS * C::CK * (C::BN / 4) && (co0 + q * 4) < p.cout;
        w_off[it] = ok ? (tap * p.cin_pad + c) * p.cout + co0 + q * 4 : 0;
        w_mask |= ok ? (1u << it) : 0u;
    }
    const float *w_cls = p.w + (size_t)cls * C::TAPS * p.cin_pad * p.cout;
    float4 r_in[C::NCHW ? 1 : C::ITEMS_IN];
    float4 r_w[C::ITEMS_W];
    float r_nchw[C::NCHW ? C::ITEMS_NCHW : 1];

    auto load_chunk = [&](int s, int c0, int wrow) {
        if constexpr (C::NCHW) {
            const int Creal = p.src_c[0];
            const size_t sstride = p.src_ld[0] ? (size_t)p.src_ld[0] : (size_t)Creal * p.H * p.W;   // floats between samples
#pragma unroll
            for (int it = 0; it < C::ITEMS_NCHW; ++it) {
                const int item = tid + it * C::THREADS;
                const int c = item / C::PIX, pix = item % C::PIX;
                const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                const bool ok = item < C::PIX * C::CK && (c0 + c) < Creal && n < p.N && iy >= 0 && iy < p.H && ix >= 0 &&
                                ix < p.W;
                const size_t off = ok ? (size_t)n * sstride + ((size_t)(c0 + c) * p.H + iy) * p.W + ix : 0;
                const float v = p.src_ptr[0][off];  // unconditional load, masked by a select
                r_nchw[it] = ok ? v : 0.f;
            }
        } else {
            const float *sp = p.src_ptr[s] + c0;
            const size_t ld = p.src_ld[s];
#pragma unroll
            for (int it = 0; it < C::ITEMS_IN; ++it)
                r_in[it] = *reinterpret_cast<const float4 *>(sp + (size_t)g_pix[it] * ld + g_c4[it]);
        }
        const float *wp = w_cls + (size_t)wrow * p.cout;
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) r_w[it] = *reinterpret_cast<const float4 *>(wp + w_off[it]);
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
                    const bool ok = (ok_mask >> it) & 1u;
                    float *d = lds_in + l_off[it];
                    d[0] = ok ? r_in[it].x : 0.f, d[1] = ok ? r_in[it].y : 0.f, d[2] = ok ? r_in[it].z : 0.f,
                    d[3] = ok ? r_in[it].w : 0.f;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) {
            const int item = tid + it * C::THREADS;
            if (item < C::TAPS * C::CK * (C::BN / 4)) {
                const bool ok = (w_mask >> it) & 1u;
                *reinterpret_cast<float4 *>(lds_w + item * 4) =
                    ok ? r_w[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
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
            if (C::SUBPIX == 2 && ((!py && tap / C::KS) || (!px && tap % C::KS))) continue;  // block-uniform
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

    // ---- epilogue: bias + activation / gradient scatter (or raw partial sums when K is split), NHWC store
    //      (32 lanes = 32 consecutive channels = 128 B per pixel)
    const bool partial = p.ksplit > 1;
    float *part = p.out + (size_t)split * p.split_stride;
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
        const int co = co0 + (wn * C::NT + nt) * 32 + l31;
        const bool co_ok = co < p.cout;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                const int oy = C::CONVT ? 2 * y + py : y, ox = C::CONVT ? 2 * x + px : x;
                if (co_ok && n < p.N && y < p.LH && x < p.LW && oy < p.OH && ox < p.OW) {
                    const size_t pix = (size_t)(n * p.OH + oy) * p.OW + ox;
                    if (partial)
                        part[pix * p.cout + co] = acc[mt][nt][r];
                    else
                        epi_store(p, pix, co, acc[mt][nt][r]);
                }
            }
        }
    }
}

// Split-K reduce: sum the partial tiles (in split order: deterministic) and run the ordinary epilogue.
// partial[s] is dense [pixels][cout]; 4 consecutive channels per lane.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvKParams p, const float *__restrict__ partial,
                                                            size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 a = reinterpret_cast<const float4 *>(partial)[i];
    for (int s = 1; s < p.ksplit; ++s) {
        const float4 b = reinterpret_cast<const float4 *>(partial + (size_t)s * p.split_stride)[i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    const size_t pix = (i * 4) / p.cout;
    const int co = (int)((i * 4) % p.cout);
    if (p.io_bf16) {  // bf16 storage (conv_bf16.hip): channel pairs as dwords
        epi_store_pair16(p, pix, co, a.x, a.y), epi_store_pair16(p, pix, co + 2, a.z, a.w);
        return;
    }
    epi_store(p, pix, co, a.x), epi_store(p, pix, co + 1, a.y), epi_store(p, pix, co + 2, a.z), epi_store(p, pix, co + 3, a.w);
}

int launch_splitk_reduce(const ConvKParams &kp, const float *partial, size_t total4, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, kp, partial, total4);
    return check_launch("splitk_reduce_kernel");
}

// ------------------------------------------------------------------------------------------------ host side
template <class C, int KID>
static int launch_cfg(ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_mfma_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    const dim3 grid = conv_grid(kp, C::BN);
    hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(C::THREADS), C::LDS_BYTES, st, kp);
    return check_launch("conv_mfma_kernel");
}

template <class C, int KID>
static constexpr TileChoice choice() {
    return TileChoice{C::TH, C::TW, C::TN, C::CK, C::BN, KID, &launch_cfg<C, KID>};
}

//                        KS S  P  subpix TH  TW  TN  CK WM WN MT NT
using K3S1_T256 = ConvCfg<3, 1, 1, 0, 16, 16, 1, 16, 4, 1, 2, 2>;
using K3S1_T128 = ConvCfg<3, 1, 1, 0, 8, 16, 1, 16, 4, 1, 1, 2>;
using K3S1_T64 = ConvCfg<3, 1, 1, 0, 8, 8, 1, 16, 2, 2, 1, 1>;
using K3S1_T64N4 = ConvCfg<3, 1, 1, 0, 4, 4, 4, 16, 2, 2, 1, 1>;
using K3S1_T64N16 = ConvCfg<3, 1, 1, 0, 2, 2, 16, 16, 2, 2, 1, 1>;
using K3S2_T256 = ConvCfg<3, 2, 1, 0, 16, 16, 1, 8, 4, 1, 2, 2>;
using K3S2_T128 = ConvCfg<3, 2, 1, 0, 8, 16, 1, 16, 4, 1, 1, 2>;
using K3S2_T64 = ConvCfg<3, 2, 1, 0, 8, 8, 1, 16, 2, 2, 1, 1>;
using K3S2_T64N4 = ConvCfg<3, 2, 1, 0, 4, 4, 4, 16, 2, 2, 1, 1>;
using K3S2_T64N16 = ConvCfg<3, 2, 1, 0, 2, 2, 16, 16, 2, 2, 1, 1>;
using K5S1_T256 = ConvCfg<5, 1, 2, 0, 16, 16, 1, 8, 4, 1, 2, 2>;
using K5S1_T256_NCHW = ConvCfg<5, 1, 2, 0, 16, 16, 1, 8, 4, 1, 2, 2, true>;
using CT4_T256 = ConvCfg<2, 1, 0, 1, 16, 16, 1, 16, 4, 1, 2, 2>;
using CT4_T128 = ConvCfg<2, 1, 0, 1, 8, 16, 1, 16, 4, 1, 1, 2>;
using CT4_T64 = ConvCfg<2, 1, 0, 1, 8, 8, 1, 16, 2, 2, 1, 1>;
using CT4_T64N4 = ConvCfg<2, 1, 0, 1, 4, 4, 4, 16, 2, 2, 1, 1>;
using CT4_T64N16 = ConvCfg<2, 1, 0, 1, 2, 2, 16, 16, 2, 2, 1, 1>;

// data-gradient kernels: conv k4 s2 p1 (gradient of ConvTranspose2d k4 s2 p1) and the sub-pixel gradient of conv k3 s2 p1
using K4S2_T256 = ConvCfg<4, 2, 1, 0, 16, 16, 1, 8, 4, 1, 2, 2>;
using K4S2_T128 = ConvCfg<4, 2, 1, 0, 8, 16, 1, 8, 4, 1, 1, 2>;
using K4S2_T64 = ConvCfg<4, 2, 1, 0, 8, 8, 1, 8, 2, 2, 1, 1>;
using K4S2_T64N4 = ConvCfg<4, 2, 1, 0, 4, 4, 4, 8, 2, 2, 1, 1>;
using K4S2_T64N16 = ConvCfg<4, 2, 1, 0, 2, 2, 16, 8, 2, 2, 1, 1>;
using SP3_T256 = ConvCfg<2, 1, 0, 2, 16, 16, 1, 16, 4, 1, 2, 2>;
using SP3_T128 = ConvCfg<2, 1, 0, 2, 8, 16, 1, 16, 4, 1, 1, 2>;
using SP3_T64 = ConvCfg<2, 1, 0, 2, 8, 8, 1, 16, 2, 2, 1, 1>;
using SP3_T64N4 = ConvCfg<2, 1, 0, 2, 4, 4, 4, 16, 2, 2, 1, 1>;
using SP3_T64N16 = ConvCfg<2, 1, 0, 2, 2, 2, 16, 16, 2, 2, 1, 1>;

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
static const TileChoice kK4S2[] = {choice<K4S2_T256, KID_DGRAD_K4S2>(), choice<K4S2_T128, KID_DGRAD_K4S2>(),
                                   choice<K4S2_T64, KID_DGRAD_K4S2>(), choice<K4S2_T64N4, KID_DGRAD_K4S2>(),
                                   choice<K4S2_T64N16, KID_DGRAD_K4S2>()};
static const TileChoice kSP3[] = {choice<SP3_T256, KID_DGRAD_SP3>(), choice<SP3_T128, KID_DGRAD_SP3>(),
                                  choice<SP3_T64, KID_DGRAD_SP3>(), choice<SP3_T64N4, KID_DGRAD_SP3>(),
                                  choice<SP3_T64N16, KID_DGRAD_SP3>()};
static const TileChoice kK5[] = {choice<K5S1_T256, KID_CONV_K5S1>()};
static const TileChoice kK5N[] = {choice<K5S1_T256_NCHW, KID_CONV_K5S1>()};

int conv_bf16_fwd(int kind, ConvKParams &kp, int cin_total, float *out, float *ws, size_t ws_floats, hipStream_t st,
                  const ProfInfo &pi);  // conv_bf16.hip; 1 = not covered
int conv_bf16_dgrad(int kind, ConvKParams &kp, int cout_f, float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi);
// Parity ledger (tools/parity_budget.py): PWS_OPT_EXPERIMENT 3000 + mask takes ONE kernel family of the fp32 forward at a time back to the plain
// direct fp32 kernel (conv_mfma_kernel), so that the whole-network error can be attributed family by family.  Bits: 1 = first layer (no
// F(2x2,5x5), no planar-LDS kernel), 2 = the 3x3 stride-1 Winograd kernels, 4 = the transposed layers' F(2x2,2x2), 8 = conv_ringf_kernel,
// 16 = conv_skinny_kernel.  Measurement only: nothing in the product sets it.
static inline bool ledger_direct(int bit) { return g_experiment >= 3000 && g_experiment < 3032 && ((g_experiment - 3000) & bit) != 0; }
int conv_ringf_try(int kind, const ConvKParams &kp, hipStream_t st, const ProfInfo &pi);   // conv_ring_f32.hip; 1 = not covered
int conv_first_try(const ConvKParams &kp, hipStream_t st, const ProfInfo &pi);             // conv_first.hip; 1 = not covered
int wino5_first_try(const ConvKParams &kp, const float *u, hipStream_t st, const ProfInfo &pi);   // conv_first_wino.hip; 1 = not covered
int conv_skinny_try(int kind, ConvKParams &kp, float *final_out, float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi);  // conv_skinny.hip; 1 = not covered

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
        PWS_REQUIRE(a->src[0].ld >= 0, "pws_conv2d_fwd: NCHW source: ld is the sample stride in floats (0 = dense)");
        kp.src_ptr[0] = a->src[0].ptr, kp.src_c[0] = a->src[0].channels, kp.src_ld[0] = a->src[0].ld;
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
    const bool bf16 = a->math == PWS_MATH_BF16 && a->w_bf16 && !nchw;
    if (a->store == PWS_STORE_BF16) {
        PWS_REQUIRE(bf16, "pws_conv2d_fwd: bf16 storage needs math == PWS_MATH_BF16, w_bf16 and NHWC sources");
        PWS_REQUIRE(a->out_ld % 2 == 0 && a->cout % 2 == 0 && (reinterpret_cast<size_t>(a->out) & 3) == 0,
                    "pws_conv2d_fwd: bf16 storage needs an even cout / out_ld and a 4-byte aligned out");
        for (int s = 0; s < a->nsrc; ++s)
            PWS_REQUIRE(a->src[s].channels % 32 == 0 && a->src[s].ld % 8 == 0,
                        "pws_conv2d_fwd: bf16 storage needs channels %% 32 == 0 and ld %% 8 == 0 (source %d: %d, %d)", s,
                        a->src[s].channels, a->src[s].ld);
        kp.io_bf16 = 1;
        kp.epi16 = a->cout % 8 == 0 && a->out_ld % 8 == 0 && (reinterpret_cast<size_t>(a->out) & 15) == 0;
    }
    kp.w_bf = a->w_bf16, kp.kpad_bf = (kp.cin_pad + 31) / 32 * 32, kp.npad_bf = (a->cout + 63) / 64 * 64;
    if (a->out_sign) {
        PWS_REQUIRE(a->store == PWS_STORE_BF16 && a->cout % 8 == 0 && a->out_sign_ld >= a->cout / 8,
                    "pws_conv2d_fwd: out_sign needs bf16 storage, cout %% 8 == 0 and out_sign_ld >= cout / 8");
        kp.out_sign = a->out_sign, kp.out_sign_ld = a->out_sign_ld;
    }

    // algorithmic work of this launch (real channels, each tensor touched once)
    auto info = [&](int k2, double out_pix) {
        ProfInfo pi;
        pi.flops = 2.0 * out_pix * a->cout * (double)cin * k2;
        const double es = a->store == PWS_STORE_BF16 ? 2.0 : 4.0, ws_ = bf16 ? 2.0 : 4.0;   // activation / weight element size
        pi.bytes = es * ((double)a->n * a->h * a->w * cin + out_pix * a->cout) + ws_ * (double)k2 * cin * a->cout;
        return pi;
    };
    switch (a->kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1: {
        kp.OH = kp.LH = a->h, kp.OW = kp.LW = a->w;
        const ProfInfo pi = info(9, (double)a->n * a->h * a->w);
        if (bf16) {
            const int rc = conv_bf16_fwd(a->kind, kp, kp.cin_pad, a->out, ws, ws_floats, st, pi);
            if (rc != 1) return rc;
        }
        // Winograd F(2x2,3x3) when there is enough of the map to fill the chip (measured cross-over, tools/conv_bench.py:
        // 256->256 @32x32 x8: 104 -> 79 us; 512->512 @16x16 x8: 106 -> 138 us); the deep maps stay direct + split-K
        if (a->w_wring && !nchw && !bf16 && !ledger_direct(2)) {   // second-generation Winograd (persistent LDS ring) where whole 16 x 32 units fill the chip
            const int rc = wring_try(a, ProfHint{pi.flops, pi.bytes}, st);
            if (rc != 1) return rc;
        }
        const long wblocks = cdiv(a->w, 16) * cdiv(a->h, 16) * a->n * cdiv(a->cout, 64);
        // (measured: with <= 64 input channels the per-workgroup prologue/epilogue outweighs the saving unless the map is huge)
        if (a->w_wino && !nchw && a->h >= 16 && a->w >= 16 && wblocks >= 128 && (cin >= 128 || wblocks >= 2048) && !ledger_direct(2))
            return wino_k3s1_launch(a, ProfHint{pi.flops, pi.bytes}, st);
        if (!bf16 && !nchw) {   // the persistent LDS-ring kernel (exact fp32) where it is covered
            kp.out = a->out;
            int rc = ledger_direct(8) ? 1 : conv_ringf_try(a->kind, kp, st, pi);
            if (rc != 1) return rc;
            rc = ledger_direct(16) ? 1 : conv_skinny_try(a->kind, kp, a->out, ws, ws_floats, st, pi);   // the deep levels: one-shot weight fetch
            if (rc != 1) return rc;
        }
        return select_and_launch(kK3S1, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, pi);
    }
    case PWS_CONV_K3S2:
        kp.OH = kp.LH = (a->h + 2 - 3) / 2 + 1, kp.OW = kp.LW = (a->w + 2 - 3) / 2 + 1;
        if (bf16) {
            const int rc = conv_bf16_fwd(a->kind, kp, kp.cin_pad, a->out, ws, ws_floats, st, info(9, (double)a->n * kp.OH * kp.OW));
            if (rc != 1) return rc;
        }
        if (!bf16 && !nchw) {
            kp.out = a->out;
            int rc = ledger_direct(8) ? 1 : conv_ringf_try(a->kind, kp, st, info(9, (double)a->n * kp.OH * kp.OW));
            if (rc != 1) return rc;
            rc = ledger_direct(16) ? 1 : conv_skinny_try(a->kind, kp, a->out, ws, ws_floats, st, info(9, (double)a->n * kp.OH * kp.OW));
            if (rc != 1) return rc;
        }
        return select_and_launch(kK3S2, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, info(9, (double)a->n * kp.OH * kp.OW));
    case PWS_CONV_K5S1:
        kp.OH = kp.LH = a->h, kp.OW = kp.LW = a->w;
        if (bf16) {
            const int rc = conv_bf16_fwd(a->kind, kp, kp.cin_pad, a->out, nullptr, 0, st, info(25, (double)a->n * a->h * a->w));
            if (rc != 1) return rc;
        }
        if (nchw && !ledger_direct(1)) {   // the persistent planar-LDS kernel where it is covered (the generator's 256 x 256 windows)
            kp.out = a->out;
            if (a->w_wring) {   // Winograd F(2x2,5x5) where the transformed weights are given and whole 8 x 16 units fill the chip
                const int rcw = wino5_first_try(kp, static_cast<const float *>(a->w_wring), st, info(25, (double)a->n * a->h * a->w));
                if (rcw != 1) return rcw;
            }
            const int rc = conv_first_try(kp, st, info(25, (double)a->n * a->h * a->w));
            if (rc != 1) return rc;
        }
        return select_and_launch(nchw ? kK5N : kK5, 1, kp, kp.cin_pad, a->out, nullptr, 0, st,
                                 info(25, (double)a->n * a->h * a->w));
    case PWS_CONVT_K4S2: {
        kp.LH = a->h, kp.LW = a->w, kp.OH = 2 * a->h, kp.OW = 2 * a->w, kp.nclasses = 4;
        ProfInfo pi = info(4, (double)a->n * kp.OH * kp.OW);  // every output pixel sees 2x2 taps
        pi.bytes += (bf16 ? 2.0 : 4.0) * 12.0 * cin * a->cout;   // all 16 taps of the weight are read
        if (bf16) {
            const int rc = conv_bf16_fwd(a->kind, kp, kp.cin_pad, a->out, ws, ws_floats, st, pi);
            if (rc != 1) return rc;
        }
        // Winograd F(3x3,2x2) per parity class (4 classes x 12x24-pixel workgroups) when the caller packed the weights for it.
        // Measured (tools/conv_bench.py, N=8): 256->64 @128x128 646 vs 661 us direct, but 512->64 @64x64 473 vs 333 us -- the
        // 3-pixel tiles waste 21-41 % of a 64/32-pixel map and the 9-output epilogue is 2.25x the F(2x2,3x3) one -- so the
        // generator's executor does not pack these weights; the path stays available for large maps.
        if (a->w_wring && !nchw && !bf16 && !ledger_direct(4)) {   // Winograd F(2x2,2x2) per parity class on the LDS ring (whole 16 x 32 input units)
            const int rc = wring_try(a, ProfHint{pi.flops, pi.bytes}, st);
            if (rc != 1) return rc;
        }
        const long wb = cdiv(a->w, 24) * cdiv(a->h, 12) * a->n * cdiv(a->cout, 64) * 4;
        if (a->w_wino && a->h >= 24 && a->w >= 24 && wb >= 256 && !ledger_direct(4)) return wino_k3s1_launch(a, ProfHint{pi.flops, pi.bytes}, st);
        if (!bf16 && !nchw) {
            kp.out = a->out;
            int rc = ledger_direct(8) ? 1 : conv_ringf_try(a->kind, kp, st, pi);
            if (rc != 1) return rc;
            rc = ledger_direct(16) ? 1 : conv_skinny_try(a->kind, kp, a->out, ws, ws_floats, st, pi);
            if (rc != 1) return rc;
        }
        return select_and_launch(kCT4, 5, kp, kp.cin_pad, a->out, ws, ws_floats, st, pi);
    }
    default:
        set_error("pws_conv2d_fwd: kind %d is not a runnable conv kind", a->kind);
        return PWS_EINVAL;
    }
}

// Data gradient of one forward layer: dx (scattered over the forward layer's sources) from dy.
// Every case is again a convolution of dy, run by the same kernel with re-packed weights (pack.hip, dgrad layouts).
int conv2d_bwd_data_impl(const pws_conv_bwd_data_args *a, hipStream_t st) {
    PWS_REQUIRE(a != nullptr, "pws_conv2d_bwd_data: args is NULL");
    PWS_REQUIRE(a->n >= 0 && a->h > 0 && a->w > 0 && a->cout > 0 && a->cout % 16 == 0, "pws_conv2d_bwd_data: bad shape");
    PWS_REQUIRE(a->gout && a->w_dgrad && a->gout_ld >= a->cout && a->gout_ld % 4 == 0, "pws_conv2d_bwd_data: bad gout / weights");
    PWS_REQUIRE(a->ndst >= 1 && a->ndst <= 4, "pws_conv2d_bwd_data: ndst %d not in 1..4", a->ndst);
    if (a->n == 0) return PWS_OK;
    ConvKParams kp{};
    int oh = a->h, ow = a->w;  // forward OUTPUT extent = extent of gout
    if (a->kind == PWS_CONV_K3S2) oh = (a->h - 1) / 2 + 1, ow = (a->w - 1) / 2 + 1;
    if (a->kind == PWS_CONVT_K4S2) oh = 2 * a->h, ow = 2 * a->w;
    kp.nsrc = 1, kp.src_ptr[0] = a->gout, kp.src_c[0] = a->cout, kp.src_ld[0] = a->gout_ld;
    kp.N = a->n, kp.H = oh, kp.W = ow;
    kp.OH = a->h, kp.OW = a->w;  // dx has the forward INPUT extent
    kp.cin_pad = a->cout;
    int cin_f = 0;
    kp.ndst = a->ndst;
    for (int s = 0; s < a->ndst; ++s) {
        const pws_dst &d = a->dst[s];
        PWS_REQUIRE(d.ptr && d.channels > 0 && d.ld >= d.channels, "pws_conv2d_bwd_data: bad destination %d", s);
        kp.dst_ptr[s] = d.ptr, kp.dst_c0[s] = cin_f, kp.dst_c1[s] = cin_f + d.channels, kp.dst_ld[s] = d.ld;
        kp.dst_acc[s] = d.accumulate ? 1 : 0;
        if (d.act_y && d.act != PWS_ACT_NONE) {
            PWS_REQUIRE(a->store == PWS_STORE_BF16 && (d.act == PWS_ACT_LRELU || d.act == PWS_ACT_RELU),
                        "pws_conv2d_bwd_data: the fused act' (dst[%d].act_y) needs bf16 storage and PWS_ACT_LRELU / PWS_ACT_RELU", s);
            PWS_REQUIRE(d.act_y_ld >= d.channels && d.act_y_ld % 2 == 0 && (reinterpret_cast<size_t>(d.act_y) & 3) == 0,
                        "pws_conv2d_bwd_data: bad act_y / act_y_ld of destination %d", s);
            kp.dst_y[s] = d.act_y, kp.dst_y_ld[s] = d.act_y_ld, kp.dst_act[s] = d.act;
            if (d.act_sign) {
                PWS_REQUIRE(d.channels % 8 == 0 && d.act_sign_ld >= d.channels / 8, "pws_conv2d_bwd_data: bad act_sign / act_sign_ld of destination %d", s);
                kp.dst_sign[s] = d.act_sign, kp.dst_sign_ld[s] = d.act_sign_ld;
            }
        }
        cin_f += d.channels;
    }
    PWS_REQUIRE(cin_f % 4 == 0, "pws_conv2d_bwd_data: total destination channels %d must be a multiple of 4", cin_f);
    kp.cout = cin_f, kp.w = a->w_dgrad, kp.bias = nullptr, kp.act = PWS_ACT_NONE, kp.out_ld = cin_f, kp.nclasses = 1;
    PWS_REQUIRE((size_t)a->n * oh * ow < (1u << 31) && (size_t)a->n * a->h * a->w < (1u << 31), "pws_conv2d_bwd_data: too large");
    float *ws = static_cast<float *>(a->ws);
    const size_t ws_floats = a->ws_bytes / sizeof(float);
    ProfInfo pi;
    const double k2 = a->kind == PWS_CONVT_K4S2 ? 4.0 : 9.0;  // taps per forward OUTPUT pixel
    pi.flops = 2.0 * a->n * oh * ow * (double)a->cout * cin_f * k2;
    // algorithmic bytes: dy read once, every destination written once -- plus what the fused epilogue reads: the old gradient of
    // an accumulating destination and the forward tensor whose act' multiplies the sum (bf16 storage: 2 bytes per element)
    {
        const double es = a->store == PWS_STORE_BF16 ? 2.0 : 4.0, in_pix = (double)a->n * a->h * a->w;
        pi.bytes = es * ((double)a->n * oh * ow * a->cout + in_pix * cin_f);
        for (int s = 0; s < a->ndst; ++s)
            pi.bytes += es * in_pix * a->dst[s].channels * (a->dst[s].accumulate ? 1.0 : 0.0) +
                        (kp.dst_act[s] == PWS_ACT_NONE ? 0.0 : (kp.dst_sign[s] ? in_pix * a->dst[s].channels / 8.0 : es * in_pix * a->dst[s].channels));
    }
    if (a->kind == PWS_CONV_K3S2)
        kp.LH = oh, kp.LW = ow, kp.nclasses = 4;
    else
        kp.LH = a->h, kp.LW = a->w;
    if (a->store == PWS_STORE_BF16) {
        PWS_REQUIRE(a->math == PWS_MATH_BF16 && a->w_dgrad_bf16 && a->cout % 32 == 0 && a->gout_ld % 8 == 0,
                    "pws_conv2d_bwd_data: bf16 storage needs bf16 math, w_dgrad_bf16, cout %% 32 == 0 and gout_ld %% 8 == 0");
        for (int s = 0; s < a->ndst; ++s)
            PWS_REQUIRE(a->dst[s].channels % 2 == 0 && a->dst[s].ld % 2 == 0 && (reinterpret_cast<size_t>(a->dst[s].ptr) & 3) == 0,
                        "pws_conv2d_bwd_data: bf16 storage needs even channels / ld and 4-byte aligned destinations");
        kp.io_bf16 = 1;
        kp.epi16 = 1;
        for (int s = 0; s < a->ndst; ++s)
            if (a->dst[s].channels % 8 || a->dst[s].ld % 8 || (reinterpret_cast<size_t>(a->dst[s].ptr) & 15) ||
                (kp.dst_act[s] != PWS_ACT_NONE && (kp.dst_y_ld[s] % 8 || (reinterpret_cast<size_t>(kp.dst_y[s]) & 15))))
                kp.epi16 = 0;
    }
    if (a->math == PWS_MATH_BF16 && a->w_dgrad_bf16) {
        kp.w_bf = a->w_dgrad_bf16, kp.kpad_bf = (a->cout + 31) / 32 * 32, kp.npad_bf = (cin_f + 63) / 64 * 64;
        const int rc = conv_bf16_dgrad(a->kind, kp, a->cout, ws, ws_floats, st, pi);
        if (rc != 1) return rc;
    }
    switch (a->kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        return select_and_launch(kK3S1, 5, kp, a->cout, nullptr, ws, ws_floats, st, pi);
    case PWS_CONV_K3S2:
        return select_and_launch(kSP3, 5, kp, a->cout, nullptr, ws, ws_floats, st, pi);
    case PWS_CONVT_K4S2:
        return select_and_launch(kK4S2, 5, kp, a->cout, nullptr, ws, ws_floats, st, pi);
    default:
        set_error("pws_conv2d_bwd_data: kind %d has no data gradient here", a->kind);
        return PWS_EINVAL;
    }
}

}  // namespace pws

extern "C" int pws_conv2d_fwd(const pws_conv_args *args, pws_stream_t stream) {
    return pws::conv2d_fwd_impl(args, pws::as_stream(stream));
}

extern "C" int pws_conv2d_bwd_data(const pws_conv_bwd_data_args *args, pws_stream_t stream) {
    return pws::conv2d_bwd_data_impl(args, pws::as_stream(stream));
}

// Implicit-GEMM convolution / transposed convolution on the bf16 matrix cores of gfx950
// (v_mfma_f32_32x32x16_bf16: 16 k per instruction, 8 passes -> 16x the rate of the exact-fp32 MFMA), fp32 accumulation.
// "bf16 with MFMA convs" of BASELINE configs 3/4; selected per launch by pws_conv_args.math / PWS_OPT_MATH.
//
// Same GEMM view, tiling, virtual concat, sub-pixel modes, split-K and epilogue as conv_mfma.hip (M = output pixels,
// N = cout, K = taps x cin); what differs is the operand path:
//   * activations stay fp32 in HBM; the halo'd input tile is converted to bf16 (round to nearest even, v_cvt_pk_bf16_f32)
//     on its way into LDS, as [pixel][CK bf16] rows of CK*2 + 16 bytes;
//   * weights come pre-packed as bf16 [class*tap][cout padded to 64][cin padded to 32] (pws_pack_weight_bf16: k contiguous),
//     staged as [tap][cout row][CK bf16] rows of the same pitch;
//   * an MFMA operand is 8 consecutive k per lane = ONE 16-byte LDS read (ds_read_b128) for 16 k-steps of the fp32 kernel's
//     scalar reads; the row pitch (80 B for CK = 32, 48 B for CK = 16) spreads 16 consecutive rows over all 64 banks.
//   * CK = 32 channels per chunk for the stride-1 kinds, 16 for the stride-2 kinds (their halo tile is 4x larger).
// Numerics: each product is exact in fp32 (bf16 x bf16), sums are fp32; the only rounding added to the fp32 path is the
// bf16 rounding of the operands (relative 2^-9 each).  tests/test_hip_bf16.py checks the kernel against the fp32 kernel run on
// pre-rounded operands (tight) and against unrounded fp32 (stated tolerance).
#include "conv_common.h"

namespace pws {

template <int KS_, int STRIDE_, int PAD_, int SUBPIX_, int TH_, int TW_, int TN_, int CK_, int WM_, int WN_, int MT_, int NT_>
struct BfCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, PAD = PAD_, TH = TH_, TW = TW_, TN = TN_, CK = CK_;
    static constexpr int SUBPIX = SUBPIX_;
    static constexpr bool CONVT = SUBPIX_ != 0;
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int BM = TH * TW * TN, BN = 32 * NT * WN;
    static_assert(BM == 32 * MT * WM, "tile pixels must equal the M extent of the wave grid");
    static_assert(CK == 16 || CK == 32, "CK");
    static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    static constexpr int PIX = TN * IH * IW;
    static constexpr int TAPS = KS * KS;
    static constexpr int PITCH = CK * 2 + 16;          // bytes per LDS pixel row
    // Bytes per TILE row of the input image in LDS.  An MFMA A-operand read (ds_read_b128) is served in four groups of 16
    // lanes = {tile-row r: 8 pixels, tile-row r+1: the other 8 columns} for TW = 16 (four rows x 4 pixels for TW = 8); with
    // the 80-byte pixel pitch (5 x 16 B, odd) the 16 addresses fall into 16 different 4-bank slots only if consecutive tile
    // rows start 0 (TW = 16) or 8 (TW = 8) slots apart mod 16.  IW * PITCH alone gave 2-way conflicts on every A read
    // (rocprofv3: SQ_LDS_BANK_CONFLICT 38 % of SQ_LDS_IDX_ACTIVE).  Stride-2 tiles step two pixels per lane: left as is.
    static constexpr int ROWP_MIN = IW * PITCH;
    static constexpr bool ROW_ALIGN = STRIDE == 1 && PITCH == 80 && (TW == 16 || TW == 8);
    static constexpr int ROW_RES = TW == 8 ? 128 : 0;
    static constexpr int ROWP = ROW_ALIGN ? ((ROWP_MIN - ROW_RES + 255) / 256) * 256 + ROW_RES : ROWP_MIN;
    static_assert(ROWP >= ROWP_MIN && ROWP % 16 == 0, "row pitch");
    static constexpr int LDS_IN = TN * IH * ROWP;      // bytes (multiple of 16)
    static constexpr int LDS_W = TAPS * BN * PITCH;
    static constexpr int LDS_BYTES = LDS_IN + LDS_W + 16;  // + a 16-byte sink for the staging items past the tile
    static constexpr int C8 = CK / 8;                  // 8-channel (16-byte bf16) items per row
    static constexpr int N_IN = PIX * C8, N_W = TAPS * BN * C8;
    static constexpr int ITEMS_IN = (N_IN + THREADS - 1) / THREADS;
    static constexpr int ITEMS_W = (N_W + THREADS - 1) / THREADS;
    static_assert(BN == 64, "weights are padded to 64 output channels per block");
};

// bias + activation / gradient scatter (or raw partial sums when K is split) and the NHWC store, as in conv_mfma.hip.
// A macro, not a function: passing the accumulator array by reference made hipcc keep it in scratch (732 B per lane).
#define PWS_BF_EPILOGUE(C, p, acc, wm, wn, l31, hi, n0, y0, x0, co0, py, px, split)                                        \
    do {                                                                                                                   \
        const bool partial_ = (p).ksplit > 1;                                                                              \
        float *part_ = (p).out + (size_t)(split) * (p).split_stride;                                                       \
        _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt) {                                                             \
            const int co = (co0) + ((wn) * C::NT + nt) * 32 + (l31);                                                       \
            const bool co_ok = co < (p).cout;                                                                              \
            _Pragma("unroll") for (int mt = 0; mt < C::MT; ++mt) {                                                         \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                    const int m = ((wm) * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (hi);                            \
                    const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);                          \
                    const int n = (n0) + tn, y = (y0) + ty, x = (x0) + tx;                                                 \
                    const int oy = C::CONVT ? 2 * y + (py) : y, ox = C::CONVT ? 2 * x + (px) : x;                          \
                    if (co_ok && n < (p).N && y < (p).LH && x < (p).LW && oy < (p).OH && ox < (p).OW) {                    \
                        const size_t pix = (size_t)(n * (p).OH + oy) * (p).OW + ox;                                        \
                        if (partial_)                                                                                      \
                            part_[pix * (p).cout + co] = acc[mt][nt][r];                                                   \
                        else                                                                                               \
                            epi_store(p, pix, co, acc[mt][nt][r]);                                                         \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)

// bf16-storage epilogue with 16-byte stores (p.epi16, no K split).  The MFMA result layout gives a lane ONE channel pair of 16
// scattered pixel rows: stored as is (epi_store_pair16) that is 16 * MT dword stores per lane, each with its own address math
// (and a read-modify-write per dword in the accumulating gradient scatter) -- measured at 20-52 % of the kernel's time.  Here
// every wave turns its 32 pixels x 64 channels of one `mt` pass round in LDS (fp32, 8 KB per wave, over the operand tiles
// that are dead by now): a lane writes its pairs with ds_write_b64 and reads back 8 CONSECUTIVE channels of one pixel, so
// that bias / activation / accumulation run on 8 channels at a time and a pixel's 64 channels leave as 8 x 16-byte stores
// (4 per lane and pass instead of 16).  Row layout: sixteen 16-byte slots, slot (h * 8 + g) = channels 8 g + 4 h .. + 3, the
// two halves swapped on odd rows: the 32-lane write groups and the 16-lane read groups (2 rows x 8 lanes) are both bank
// conflict-free.  The arithmetic is that of epi_store_pair16 (fp32 sums, one rounding to bf16).
#define PWS_BF_EPI16(C, p, acc, wv, wm, lane, l31, hi, n0, y0, x0, co0, py, px, lds_raw, LDS_AVAIL)                       \
    do {                                                                                                                   \
        static_assert(C::WM * C::WN * 8192 <= (LDS_AVAIL), "the epilogue tiles reuse the operand tiles' LDS");            \
        __syncthreads(); /* every wave is done with the operand tiles */                                                  \
        unsigned char *et_ = (lds_raw) + (wv) * 8192;                                                                      \
        const int g_ = (lane) & 7;                                                                                         \
        const int co_ = (co0) + g_ * 8;                                                                                    \
        const bool co_ok_ = co_ < (p).cout;                                                                                \
        float b_[8];                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) b_[k] = ((p).bias && co_ok_) ? (p).bias[co_ + k] : 0.f;             \
        __bf16 *dbase_ = reinterpret_cast<__bf16 *>((p).out) + co_;                                                        \
        size_t dld_ = (p).out_ld;                                                                                          \
        bool dacc_ = false, dok_ = co_ok_;                                                                                 \
        const __bf16 *ybase_ = nullptr;                                                                                    \
        size_t yld_ = 0;                                                                                                   \
        float yslope_ = 1.f;                                                                                               \
        if ((p).ndst != 0) {                                                                                               \
            dok_ = false;                                                                                                  \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                             \
                if (s_ < (p).ndst && co_ >= (p).dst_c0[s_] && co_ < (p).dst_c1[s_]) {                                      \
                    dbase_ = reinterpret_cast<__bf16 *>((p).dst_ptr[s_]) + (co_ - (p).dst_c0[s_]);                         \
                    dld_ = (p).dst_ld[s_], dacc_ = (p).dst_acc[s_] != 0, dok_ = true;                                      \
                    if ((p).dst_act[s_] != PWS_ACT_NONE) {                                                                 \
                        ybase_ = static_cast<const __bf16 *>((p).dst_y[s_]) + (co_ - (p).dst_c0[s_]);                      \
                        yld_ = (p).dst_y_ld[s_], yslope_ = (p).dst_act[s_] == PWS_ACT_LRELU ? 0.2f : 0.f;                  \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
        const int wslot_ = (((l31) >> 1) & 1) * 8 + ((l31) >> 2);                                                          \
        _Pragma("unroll") for (int mt = 0; mt < C::MT; ++mt) {                                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (hi);                                                         \
                *reinterpret_cast<float2 *>(et_ + row * 256 + ((wslot_ ^ ((r & 1) * 8)) * 16) + ((l31) & 1) * 8) =         \
                    make_float2(acc[mt][0][r], acc[mt][1][r]);                                                             \
            }                                                                                                              \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                         \
            __builtin_amdgcn_wave_barrier();                                                                               \
            _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                             \
                const int row = it * 8 + ((lane) >> 3);                                                                    \
                const int sw_ = (row & 1) * 8;                                                                             \
                const float4 lo_ = *reinterpret_cast<const float4 *>(et_ + row * 256 + ((g_ ^ sw_) * 16));                 \
                const float4 hi_ = *reinterpret_cast<const float4 *>(et_ + row * 256 + (((8 + g_) ^ sw_) * 16));           \
                const int m = ((wm) * C::MT + mt) * 32 + row;                                                              \
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);                              \
                const int n = (n0) + tn, y = (y0) + ty, x = (x0) + tx;                                                     \
                const int oy = C::CONVT ? 2 * y + (py) : y, ox = C::CONVT ? 2 * x + (px) : x;                              \
                if (dok_ && n < (p).N && y < (p).LH && x < (p).LW && oy < (p).OH && ox < (p).OW) {                         \
                    float v_[8] = {lo_.x, lo_.y, lo_.z, lo_.w, hi_.x, hi_.y, hi_.z, hi_.w};                                \
                    u32x4 *d_ = reinterpret_cast<u32x4 *>(dbase_ + ((size_t)(n * (p).OH + oy) * (p).OW + ox) * dld_);      \
                    if ((p).ndst == 0) {                                                                                   \
                        _Pragma("unroll") for (int k = 0; k < 8; ++k) v_[k] = act_apply(v_[k] + b_[k], (p).act);           \
                    } else if (dacc_) {                                                                                    \
                        const u32x4 o_ = *d_;                                                                              \
                        v_[0] += bf16_lo(o_.x), v_[1] += bf16_hi(o_.x), v_[2] += bf16_lo(o_.y), v_[3] += bf16_hi(o_.y);    \
                        v_[4] += bf16_lo(o_.z), v_[5] += bf16_hi(o_.z), v_[6] += bf16_lo(o_.w), v_[7] += bf16_hi(o_.w);    \
                    }                                                                                                      \
                    if (ybase_) { /* act'(y) of the tensor this destination is the gradient of */                          \
                        const u32x4 y_ = *reinterpret_cast<const u32x4 *>(ybase_ + ((size_t)(n * (p).OH + oy) * (p).OW + ox) * yld_); \
                        v_[0] *= bf16_lo(y_.x) > 0.f ? 1.f : yslope_, v_[1] *= bf16_hi(y_.x) > 0.f ? 1.f : yslope_;        \
                        v_[2] *= bf16_lo(y_.y) > 0.f ? 1.f : yslope_, v_[3] *= bf16_hi(y_.y) > 0.f ? 1.f : yslope_;        \
                        v_[4] *= bf16_lo(y_.z) > 0.f ? 1.f : yslope_, v_[5] *= bf16_hi(y_.z) > 0.f ? 1.f : yslope_;        \
                        v_[6] *= bf16_lo(y_.w) > 0.f ? 1.f : yslope_, v_[7] *= bf16_hi(y_.w) > 0.f ? 1.f : yslope_;        \
                    }                                                                                                      \
                    u32x4 w_;                                                                                              \
                    w_.x = cvt_pk_bf16(v_[0], v_[1]), w_.y = cvt_pk_bf16(v_[2], v_[3]);                                    \
                    w_.z = cvt_pk_bf16(v_[4], v_[5]), w_.w = cvt_pk_bf16(v_[6], v_[7]);                                    \
                    *d_ = w_;                                                                                              \
                    if ((p).ndst == 0 && (p).out_sign) { /* forward: the sign bits of the rounded values (ConvKParams.out_sign) */ \
                        const unsigned m_ = (bf16_lo(w_.x) > 0.f ? 1u : 0u) | (bf16_hi(w_.x) > 0.f ? 2u : 0u) | (bf16_lo(w_.y) > 0.f ? 4u : 0u) | \
                                            (bf16_hi(w_.y) > 0.f ? 8u : 0u) | (bf16_lo(w_.z) > 0.f ? 16u : 0u) | (bf16_hi(w_.z) > 0.f ? 32u : 0u) | \
                                            (bf16_lo(w_.w) > 0.f ? 64u : 0u) | (bf16_hi(w_.w) > 0.f ? 128u : 0u);         \
                        static_cast<unsigned char *>((p).out_sign)[((size_t)(n * (p).OH + oy) * (p).OW + ox) * (size_t)(p).out_sign_ld + (co_ >> 3)] = \
                            (unsigned char)m_;                                                                             \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                         \
            __builtin_amdgcn_wave_barrier();                                                                               \
        }                                                                                                                  \
    } while (0)

// IO16: activations / gradients live in HBM as bf16 (io_bf16): a staging item is ONE 16-byte load of 8 channels copied to
// LDS as is, and the epilogue stores 8 consecutive channels as 16 bytes (PWS_BF_EPI16 above; channel pairs as dwords where a
// destination is not 16-byte aligned and for split-K partials).  For that the 64 output channels of a workgroup are dealt to
// the lanes as (2 l, 2 l + 1) -> (nt 0, nt 1) instead of (l, l + 32): the weight rows are permuted while they are staged
// (LDS row (nn & 1) * 32 + (nn >> 1) holds output channel nn), the MFMA side is unchanged.  Requires NT == 2.
// ABL (interference probe, tools/probes/kernel_victim_probe.py; -DPWS_INTERFERENCE_PROBE builds only: PWS_OPT_EXPERIMENT 2100 + ABL, one tile shape only): while this kernel
// runs, kernels of OTHER streams and processes that share its CUs compute wrong values in a few lanes.  Variants to find out what in it does that:
// 1 no matrix instructions, 2 no LDS operand reads, 4 no LDS stores, 8 no global loads (results meaningless), 16 the gfx90a instruction
// v_mfma_f32_32x32x8_bf16_1k twice in place of v_mfma_f32_32x32x16_bf16 (same results up to the summation order).  Measured: 1 -> clean; 14 (the x16
// matrix instructions and nothing else) -> as bad as the complete kernel; 16 and 30 -> clean, at +29 % of the launch time on the 128 -> 128 @128^2 layer.
// 32: four v_mfma_f32_16x16x32_bf16 (the other gfx950 bf16 shape; results meaningless) per product: as bad as the complete kernel -- but 46 (those and nothing else) clean.
template <class C, bool IO16, int ABL = 0>
__global__ void __launch_bounds__(C::THREADS, 2) conv_bf16_kernel(const ConvKParams p) {
    static_assert(!IO16 || (C::NT == 2 && C::WN == 1), "bf16 storage pairs the two 32-channel blocks of a wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *lds_in = lds_raw;
    unsigned char *lds_w = lds_raw + C::LDS_IN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wv / C::WN, wn = wv % C::WN;

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

    // ---- staging descriptors.  All global loads are unconditional (masked items read a valid dummy address and are
    // zeroed by a select): a conditional load would make hipcc wait vmcnt(0) per load.
    int g_pix[C::ITEMS_IN], l_off[C::ITEMS_IN];
    unsigned ok_mask = 0;
#pragma unroll
    for (int it = 0; it < C::ITEMS_IN; ++it) {
        const int item = tid + it * C::THREADS;
        const int pix = item / C::C8, c8 = item % C::C8;
        const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
        const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
        const bool ok = item < C::N_IN && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        g_pix[it] = ok ? (n * p.H + iy) * p.W + ix : 0;
        ok_mask |= ok ? (1u << it) : 0u;
        l_off[it] = item < C::N_IN ? (tn * C::IH + ly) * C::ROWP + lx * C::PITCH + c8 * 16 : C::LDS_IN + C::LDS_W;  // else: the sink
    }
    static_assert(C::ITEMS_IN <= 32, "mask width");
    // weight items: (tap, cout row, 8-k group); rows co0..co0+63 always exist in the padded bf16 weights
    int w_off[C::ITEMS_W], lw_off[C::ITEMS_W];
#pragma unroll
    for (int it = 0; it < C::ITEMS_W; ++it) {
        const int item = tid + it * C::THREADS;
        const int row = item / C::C8, c8 = item % C::C8;
        const int tap = row / C::BN, nn = row % C::BN;
        w_off[it] = item < C::N_W ? ((tap * p.npad_bf + co0 + nn) * p.kpad_bf + c8 * 8) : 0;
        const int lrow = IO16 ? tap * C::BN + (nn & 1) * 32 + (nn >> 1) : row;
        lw_off[it] = item < C::N_W ? C::LDS_IN + lrow * C::PITCH + c8 * 16 : C::LDS_IN + C::LDS_W;
    }
    const __bf16 *w_cls = static_cast<const __bf16 *>(p.w_bf) + (size_t)cls * C::TAPS * p.npad_bf * p.kpad_bf;
    f32x4 r_in[IO16 ? 1 : C::ITEMS_IN][2];
    u32x4 r_in16[IO16 ? C::ITEMS_IN : 1];
    u32x4 r_w[C::ITEMS_W];

    auto load_chunk = [&](int s, int c0, int wrow) {
        const size_t ld = p.src_ld[s];
        if constexpr (IO16) {
            const __bf16 *sp = reinterpret_cast<const __bf16 *>(p.src_ptr[s]) + c0;
#pragma unroll
            for (int it = 0; it < C::ITEMS_IN; ++it)
                r_in16[it] = *reinterpret_cast<const u32x4 *>(sp + (size_t)g_pix[it] * ld + ((tid + it * C::THREADS) % C::C8) * 8);
        } else {
            const float *sp = p.src_ptr[s] + c0;
#pragma unroll
            for (int it = 0; it < C::ITEMS_IN; ++it) {
                const float *g = sp + (size_t)g_pix[it] * ld + ((tid + it * C::THREADS) % C::C8) * 8;
                r_in[it][0] = *reinterpret_cast<const f32x4 *>(g);
                r_in[it][1] = *reinterpret_cast<const f32x4 *>(g + 4);
            }
        }
        const __bf16 *wp = w_cls + wrow;
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) r_w[it] = *reinterpret_cast<const u32x4 *>(wp + w_off[it]);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < C::ITEMS_IN; ++it) {
            const bool ok = (ok_mask >> it) & 1u;
            u32x4 v;
            if constexpr (IO16) {
                v.x = ok ? r_in16[it].x : 0u, v.y = ok ? r_in16[it].y : 0u, v.z = ok ? r_in16[it].z : 0u, v.w = ok ? r_in16[it].w : 0u;
            } else {
                v.x = ok ? cvt_pk_bf16(r_in[it][0].x, r_in[it][0].y) : 0u, v.y = ok ? cvt_pk_bf16(r_in[it][0].z, r_in[it][0].w) : 0u;
                v.z = ok ? cvt_pk_bf16(r_in[it][1].x, r_in[it][1].y) : 0u, v.w = ok ? cvt_pk_bf16(r_in[it][1].z, r_in[it][1].w) : 0u;
            }
            *reinterpret_cast<u32x4 *>(lds_raw + l_off[it]) = v;
        }
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) *reinterpret_cast<u32x4 *>(lds_raw + lw_off[it]) = r_w[it];
    };

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // operand bases (bytes): lane = (row l31, k half hi) reads 16 bytes = k 8*hi .. 8*hi+7 of the current 16-k step
    int a_base[C::MT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int m = (wm * C::MT + mt) * 32 + l31;
        const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
        a_base[mt] = (tn * C::IH + ty * C::STRIDE) * C::ROWP + tx * C::STRIDE * C::PITCH + hi * 16;
    }
    const int b_base = (wn * C::NT * 32 + l31) * C::PITCH + hi * 16;

    int total_chunks = 0;
    for (int s = 0; s < p.nsrc; ++s) total_chunks += p.src_c[s] / C::CK;
    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(total_chunks, ch_begin + p.chunks_per_split);
    int s = 0, c0 = ch_begin * C::CK, wrow = ch_begin * C::CK;
    while (s < p.nsrc - 1 && c0 >= p.src_c[s]) c0 -= p.src_c[s], ++s;
    if constexpr (ABL & 8) {
#pragma unroll
        for (int it = 0; it < C::ITEMS_W; ++it) r_w[it] = u32x4{1u, 2u, 3u, 4u};
#pragma unroll
        for (int it = 0; it < (IO16 ? C::ITEMS_IN : 1); ++it) r_in16[it] = u32x4{5u, 6u, 7u, 8u};
    }
    if (ch_begin < ch_end && !(ABL & 8)) load_chunk(s, c0, wrow);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();  // everyone finished reading the previous chunk from LDS
        if (!(ABL & 4)) store_chunk();
        __syncthreads();
        c0 += C::CK, wrow += C::CK;
        if (c0 >= p.src_c[s]) ++s, c0 = 0;
        if (ch + 1 < ch_end && !(ABL & 8)) load_chunk(s, c0, wrow);

#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            if (C::SUBPIX == 2 && ((!py && tap / C::KS) || (!px && tap % C::KS))) continue;  // block-uniform
            const int toff = (tap / C::KS) * C::ROWP + (tap % C::KS) * C::PITCH;
#pragma unroll
            for (int ks = 0; ks < C::CK / 16; ++ks) {
                bf16x8 a[C::MT], b[C::NT];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    if (ABL & 2) a[mt] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)(tap + ch), 1u, 2u, (unsigned)lane});
                    else a[mt] = *reinterpret_cast<const bf16x8 *>(lds_in + a_base[mt] + toff + ks * 32);
                }
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt) {
                    if (ABL & 2) b[nt] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)(tap + ch), 3u, 4u, (unsigned)lane});
                    else b[nt] = *reinterpret_cast<const bf16x8 *>(lds_w + b_base + (tap * C::BN + nt * 32) * C::PITCH + ks * 32);
                }
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt) {
                        if (ABL & 32) {   // PROBE ONLY (results meaningless): the other gfx950 bf16 shape, v_mfma_f32_16x16x32_bf16, four per 32 x 32 x 16 product
                            typedef float f32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) {
                                f32x4_ c4 = {acc[mt][nt][4 * q4], acc[mt][nt][4 * q4 + 1], acc[mt][nt][4 * q4 + 2], acc[mt][nt][4 * q4 + 3]};
                                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], c4, 0, 0, 0);
                                acc[mt][nt][4 * q4] = c4[0], acc[mt][nt][4 * q4 + 1] = c4[1], acc[mt][nt][4 * q4 + 2] = c4[2], acc[mt][nt][4 * q4 + 3] = c4[3];
                            }
                        } else if (ABL & 1) acc[mt][nt][0] += (float)a[mt][0] * (float)b[nt][1];
                        else if (ABL & 16) {   // the gfx90a instruction, twice: k {0..3, 8..11} then {4..7, 12..15} (a lane's 8 values = its k half)
                            typedef short s16x4_ __attribute__((ext_vector_type(4)));
                            typedef short s16x8_ __attribute__((ext_vector_type(8)));
                            const s16x8_ av = __builtin_bit_cast(s16x8_, a[mt]), bv = __builtin_bit_cast(s16x8_, b[nt]);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_shufflevector(av, av, 0, 1, 2, 3), __builtin_shufflevector(bv, bv, 0, 1, 2, 3), acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_shufflevector(av, av, 4, 5, 6, 7), __builtin_shufflevector(bv, bv, 4, 5, 6, 7), acc[mt][nt], 0, 0, 0);
                        } else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
        }
    }

    if constexpr (IO16) {
        // channel pair (2 l31, 2 l31 + 1) of this wave's 64 channels: one dword per pixel and lane, 128 B per pixel and half-wave
        const bool partial = p.ksplit > 1;
        if (p.epi16 && !partial) {   // block-uniform
            PWS_BF_EPI16(C, p, acc, wv, wm, lane, l31, hi, n0, y0, x0, co0, py, px, lds_raw, C::LDS_BYTES);
            return;
        }
        float *part = p.out + (size_t)split * p.split_stride;
        const int co = co0 + 2 * l31;
        const bool co_ok = co < p.cout;   // cout is even
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
                        *reinterpret_cast<float2 *>(part + pix * p.cout + co) = make_float2(acc[mt][0][r], acc[mt][1][r]);
                    else
                        epi_store_pair16(p, pix, co, acc[mt][0][r], acc[mt][1][r]);
                }
            }
        }
    } else {
        PWS_BF_EPILOGUE(C, p, acc, wm, wn, l31, hi, n0, y0, x0, co0, py, px, split);
    }
}

// First layer (Conv2d k5 s1 p2, 31 -> 64 channels): 25 taps x 64 output rows of weights do not fit LDS beside the input tile
// at two workgroups per CU, so the taps are walked one kernel ROW at a time: the halo'd input tile of a 32-channel chunk is
// staged once, then for each of the 5 tap rows the 5 x 64 weight rows (25.6 KB) are staged and multiplied (the next row's
// weights are prefetched into registers meanwhile).  The source is the NHWC copy of the NCHW window padded to 32 channels
// (pws_nchw_to_nhwc_pad).  C = BfCfg<5,1,2,0,16,16,1,32,...>; C::LDS_W is not used (see K5_LDS_BYTES).
template <class C>
struct K5Lds {
    static constexpr int ROWW = C::KS * C::BN * C::PITCH;  // one tap row of weights
    static constexpr int BYTES = C::LDS_IN + ROWW + 16;
    static constexpr int N_WR = C::KS * C::BN * C::C8, ITEMS_WR = (N_WR + C::THREADS - 1) / C::THREADS;
};

template <class C, bool IO16>
__global__ void __launch_bounds__(C::THREADS, 2) conv_bf16_k5_kernel(const ConvKParams p) {
    using K = K5Lds<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *lds_in = lds_raw;
    unsigned char *lds_w = lds_raw + C::LDS_IN;
    constexpr int SINK = C::LDS_IN + K::ROWW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wv / C::WN, wn = wv % C::WN;
    const unsigned tile = xcd_remap(blockIdx.x, p.ntiles);
    const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, tn_i = tile / (p.tiles_x * p.tiles_y);
    const int n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
    const int co0 = blockIdx.y * C::BN;
    const int iy0 = y0 - C::PAD, ix0 = x0 - C::PAD;

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
        a_base[mt] = (tn * C::IH + ty) * C::ROWP + tx * C::PITCH + hi * 16;
    }
    const int b_base = (wn * C::NT * 32 + l31) * C::PITCH + hi * 16;

    // weight-row items: (tap in row, cout row, 8-k group)
    int w_off[K::ITEMS_WR], lw_off[K::ITEMS_WR];
#pragma unroll
    for (int it = 0; it < K::ITEMS_WR; ++it) {
        const int item = tid + it * C::THREADS;
        const int row = item / C::C8, c8 = item % C::C8;
        const int tap = row / C::BN, nn = row % C::BN;
        w_off[it] = item < K::N_WR ? ((tap * p.npad_bf + co0 + nn) * p.kpad_bf + c8 * 8) : 0;
        const int lrow = IO16 ? tap * C::BN + (nn & 1) * 32 + (nn >> 1) : row;  // channel pairs per lane, see conv_bf16_kernel
        lw_off[it] = item < K::N_WR ? C::LDS_IN + lrow * C::PITCH + c8 * 16 : SINK;
    }
    const __bf16 *wbase = static_cast<const __bf16 *>(p.w_bf);
    u32x4 r_w[K::ITEMS_WR];
    auto load_wrow = [&](int r, int wrow) {
        const __bf16 *wp = wbase + (size_t)r * C::KS * p.npad_bf * p.kpad_bf + wrow;
#pragma unroll
        for (int it = 0; it < K::ITEMS_WR; ++it) r_w[it] = *reinterpret_cast<const u32x4 *>(wp + w_off[it]);
    };

    const int nchunks = p.src_c[0] / C::CK;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        // ---- input tile of this channel chunk (batches of 4 items: loads first, then convert + store)
        {
            constexpr int ITS = C::ITEMS_IN, BATCH = 4;
            const float *sp = p.src_ptr[0] + (IO16 ? 0 : ch * C::CK);
            const __bf16 *sp16 = reinterpret_cast<const __bf16 *>(p.src_ptr[0]) + ch * C::CK;
            const size_t ld = p.src_ld[0];
#pragma unroll 1
            for (int it0 = 0; it0 < ITS; it0 += BATCH) {
                f32x4 r[BATCH][2];
                u32x4 r16[BATCH];
                int off[BATCH];
                bool okv[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int item = tid + (it0 + k) * C::THREADS;
                    const int pix = item / C::C8, c8 = item % C::C8;
                    const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                    const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                    const bool ok = item < C::N_IN && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    const size_t e = ok ? ((size_t)(n * p.H + iy) * p.W + ix) * ld + c8 * 8 : 0;
                    if constexpr (IO16) {
                        r16[k] = *reinterpret_cast<const u32x4 *>(sp16 + e);
                    } else {
                        r[k][0] = *reinterpret_cast<const f32x4 *>(sp + e);
                        r[k][1] = *reinterpret_cast<const f32x4 *>(sp + e + 4);
                    }
                    okv[k] = ok;
                    off[k] = item < C::N_IN ? (tn * C::IH + ly) * C::ROWP + lx * C::PITCH + c8 * 16 : SINK;
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    u32x4 v;
                    if constexpr (IO16) {
                        v.x = okv[k] ? r16[k].x : 0u, v.y = okv[k] ? r16[k].y : 0u, v.z = okv[k] ? r16[k].z : 0u, v.w = okv[k] ? r16[k].w : 0u;
                    } else {
                        v.x = okv[k] ? cvt_pk_bf16(r[k][0].x, r[k][0].y) : 0u, v.y = okv[k] ? cvt_pk_bf16(r[k][0].z, r[k][0].w) : 0u;
                        v.z = okv[k] ? cvt_pk_bf16(r[k][1].x, r[k][1].y) : 0u, v.w = okv[k] ? cvt_pk_bf16(r[k][1].z, r[k][1].w) : 0u;
                    }
                    *reinterpret_cast<u32x4 *>(lds_raw + off[k]) = v;
                }
            }
        }
        load_wrow(0, ch * C::CK);
#pragma unroll 1
        for (int r = 0; r < C::KS; ++r) {
            __syncthreads();  // input tile visible (r == 0) / previous row's weights consumed
#pragma unroll
            for (int it = 0; it < K::ITEMS_WR; ++it) *reinterpret_cast<u32x4 *>(lds_raw + lw_off[it]) = r_w[it];
            __syncthreads();
            if (r + 1 < C::KS) load_wrow(r + 1, ch * C::CK);
#pragma unroll
            for (int t = 0; t < C::KS; ++t) {
                const int toff = r * C::ROWP + t * C::PITCH;
#pragma unroll
                for (int ks = 0; ks < C::CK / 16; ++ks) {
                    bf16x8 a[C::MT], b[C::NT];
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8 *>(lds_in + a_base[mt] + toff + ks * 32);
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
                        b[nt] = *reinterpret_cast<const bf16x8 *>(lds_w + b_base + (t * C::BN + nt * 32) * C::PITCH + ks * 32);
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < C::NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (IO16) {
        if (p.epi16) {   // block-uniform
            PWS_BF_EPI16(C, p, acc, wv, wm, lane, l31, hi, n0, y0, x0, co0, 0, 0, lds_raw, K::BYTES);
            return;
        }
        const int co = co0 + 2 * l31;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                if (co < p.cout && n < p.N && y < p.LH && x < p.LW)
                    epi_store_pair16(p, (size_t)(n * p.OH + y) * p.OW + x, co, acc[mt][0][r], acc[mt][1][r]);
            }
        }
    } else {
        PWS_BF_EPILOGUE(C, p, acc, wm, wn, l31, hi, n0, y0, x0, co0, 0, 0, 0);
    }
}

// NCHW fp32 [n][c][hw] -> NHWC fp32 [n][hw][cpad] with zero padding channels (cpad <= 32, a multiple of 4): 64 pixels per
// workgroup through LDS so that both sides are coalesced.
template <bool IO16>
__global__ void __launch_bounds__(256) nchw_to_nhwc_pad_kernel(const float *__restrict__ x, float *__restrict__ out, int c, int cpad,
                                                              size_t hw, size_t sstride) {
    __shared__ float t[32][65];
    const size_t p0 = (size_t)blockIdx.x * 64;
    const int n = blockIdx.y;
    const float *xn = x + (size_t)n * sstride;   // sstride: floats between samples (c * hw when dense; hw for sliding windows)
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int ch = i >> 6, px = i & 63;
        t[ch][px] = (ch < c && p0 + px < hw) ? xn[(size_t)ch * hw + p0 + px] : 0.f;
    }
    __syncthreads();
    if constexpr (IO16) {
        unsigned *on = reinterpret_cast<unsigned *>(reinterpret_cast<__bf16 *>(out) + ((size_t)n * hw + p0) * cpad);
        for (int i = threadIdx.x; i < 64 * cpad / 2; i += 256) {
            const int px = i / (cpad / 2), ch = (i % (cpad / 2)) * 2;
            if (p0 + px < hw) on[i] = cvt_pk_bf16(t[ch][px], t[ch + 1][px]);
        }
    } else {
        float *on = out + ((size_t)n * hw + p0) * cpad;
        for (int i = threadIdx.x; i < 64 * cpad; i += 256) {
            const int px = i / cpad, ch = i % cpad;
            if (p0 + px < hw) on[i] = t[ch][px];
        }
    }
}

// fp32 <-> bf16 copies of small tensors (the theta head's 2x2x4ngf input and its gradient under bf16 storage)
__global__ void __launch_bounds__(256) cvt_bf16_to_f32_kernel(const unsigned short *__restrict__ src, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = __builtin_bit_cast(float, (unsigned)src[i] << 16);
}
__global__ void __launch_bounds__(256) cvt_f32_to_bf16_kernel(const float *__restrict__ src, unsigned *__restrict__ dst, size_t npairs,
                                                             int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    float a = src[2 * i], b = src[2 * i + 1];
    if (accumulate) {
        const unsigned o = dst[i];
        a += bf16_lo(o), b += bf16_hi(o);
    }
    dst[i] = cvt_pk_bf16(a, b);
}

// ------------------------------------------------------------------------------------------------ host side
template <class C, bool IO16>
static int launch_bf_io(ConvKParams &kp, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_kernel<C, IO16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_bf16_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    const dim3 grid = conv_grid(kp, C::BN);
#ifdef PWS_INTERFERENCE_PROBE   // diagnostic build only (round 5: the six probe instantiations are no longer part of the shipped library; -DPWS_INTERFERENCE_PROBE brings them back for tools/probes/kernel_victim_probe.py)
    if constexpr (IO16 && C::KS == 3 && C::TH == 16 && C::TW == 16 && C::STRIDE == 1 && C::SUBPIX == 0) {   // the probe variants of this one tile (see conv_bf16_kernel)
        switch (g_experiment) {
#define PWS_BF_ABL_CASE(k) case 2100 + k: (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_kernel<C, IO16, k>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES); hipLaunchKernelGGL((conv_bf16_kernel<C, IO16, k>), grid, dim3(C::THREADS), C::LDS_BYTES, st, kp); return check_launch("conv_bf16_kernel");
            PWS_BF_ABL_CASE(1) PWS_BF_ABL_CASE(14) PWS_BF_ABL_CASE(16) PWS_BF_ABL_CASE(30) PWS_BF_ABL_CASE(32) PWS_BF_ABL_CASE(46)
        default: break;
        }
    }
#endif
    hipLaunchKernelGGL((conv_bf16_kernel<C, IO16>), grid, dim3(C::THREADS), C::LDS_BYTES, st, kp);
    return check_launch("conv_bf16_kernel");
}

// C: configuration for fp32 storage; CS: same tile with 64 channels per wave, used for bf16 storage (may be C itself)
template <class C, class CS>
static int launch_bf(ConvKParams &kp, hipStream_t st, const ProfInfo &) {
    static_assert(C::TH == CS::TH && C::TW == CS::TW && C::TN == CS::TN && C::CK == CS::CK, "same tile");
    if (kp.io_bf16) return launch_bf_io<CS, true>(kp, st);
    return launch_bf_io<C, false>(kp, st);
}

template <class C, class CS = C>
static constexpr TileChoice bchoice() {
    return TileChoice{C::TH, C::TW, C::TN, C::CK, C::BN, KID_CONV_BF16, &launch_bf<C, CS>};
}

// (*_S: the 64-pixel tiles with 2 waves x 64 channels instead of 4 waves x 32: what the bf16-storage epilogue needs)
//                     KS S  P  subpix TH  TW  TN  CK WM WN MT NT
using B_K3S1_T256 = BfCfg<3, 1, 1, 0, 16, 16, 1, 32, 4, 1, 2, 2>;
using B_K3S1_T128 = BfCfg<3, 1, 1, 0, 8, 16, 1, 32, 4, 1, 1, 2>;
using B_K3S1_T64 = BfCfg<3, 1, 1, 0, 8, 8, 1, 32, 2, 2, 1, 1>;
using B_K3S1_T64_S = BfCfg<3, 1, 1, 0, 8, 8, 1, 32, 2, 1, 1, 2>;
using B_K3S1_T64N4 = BfCfg<3, 1, 1, 0, 4, 4, 4, 32, 2, 2, 1, 1>;
using B_K3S1_T64N4_S = BfCfg<3, 1, 1, 0, 4, 4, 4, 32, 2, 1, 1, 2>;
using B_K3S1_T64N16 = BfCfg<3, 1, 1, 0, 2, 2, 16, 32, 2, 2, 1, 1>;
using B_K3S1_T64N16_S = BfCfg<3, 1, 1, 0, 2, 2, 16, 32, 2, 1, 1, 2>;
using B_K3S2_T256 = BfCfg<3, 2, 1, 0, 16, 16, 1, 16, 4, 1, 2, 2>;
using B_K3S2_T128 = BfCfg<3, 2, 1, 0, 8, 16, 1, 16, 4, 1, 1, 2>;
using B_K3S2_T64 = BfCfg<3, 2, 1, 0, 8, 8, 1, 16, 2, 2, 1, 1>;
using B_K3S2_T64_S = BfCfg<3, 2, 1, 0, 8, 8, 1, 16, 2, 1, 1, 2>;
using B_K3S2_T64N4 = BfCfg<3, 2, 1, 0, 4, 4, 4, 16, 2, 2, 1, 1>;
using B_K3S2_T64N4_S = BfCfg<3, 2, 1, 0, 4, 4, 4, 16, 2, 1, 1, 2>;
using B_K3S2_T64N16 = BfCfg<3, 2, 1, 0, 2, 2, 16, 16, 2, 2, 1, 1>;
using B_K3S2_T64N16_S = BfCfg<3, 2, 1, 0, 2, 2, 16, 16, 2, 1, 1, 2>;
using B_CT4_T256 = BfCfg<2, 1, 0, 1, 16, 16, 1, 32, 4, 1, 2, 2>;
using B_CT4_T128 = BfCfg<2, 1, 0, 1, 8, 16, 1, 32, 4, 1, 1, 2>;
using B_CT4_T64 = BfCfg<2, 1, 0, 1, 8, 8, 1, 32, 2, 2, 1, 1>;
using B_CT4_T64_S = BfCfg<2, 1, 0, 1, 8, 8, 1, 32, 2, 1, 1, 2>;
using B_CT4_T64N4 = BfCfg<2, 1, 0, 1, 4, 4, 4, 32, 2, 2, 1, 1>;
using B_CT4_T64N4_S = BfCfg<2, 1, 0, 1, 4, 4, 4, 32, 2, 1, 1, 2>;
using B_CT4_T64N16 = BfCfg<2, 1, 0, 1, 2, 2, 16, 32, 2, 2, 1, 1>;
using B_CT4_T64N16_S = BfCfg<2, 1, 0, 1, 2, 2, 16, 32, 2, 1, 1, 2>;
// data-gradient kinds: conv k4 s2 p1 over dy (gradient of ConvTranspose2d k4 s2 p1), sub-pixel gradient of conv k3 s2 p1
using B_K4S2_T128 = BfCfg<4, 2, 1, 0, 8, 16, 1, 16, 4, 1, 1, 2>;
using B_K4S2_T64 = BfCfg<4, 2, 1, 0, 8, 8, 1, 16, 2, 2, 1, 1>;
using B_K4S2_T64_S = BfCfg<4, 2, 1, 0, 8, 8, 1, 16, 2, 1, 1, 2>;
using B_K4S2_T64N4 = BfCfg<4, 2, 1, 0, 4, 4, 4, 16, 2, 2, 1, 1>;
using B_K4S2_T64N4_S = BfCfg<4, 2, 1, 0, 4, 4, 4, 16, 2, 1, 1, 2>;
using B_K4S2_T64N16 = BfCfg<4, 2, 1, 0, 2, 2, 16, 16, 2, 2, 1, 1>;
using B_K4S2_T64N16_S = BfCfg<4, 2, 1, 0, 2, 2, 16, 16, 2, 1, 1, 2>;
using B_SP3_T256 = BfCfg<2, 1, 0, 2, 16, 16, 1, 32, 4, 1, 2, 2>;
using B_SP3_T128 = BfCfg<2, 1, 0, 2, 8, 16, 1, 32, 4, 1, 1, 2>;
using B_SP3_T64 = BfCfg<2, 1, 0, 2, 8, 8, 1, 32, 2, 2, 1, 1>;
using B_SP3_T64_S = BfCfg<2, 1, 0, 2, 8, 8, 1, 32, 2, 1, 1, 2>;
using B_SP3_T64N4 = BfCfg<2, 1, 0, 2, 4, 4, 4, 32, 2, 2, 1, 1>;
using B_SP3_T64N4_S = BfCfg<2, 1, 0, 2, 4, 4, 4, 32, 2, 1, 1, 2>;
using B_SP3_T64N16 = BfCfg<2, 1, 0, 2, 2, 2, 16, 32, 2, 2, 1, 1>;
using B_SP3_T64N16_S = BfCfg<2, 1, 0, 2, 2, 2, 16, 32, 2, 1, 1, 2>;

static const TileChoice kBK3S1[] = {bchoice<B_K3S1_T256>(), bchoice<B_K3S1_T128>(), bchoice<B_K3S1_T64, B_K3S1_T64_S>(), bchoice<B_K3S1_T64N4, B_K3S1_T64N4_S>(),
                                    bchoice<B_K3S1_T64N16, B_K3S1_T64N16_S>()};
static const TileChoice kBK3S2[] = {bchoice<B_K3S2_T256>(), bchoice<B_K3S2_T128>(), bchoice<B_K3S2_T64, B_K3S2_T64_S>(), bchoice<B_K3S2_T64N4, B_K3S2_T64N4_S>(),
                                    bchoice<B_K3S2_T64N16, B_K3S2_T64N16_S>()};
static const TileChoice kBCT4[] = {bchoice<B_CT4_T256>(), bchoice<B_CT4_T128>(), bchoice<B_CT4_T64, B_CT4_T64_S>(), bchoice<B_CT4_T64N4, B_CT4_T64N4_S>(),
                                   bchoice<B_CT4_T64N16, B_CT4_T64N16_S>()};
static const TileChoice kBK4S2[] = {bchoice<B_K4S2_T128>(), bchoice<B_K4S2_T64, B_K4S2_T64_S>(), bchoice<B_K4S2_T64N4, B_K4S2_T64N4_S>(), bchoice<B_K4S2_T64N16, B_K4S2_T64N16_S>()};
static const TileChoice kBSP3[] = {bchoice<B_SP3_T256>(), bchoice<B_SP3_T128>(), bchoice<B_SP3_T64, B_SP3_T64_S>(), bchoice<B_SP3_T64N4, B_SP3_T64N4_S>(),
                                   bchoice<B_SP3_T64N16, B_SP3_T64N16_S>()};

// 256-pixel tiles for the DEEP maps of a training batch (8 x 8 x 4 samples, 4 x 4 x 16, 2 x 2 x 64): at batch 64 these levels hold
// 256 .. 4096 pixels, and the 64-pixel tiles above re-read a layer's weights once per tile (512 -> 512 @4x4 x 64: 16 tiles x 9.4 MB
// from L2 per launch, 26 us = 183 TFLOP/s); a 256-pixel tile quarters that and has the 2 x 2 fragments per wave of the large-map
// tiles.  Stride-1 kinds only (the stride-2 kinds' input tiles of 256 output pixels exceed half the LDS).
using B_K3S1_D8 = BfCfg<3, 1, 1, 0, 8, 8, 4, 16, 4, 1, 2, 2>;
using B_K3S1_D4 = BfCfg<3, 1, 1, 0, 4, 4, 16, 16, 4, 1, 2, 2>;
using B_K3S1_D2 = BfCfg<3, 1, 1, 0, 2, 2, 64, 16, 4, 1, 2, 2>;
using B_CT4_D8 = BfCfg<2, 1, 0, 1, 8, 8, 4, 32, 4, 1, 2, 2>;
using B_CT4_D4 = BfCfg<2, 1, 0, 1, 4, 4, 16, 32, 4, 1, 2, 2>;
using B_CT4_D2 = BfCfg<2, 1, 0, 1, 2, 2, 64, 32, 4, 1, 2, 2>;
using B_SP3_D8 = BfCfg<2, 1, 0, 2, 8, 8, 4, 32, 4, 1, 2, 2>;
using B_SP3_D4 = BfCfg<2, 1, 0, 2, 4, 4, 16, 32, 4, 1, 2, 2>;
using B_SP3_D2 = BfCfg<2, 1, 0, 2, 2, 2, 64, 32, 4, 1, 2, 2>;
static const TileChoice kBK3S1Deep[] = {bchoice<B_K3S1_D8>(), bchoice<B_K3S1_D4>(), bchoice<B_K3S1_D2>()};
static const TileChoice kBCT4Deep[] = {bchoice<B_CT4_D8>(), bchoice<B_CT4_D4>(), bchoice<B_CT4_D2>()};
static const TileChoice kBSP3Deep[] = {bchoice<B_SP3_D8>(), bchoice<B_SP3_D4>(), bchoice<B_SP3_D2>()};

// the deep tile of this launch, or nullptr: the map is exactly 8 x 8 / 4 x 4 / 2 x 2, the batch fills the tile's samples, and K is
// long enough for the K split to give every CU work (PWS_OPT_EXPERIMENT 95: never, 96: whenever the map fits)
static const TileChoice *deep_tile(const TileChoice *deep, const ConvKParams &kp, int cin_total) {
    if (g_experiment == 95) return nullptr;
    const int idx = (kp.LH == 8 && kp.LW == 8) ? 0 : ((kp.LH == 4 && kp.LW == 4) ? 1 : ((kp.LH == 2 && kp.LW == 2) ? 2 : -1));
    if (idx < 0) return nullptr;
    const TileChoice &c = deep[idx];
    if (g_experiment == 96) return &c;
    // measured at batch 64 (tools/conv_bench.py, bf16 storage): 8 x 8: 512 -> 512 3x3 51.5 -> 43.7 us; 4 x 4: transposed 1024 -> 512 47.7 ->
    // 43.9 us but 3x3 512 -> 512 26.4 -> 27.2 us; 2 x 2: 16.7 -> 21.0 us -- the deep launches are bound by their chunk-serial K loop and
    // the K-split reduce, not by weight re-reads: only the two winning cases are taken
    if (idx == 2 || (idx == 1 && kp.nclasses != 4)) return nullptr;
    if (kp.N < c.tn) return nullptr;
    const long blocks = cdiv(kp.N, c.tn) * cdiv(kp.cout, 64) * kp.nclasses;
    const long chunks = cin_total / c.ck;
    if (blocks * (chunks / 2) < 128) return nullptr;
    return &c;
}

using B_K5S1_T256 = BfCfg<5, 1, 2, 0, 16, 16, 1, 32, 4, 1, 2, 2>;

static int launch_k5(ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    using C = B_K5S1_T256;
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_k5_kernel<C, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, K5Lds<C>::BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_k5_kernel<C, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, K5Lds<C>::BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_bf16_k5_kernel): %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    kp.tiles_x = (kp.LW + C::TW - 1) / C::TW, kp.tiles_y = (kp.LH + C::TH - 1) / C::TH;
    kp.ntiles = (unsigned)(kp.tiles_x * kp.tiles_y * kp.N);
    kp.ksplit = 1, kp.chunks_per_split = 0, kp.split_stride = 0;
    ProfScope prof(KID_CONV_BF16, pi.flops, pi.bytes, st);
    const dim3 grid(kp.ntiles, (kp.cout + C::BN - 1) / C::BN);
    if (kp.io_bf16)
        hipLaunchKernelGGL((conv_bf16_k5_kernel<C, true>), grid, dim3(C::THREADS), K5Lds<C>::BYTES, st, kp);
    else
        hipLaunchKernelGGL((conv_bf16_k5_kernel<C, false>), grid, dim3(C::THREADS), K5Lds<C>::BYTES, st, kp);
    return check_launch("conv_bf16_k5_kernel");
}

// Forward kinds.  kp is fully prepared by conv2d_fwd_impl (conv_mfma.hip); returns 1 when this launch is not covered by
// the bf16 kernels (the caller then runs the fp32 path), else the launch status.
int conv_ring_try(int kind, bool dgrad, const ConvKParams &kp, hipStream_t st, const ProfInfo &pi);   // conv_ring.hip; 1 = not covered
int conv_skinny16_try(int kind, bool dgrad, ConvKParams &kp, int kchan, float *final_out, float *ws, size_t ws_floats, hipStream_t st,
                      const ProfInfo &pi);   // conv_skinny16.hip (the deep levels, bf16 storage); 1 = not covered

static int conv_bf16_fwd_rest(int kind, ConvKParams &kp, int cin_total, float *out, float *ws, size_t ws_floats, hipStream_t st,
                              const ProfInfo &pi, bool *tiled) {
    if (kind != PWS_CONV_K5S1) {
        const int rc = conv_skinny16_try(kind, false, kp, cin_total, out, ws, ws_floats, st, pi);
        if (rc != 1) return rc;
    }
    *tiled = true;
    switch (kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        if (const TileChoice *d = deep_tile(kBK3S1Deep, kp, cin_total)) return select_and_launch(d, 1, kp, cin_total, out, ws, ws_floats, st, pi, kFillBlocksBf16);
        return select_and_launch(kBK3S1, 5, kp, cin_total, out, ws, ws_floats, st, pi, kFillBlocksBf16);
    case PWS_CONV_K3S2: return select_and_launch(kBK3S2, 5, kp, cin_total, out, ws, ws_floats, st, pi, kFillBlocksBf16);
    case PWS_CONVT_K4S2:
        if (const TileChoice *d = deep_tile(kBCT4Deep, kp, cin_total)) return select_and_launch(d, 1, kp, cin_total, out, ws, ws_floats, st, pi, kFillBlocksBf16);
        return select_and_launch(kBCT4, 5, kp, cin_total, out, ws, ws_floats, st, pi, kFillBlocksBf16);
    case PWS_CONV_K5S1:
        if (kp.nsrc != 1) return 1;
        kp.out = out;
        return launch_k5(kp, st, pi);
    default: return 1;
    }
}

// Sign bits of a bf16 NHWC tensor (ConvKParams.out_sign) for the forward kernels that do not write them themselves: one lane per
// (pixel, 8 channels): a 16-byte load, a byte store.
__global__ void __launch_bounds__(256) sign_bits_kernel(const __bf16 *__restrict__ y, int ld, size_t pixels, int groups, unsigned char *__restrict__ sign,
                                                        int sign_ld) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * (size_t)groups) return;
    const size_t pix = i / (size_t)groups;
    const int g = (int)(i - pix * (size_t)groups);
    const u32x4 v = *reinterpret_cast<const u32x4 *>(y + pix * (size_t)ld + g * 8);
    const unsigned m = (bf16_lo(v.x) > 0.f ? 1u : 0u) | (bf16_hi(v.x) > 0.f ? 2u : 0u) | (bf16_lo(v.y) > 0.f ? 4u : 0u) | (bf16_hi(v.y) > 0.f ? 8u : 0u) |
                       (bf16_lo(v.z) > 0.f ? 16u : 0u) | (bf16_hi(v.z) > 0.f ? 32u : 0u) | (bf16_lo(v.w) > 0.f ? 64u : 0u) | (bf16_hi(v.w) > 0.f ? 128u : 0u);
    sign[pix * (size_t)sign_ld + g] = (unsigned char)m;
}

int conv_bf16_fwd(int kind, ConvKParams &kp, int cin_total, float *out, float *ws, size_t ws_floats, hipStream_t st,
                  const ProfInfo &pi) {
    for (int s = 0; s < kp.nsrc; ++s)
        if (kp.src_c[s] % 32 != 0) return 1;
    {   // the persistent LDS-ring kernel takes the launches it covers (bf16 storage, maps >= 16 x 32; round 6: the first layer too)
        kp.out = out;
        const int rc = conv_ring_try(kind, false, kp, st, pi);   // (writes kp.out_sign itself)
        if (rc != 1) return rc;
    }
    bool tiled = false;   // conv_bf16_kernel / conv_bf16_k5_kernel ran (not the one-shot kernel)
    const int rc = conv_bf16_fwd_rest(kind, kp, cin_total, out, ws, ws_floats, st, pi, &tiled);
    if (rc != PWS_OK || !kp.out_sign || !kp.io_bf16) return rc;
    if (tiled && kp.epi16 && kp.ksplit <= 1) return rc;   // its 16-byte epilogue wrote the sign bytes (no split-K reduce in between)
    const size_t pixels = (size_t)kp.N * kp.OH * kp.OW, items = pixels * (size_t)(kp.cout / 8);
    hipLaunchKernelGGL(sign_bits_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(out), kp.out_ld, pixels,
                       kp.cout / 8, static_cast<unsigned char *>(kp.out_sign), kp.out_sign_ld);
    return check_launch("sign_bits_kernel");
}

// Data-gradient kinds (kp prepared by conv2d_bwd_data_impl).
int conv_bf16_dgrad(int kind, ConvKParams &kp, int cout_f, float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi) {
    if (cout_f % 32 != 0) return 1;
    {
        int rc = conv_ring_try(kind, true, kp, st, pi);
        if (rc != 1) return rc;
        rc = conv_skinny16_try(kind, true, kp, cout_f, nullptr, ws, ws_floats, st, pi);
        if (rc != 1) return rc;
    }
    switch (kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        if (const TileChoice *d = deep_tile(kBK3S1Deep, kp, cout_f)) return select_and_launch(d, 1, kp, cout_f, nullptr, ws, ws_floats, st, pi, kFillBlocksBf16);
        return select_and_launch(kBK3S1, 5, kp, cout_f, nullptr, ws, ws_floats, st, pi, kFillBlocksBf16);
    case PWS_CONV_K3S2:
        if (const TileChoice *d = deep_tile(kBSP3Deep, kp, cout_f)) return select_and_launch(d, 1, kp, cout_f, nullptr, ws, ws_floats, st, pi, kFillBlocksBf16);
        return select_and_launch(kBSP3, 5, kp, cout_f, nullptr, ws, ws_floats, st, pi, kFillBlocksBf16);
    case PWS_CONVT_K4S2: return select_and_launch(kBK4S2, 4, kp, cout_f, nullptr, ws, ws_floats, st, pi, kFillBlocksBf16);
    default: return 1;
    }
}

// [planes][krows][ncols] fp32 -> [planes][ncols padded to 64][krows padded to 32] bf16 (zero padded); one lane per
// destination pair of k (coalesced 4-byte writes; the strided reads go through L2 -- this runs once per weight update).
__global__ void pack_bf16_kernel(const float *__restrict__ w, unsigned *__restrict__ out, int krows, int ncols, int kpad, int npad,
                                 size_t total_pairs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_pairs) return;
    const int kp2 = kpad / 2;
    const int k = (int)(i % kp2) * 2;
    const size_t t = i / kp2;
    const int n = (int)(t % npad);
    const size_t plane = t / npad;
    float a = 0.f, b = 0.f;
    if (n < ncols) {
        const float *src = w + (plane * krows) * ncols + n;
        if (k < krows) a = src[(size_t)k * ncols];
        if (k + 1 < krows) b = src[(size_t)(k + 1) * ncols];
    }
    out[i] = cvt_pk_bf16(a, b);
}

}  // namespace pws

extern "C" int pws_nchw_to_nhwc_pad(const float *x, float *out, int n, int c, int h, int w, int cpad, pws_stream_t stream) {
    return pws_nchw_to_nhwc_pad_s(x, out, n, c, h, w, cpad, PWS_STORE_FP32, stream);
}

extern "C" int pws_cvt_bf16_to_f32(const void *src, float *dst, size_t count, pws_stream_t stream) {
    PWS_REQUIRE(count == 0 || (src && dst), "pws_cvt_bf16_to_f32: NULL pointer");
    if (count == 0) return PWS_OK;
    hipLaunchKernelGGL(pws::cvt_bf16_to_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, pws::as_stream(stream),
                       static_cast<const unsigned short *>(src), dst, count);
    return pws::check_launch("cvt_bf16_to_f32_kernel");
}

extern "C" int pws_cvt_f32_to_bf16(const float *src, void *dst, size_t count, int accumulate, pws_stream_t stream) {
    PWS_REQUIRE(count % 2 == 0 && (count == 0 || (src && dst)), "pws_cvt_f32_to_bf16: count must be even, pointers non-NULL");
    if (count == 0) return PWS_OK;
    hipLaunchKernelGGL(pws::cvt_f32_to_bf16_kernel, dim3((unsigned)((count / 2 + 255) / 256)), dim3(256), 0, pws::as_stream(stream), src,
                       static_cast<unsigned *>(dst), count / 2, accumulate);
    return pws::check_launch("cvt_f32_to_bf16_kernel");
}

namespace pws {
// sample_stride: floats between consecutive samples of x (0 = dense)
int nchw_to_nhwc_pad_strided(const float *x, size_t sample_stride, float *out, int n, int c, int h, int w, int cpad, int store, hipStream_t st) {
    PWS_REQUIRE(x && out && n >= 0 && c > 0 && h > 0 && w > 0, "pws_nchw_to_nhwc_pad: bad arguments");
    PWS_REQUIRE(cpad >= c && cpad <= 32 && cpad % 4 == 0, "pws_nchw_to_nhwc_pad: cpad %d must be a multiple of 4 in [c, 32]", cpad);
    if (n == 0) return PWS_OK;
    const size_t hw = (size_t)h * w;
    const size_t ss = sample_stride ? sample_stride : (size_t)c * hw;
    if (store == PWS_STORE_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<true>, dim3((unsigned)((hw + 63) / 64), (unsigned)n), dim3(256), 0, st, x, out, c, cpad, hw, ss);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<false>, dim3((unsigned)((hw + 63) / 64), (unsigned)n), dim3(256), 0, st, x, out, c, cpad, hw, ss);
    return check_launch("nchw_to_nhwc_pad_kernel");
}
}  // namespace pws

extern "C" int pws_nchw_to_nhwc_pad_s(const float *x, float *out, int n, int c, int h, int w, int cpad, int store, pws_stream_t stream) {
    return pws::nchw_to_nhwc_pad_strided(x, 0, out, n, c, h, w, cpad, store, pws::as_stream(stream));
}

extern "C" size_t pws_packed_bf16_floats(int planes, int krows, int ncols) {
    if (planes <= 0 || krows <= 0 || ncols <= 0) return 0;
    const size_t kpad = (size_t)(krows + 31) / 32 * 32, npad = (size_t)(ncols + 63) / 64 * 64;
    return (size_t)planes * npad * kpad / 2;
}

extern "C" int pws_pack_weight_bf16(const float *w_packed, void *w_bf16, int planes, int krows, int ncols, pws_stream_t stream) {
    PWS_REQUIRE(w_packed && w_bf16 && planes > 0 && krows > 0 && ncols > 0, "pws_pack_weight_bf16: bad arguments");
    const int kpad = (krows + 31) / 32 * 32, npad = (ncols + 63) / 64 * 64;
    const size_t pairs = (size_t)planes * npad * kpad / 2;
    hipLaunchKernelGGL(pws::pack_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, pws::as_stream(stream), w_packed,
                       static_cast<unsigned *>(w_bf16), krows, ncols, kpad, npad, pairs);
    return pws::check_launch("pack_bf16_kernel");
}

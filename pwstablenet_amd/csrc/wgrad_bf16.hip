// Weight gradient on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- the PWS_MATH_BF16 variant of
// conv_wgrad.hip, same result layout:  dP[class][tap][ci][co] += sum_m x[m*S + tap - pad][ci] * dy[m][co]   (K = pixels).
//
// The contraction runs over PIXELS, but both operands live in memory as [pixel][channel]: an MFMA operand (8 consecutive k
// per lane for one channel) is a COLUMN of that image.  gfx950's transposing LDS read does exactly this:
// ds_read_b64_tr_b16 hands every lane of a 16-lane group one column (4 rows) of a 4x16 bf16 block whose rows are addressed
// per lane, so the tiles are staged row-major (one coalesced 16-byte write per 8 channels) and read column-major, two reads
// per operand.  Semantics pinned by tools/probes/tr16_read_probe.hip.
//
//   * A workgroup (4 waves) owns 64 input x 64 output channels of one (class, tap group) and a strided subset of the spatial
//     tiles; wave (a, b) owns the 32 x 32 quadrant (ci half a, co half b) with one accumulator per tap, and walks ALL pixels
//     of the tile, so no cross-wave reduction is needed; the result is added to dW with fp32 atomics as in conv_wgrad.hip.
//   * Per tile the halo'd x tile and the dy tile are converted fp32 -> bf16 (RNE) on their way into LDS as two 32-channel
//     planes of 64-byte rows: the 4 rows x 2 channel halves that one half-wave reads cover all 64 banks exactly once.
//   * The dy fragment of a 16-pixel k-step is shared by all taps of the step.
#include "common.h"
#include <type_traits>

namespace pws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradBfParams {
    const float *src_ptr[4];
    int src_c[4];
    int src_ld[4];
    int nsrc;
    int cin, cin_pad, cout;
    int N, H, W;     // forward input extent
    int LH, LW;      // logical extent walked by tiles
    int OH, OW;      // forward output extent (extent of dy)
    const float *gout;
    int gout_ld;
    float *dw;
    int io_bf16;     // x sources and gout hold bf16 elements
    int tiles_x, tiles_y, tiles_n;
    int ntiles;
    int ci_blocks, co_blocks;  // of 64 channels
    float *dbias;    // optional: dbias[co] += column sums of dy (pws_conv_bwd_weight_args.dbias)
    // second operand pair (pws_conv_bwd_weight_args.gout2; bf16 storage only): samples N1 .. N - 1 of the tile walk are samples 0 .. N - N1 - 1 of
    // these tensors (same geometry).  A layer shared by stages 2 and 3 then takes ONE launch: one prologue, one set of epilogue atomics on dW
    // (the deep levels' launches are bound by exactly those: 4.7-16.8 MB of fp32 atomics behind 4-16 tiles).  N1 % TN == 0 (wgrad_bf16_launch).
    const float *src_ptr2[4];
    const float *gout2;
    int N1;          // samples of the first pair (== N without a second one)
};

template <int KS_, int STRIDE_, int PAD_, int SUBPIX_, int TH_, int TW_, int TN_, int TG_, bool CI32_ = false, int COW_ = 2>
struct WbCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, PAD = PAD_, SUBPIX = SUBPIX_, TH = TH_, TW = TW_, TN = TN_;
    static constexpr int TG = TG_;  // taps per workgroup
    // COW: 32-channel output blocks (= waves) per input-channel half.  2: the workgroup owns 64 x 64 channels (4 waves).  4: 64 x 128
    // (8 waves): the x tile -- the larger operand of the stride-2 kinds (a 17 x 17 halo for 8 x 8 outputs: 37 of the 45 KB a
    // 64 x 64 workgroup stages per tile) -- is staged once for twice the matrix work.  These kernels are bound by what they stage
    // (15 % matrix-pipe busy at 80 KB per 8.4 MFLOP for the stride-2 kind, profiles/r02_pmc_train_bf16_table.log), not by the pipe.
    static constexpr int COW = COW_, THREADS = 64 * 2 * COW, CO_BLK = 32 * COW;
    static_assert(COW == 2 || (COW == 4 && !CI32_), "output-channel waves");
    // CI32: layers with <= 32 (padded) input channels (the first layer): a workgroup owns 32 input x 64 output channels and its
    // two wave pairs take two consecutive tap groups instead of two input-channel halves (no wave multiplies zeros)
    static constexpr bool CI32 = CI32_;
    static constexpr int TAPS = KS * KS, NGROUPS = CI32 ? (TAPS / TG + 1) / 2 : TAPS / TG;
    static constexpr int XPL = CI32 ? 1 : 2;                     // 32-channel planes of the x tile
    static constexpr int BM = TH * TW * TN, KSTEPS = BM / 16;
    static_assert(BM % 16 == 0 && TAPS % TG == 0, "tile");
    static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    static constexpr int PIX = TN * IH * IW;
    static constexpr int ROW = 64;                               // bytes per LDS row: 32 bf16 channels
    static constexpr int LDS_X = XPL * PIX * ROW, LDS_G = COW * BM * ROW;
    static constexpr int LDS_BYTES = LDS_X + LDS_G + 16;         // + sink for staging items past the tile
    // Stride 2 (round 4): the pixels of a halo row lie in LDS as [even columns | odd columns].  A k-step's 16 output pixels read the
    // input columns 2 c + kx -- one parity -- so with the split they are CONSECUTIVE 64-byte rows and the 4 rows x 2 channel halves of
    // a half-wave's transposing read cover the 64 banks once, as for the stride-1 kinds.  (Dense rows put them 128 bytes apart: two
    // rows per bank group, 38-43 % of the LDS cycles of the stride-2 kinds were bank conflicts, profiles/r03_pmc_train_bf16_table.log.)
    static constexpr bool SPLIT = STRIDE_ == 2;
    static constexpr int HW = (IW + 1) / 2;                      // even columns of a halo row
    // column of the halo tile whose pixel sits at position q of its LDS row
    __host__ __device__ static constexpr int col_at(int q) { return SPLIT ? (q < HW ? 2 * q : 2 * (q - HW) + 1) : q; }
    // LDS row offset (in rows) of filter tap (ky, kx) relative to the k-step's own pixel rows
    __host__ __device__ static constexpr int tap_rows(int ky, int kx) { return SPLIT ? ky * IW + (kx & 1) * HW + (kx >> 1) : ky * IW + kx; }
};

// halo-tile pixel of tile-local output pixel p
template <class C>
__host__ __device__ constexpr int wb_xoff(int p) {
    return ((p / (C::TH * C::TW)) * C::IH + ((p % (C::TH * C::TW)) / C::TW) * C::STRIDE) * C::IW + (p % C::TW) * (C::SPLIT ? 1 : C::STRIDE);
}
// the k-step walk relies on xoff(16 j + c) == xoff(c) + j * xoff(16) for c < 16
template <class C>
constexpr bool wb_linear() {
    for (int j = 0; j < C::KSTEPS; ++j)
        for (int c = 0; c < 16; ++c)
            if (wb_xoff<C>(16 * j + c) != wb_xoff<C>(c) + j * wb_xoff<C>(16)) return false;
    return true;
}

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char *lds, int off0, int off1) {
    typedef __attribute__((address_space(3))) s16x4 *lptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// PAIR: the tile walk covers two operand pairs (WgradBfParams.src_ptr2 / gout2); instantiated for the 64-pixel tiles only -- the deep levels, where
// a launch is its prologue and its atomics -- so that the large tiles' kernels keep their registers (the 16 x 16 tile's prefetch sets fill all 256)
template <class C, bool IO16, bool PAIR = false>
__global__ void __launch_bounds__(C::THREADS, 2) wgrad_bf16_kernel(const WgradBfParams p) {
    static_assert(!PAIR || (IO16 && C::BM <= 128 && !(C::STRIDE == 2 && C::TN == 4)), "operand pairs: bf16 storage, tiles of at most 128 pixels");
    static_assert(wb_linear<C>(), "tile shape breaks the k-step address walk");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lg = lane >> 4;

    const int cb = blockIdx.y;
    const int ci0 = (cb / p.co_blocks) * (C::CI32 ? 32 : 64), co0 = (cb % p.co_blocks) * C::CO_BLK;
    const int cls = C::SUBPIX ? (int)(blockIdx.z & 3) : 0;
    const int tg = C::SUBPIX ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    const int py = cls >> 1, px = cls & 1;
    const int pad_y = C::SUBPIX ? 1 - py : C::PAD, pad_x = C::SUBPIX ? 1 - px : C::PAD;
    const int wci = C::CI32 ? 0 : wv / C::COW, wco = wv % C::COW;  // this wave's 32 x 32 quadrant
    const int wtg = C::CI32 ? 2 * tg + (wv >> 1) : tg;             // ... and tap group (wave-uniform)

    f32x16 acc[C::TG];
#pragma unroll
    for (int t = 0; t < C::TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read addresses (bytes): read q of a k-step covers pixels 8*(lg>>1) + 4q + (li>>2) of the step, this lane
    // supplies row (li>>2) and the 4 columns 4*(li&3).. of its group's 16 channels 16*(lg&1)..
    const int colb = (16 * (lg & 1) + 4 * (li & 3)) * 2;
    int a_lane[2], b_lane[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = 8 * (lg >> 1) + 4 * q + (li >> 2);
        a_lane[q] = wci * C::PIX * C::ROW + wb_xoff<C>(c) * C::ROW + colb;
        b_lane[q] = C::LDS_X + wco * C::BM * C::ROW + c * C::ROW + colb;
    }
    constexpr int SINK = C::LDS_X + C::LDS_G;

    // bias gradient: the waves of the first wave pair of the workgroups of input-channel block 0 / tap group 0 see every pixel of
    // dy exactly once (tiles partition the pixels, the parity classes the pixels of a tile)
    const bool do_bias = p.dbias != nullptr && cb / p.co_blocks == 0 && tg == 0 && wv / C::COW == 0;
    float bsum = 0.f;
    auto compute = [&]() {
        // ---- K steps of 16 pixels over the whole tile
#pragma unroll
        for (int j = 0; j < C::KSTEPS; ++j) {
            const bf16x8 b = tr_pair(lds, b_lane[0] + j * 16 * C::ROW, b_lane[1] + j * 16 * C::ROW);
            if (do_bias) {   // wave-uniform: this lane holds 8 pixels of dy column l31 (masked pixels are zeros)
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 0, 1), one2, bsum, false);   // v_dot2_f32_bf16: exact products
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 2, 3), one2, bsum, false);
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 4, 5), one2, bsum, false);
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 6, 7), one2, bsum, false);
            }
#pragma unroll
            for (int t = 0; t < C::TG; ++t) {
                const int tap = (C::NGROUPS == 1 ? 0 : wtg * C::TG) + t;  // wave-uniform
                if (C::CI32 && tap >= C::TAPS) continue;
                const int toff = C::tap_rows(tap / C::KS, tap % C::KS) * C::ROW;
                const int joff = j * wb_xoff<C>(16) * C::ROW + toff;
                const bf16x8 a = tr_pair(lds, a_lane[0] + joff, a_lane[1] + joff);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
    };

    if constexpr (IO16) {
        // bf16 storage: the whole tile (x halo + dy) of the NEXT tile is in flight in registers while the matrix cores work on
        // the current one (a staging item is one 16-byte load; 256 % 8 == 0, so a lane keeps ONE 8-channel group for all its
        // items: source, channel offset and LDS column are tile-invariant, and an item's LDS row is p0 + 32 * it).
        // Before: batches of 4 loads, each batch waited for before its LDS stores -- 7 exposed memory latencies per tile, which
        // was 80 % of the kernel's time (64->64 @256x256 x32: 305 us, 116 us with the loads removed but the rest in place).
        constexpr int XC = 4 * C::XPL, XPP = C::THREADS / XC;   // 16-byte groups per x pixel, x pixels per pass of the workgroup
        constexpr int GC = 4 * C::COW;                           // 16-byte groups per dy pixel; C::THREADS / GC = 32 pixels per pass
        constexpr int XITS = (C::PIX + XPP - 1) / XPP, GITS = (C::BM + 31) / 32;
        static_assert(XITS <= 16 && GITS <= 16 && C::THREADS / GC == 32, "mask width / dy pass");
        const int c8 = tid % GC, p0 = tid / GC;
        const int cx = tid % XC, px0 = tid / XC;
        int xch = ci0 + cx * 8, xs = 0;
        const bool xc_ok = xch < p.cin;
        while (xs < p.nsrc - 1 && xch >= p.src_c[xs]) xch -= p.src_c[xs], ++xs;
        if (!xc_ok) xs = 0, xch = 0;
        const __bf16 *const xsrc1 = reinterpret_cast<const __bf16 *>(p.src_ptr[xs]) + xch;
        const __bf16 *const xsrc2 = PAIR ? reinterpret_cast<const __bf16 *>(p.src_ptr2[xs]) + xch : xsrc1;
        const size_t xld = p.src_ld[xs];
        const bool gc_ok = co0 + c8 * 8 < p.cout;   // (c8 >> 2 = the 32-channel plane of the dy tile)
        const __bf16 *const gsrc1 = reinterpret_cast<const __bf16 *>(p.gout) + (gc_ok ? co0 + c8 * 8 : 0);
        const __bf16 *const gsrc2 = PAIR ? reinterpret_cast<const __bf16 *>(p.gout2) + (gc_ok ? co0 + c8 * 8 : 0) : gsrc1;
        bool second = false;   // (tile-uniform) the tile being requested belongs to the second operand pair (locate)
        int nlim = p.N1;       // ... and that pair's sample count
        const int xl0 = (cx >> 2) * C::PIX * C::ROW + px0 * C::ROW + (cx & 3) * 16;
        const int gl0 = C::LDS_X + (c8 >> 2) * C::BM * C::ROW + p0 * C::ROW + (c8 & 3) * 16;
        u32x4 rx[XITS], rg[GITS];
        unsigned okx = 0, okg = 0;

        int n0 = 0, y0 = 0, x0 = 0;
        auto locate = [&](int tile) {
            const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, tn_i = tile / (p.tiles_x * p.tiles_y);
            n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
            if constexpr (PAIR) {
                second = n0 >= p.N1;   // (N1 % TN == 0: a tile never straddles the pairs)
                n0 -= second ? p.N1 : 0, nlim = second ? p.N - p.N1 : p.N1;
            }
        };
        auto issue_x = [&](auto B, auto E) {   // items [B, E); all loads unconditional (masked items read a valid dummy address)
            const int iy0 = y0 * C::STRIDE - pad_y, ix0 = x0 * C::STRIDE - pad_x;
            if (B.value == 0) okx = 0;
#pragma unroll
            for (int it = B.value; it < E.value; ++it) {
                const int pix = px0 + XPP * it;
                const int lx = C::col_at(pix % C::IW), ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                const bool ok = pix < C::PIX && xc_ok && n < nlim && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const size_t e = ok ? (size_t)((n * p.H + iy) * p.W + ix) * xld : 0;
                rx[it] = *reinterpret_cast<const u32x4 *>((PAIR && second ? xsrc2 : xsrc1) + e);
                okx |= ok ? (1u << it) : 0u;
            }
        };
        auto issue_g = [&]() {
            okg = 0;
#pragma unroll
            for (int it = 0; it < GITS; ++it) {
                const int m = p0 + 32 * it;
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                const int oy = C::SUBPIX ? 2 * y + py : y, ox = C::SUBPIX ? 2 * x + px : x;
                const bool ok = m < C::BM && gc_ok && n < nlim && y < p.LH && x < p.LW && oy < p.OH && ox < p.OW;
                const size_t e = ok ? (size_t)((n * p.OH + oy) * p.OW + ox) * p.gout_ld : 0;
                rg[it] = *reinterpret_cast<const u32x4 *>((PAIR && second ? gsrc2 : gsrc1) + e);
                okg |= ok ? (1u << it) : 0u;
            }
        };
        auto commit_x = [&](auto B, auto E) {
#pragma unroll
            for (int it = B.value; it < E.value; ++it) {
                const bool ok = (okx >> it) & 1u;
                u32x4 v;
                v.x = ok ? rx[it].x : 0u, v.y = ok ? rx[it].y : 0u, v.z = ok ? rx[it].z : 0u, v.w = ok ? rx[it].w : 0u;
                const int off = (XPP * it + XPP - 1 < C::PIX || px0 + XPP * it < C::PIX) ? xl0 + it * XPP * C::ROW : SINK;
                *reinterpret_cast<u32x4 *>(lds + off) = v;
            }
        };
        auto commit_g = [&]() {
#pragma unroll
            for (int it = 0; it < GITS; ++it) {
                const bool ok = (okg >> it) & 1u;
                u32x4 v;
                v.x = ok ? rg[it].x : 0u, v.y = ok ? rg[it].y : 0u, v.z = ok ? rg[it].z : 0u, v.w = ok ? rg[it].w : 0u;
                const int off = (32 * it + 31 < C::BM || p0 + 32 * it < C::BM) ? gl0 + it * 32 * C::ROW : SINK;
                *reinterpret_cast<u32x4 *>(lds + off) = v;
            }
        };

        // the prefetch registers live across the matrix phase beside the TG accumulators: where that does not fit the 256
        // registers of a wave at 2 workgroups per CU, the x tile (in at most two parts) and the dy tile are each loaded in one go
        // just before they are stored (2-3 exposed latencies per tile)
        constexpr bool PREFETCH = (XITS + GITS) * 4 + C::TG * 16 <= 192;
        constexpr int XH = XITS > 11 ? (XITS + 1) / 2 : XITS;
        constexpr std::integral_constant<int, 0> I0;
        constexpr std::integral_constant<int, XH> IH_;
        constexpr std::integral_constant<int, XITS> IX;
        int tile = blockIdx.x;
        if (PREFETCH && tile < p.ntiles) locate(tile), issue_x(I0, IX), issue_g();
        for (; tile < p.ntiles; tile += gridDim.x) {
            __syncthreads();  // previous tile fully consumed
            if constexpr (PREFETCH) {
                commit_x(I0, IX), commit_g();
            } else {
                locate(tile);
                issue_x(I0, IH_), commit_x(I0, IH_);
                if constexpr (XH < XITS) issue_x(IH_, IX), commit_x(IH_, IX);
                issue_g(), commit_g();
            }
            __syncthreads();
            if (PREFETCH && tile + (int)gridDim.x < p.ntiles) locate(tile + gridDim.x), issue_x(I0, IX), issue_g();
            compute();
        }
    } else {
        for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
            const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, tn_i = tile / (p.tiles_x * p.tiles_y);
            const int n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
            const int iy0 = y0 * C::STRIDE - pad_y, ix0 = x0 * C::STRIDE - pad_x;
            __syncthreads();  // previous tile fully consumed
            // ---- x halo tile, channels ci0..ci0+63 of the (virtually concatenated) sources: 8-channel items, loads of a batch
            // issued unconditionally before the first LDS store (masked items read a valid dummy address)
            {
                constexpr int NIT = C::PIX * 8, ITS = (NIT + C::THREADS - 1) / C::THREADS, BATCH = 4;
    #pragma unroll 1
                for (int it0 = 0; it0 < ITS; it0 += BATCH) {
                    f32x4 r[BATCH][2];
                    u32x4 r16[BATCH];
                    int off[BATCH];
                    bool okv[BATCH];
    #pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        const int item = tid + (it0 + k) * C::THREADS;
                        const int pix = item >> 3, c8 = item & 7;
                        const int lx = C::col_at(pix % C::IW), ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                        const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                        int ch = ci0 + c8 * 8;
                        const bool ok = item < NIT && ch < p.cin && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                        int s = 0;
                        while (s < p.nsrc - 1 && ch >= p.src_c[s]) ch -= p.src_c[s], ++s;
                        const size_t e = ok ? ((size_t)(n * p.H + iy) * p.W + ix) * p.src_ld[s] + ch : 0;
                        if constexpr (IO16) {
                            r16[k] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __bf16 *>(p.src_ptr[ok ? s : 0]) + e);
                        } else {
                            const float *g = p.src_ptr[ok ? s : 0] + e;
                            r[k][0] = *reinterpret_cast<const f32x4 *>(g);
                            r[k][1] = *reinterpret_cast<const f32x4 *>(g + 4);
                        }
                        okv[k] = ok;
                        off[k] = item < NIT && c8 < 4 * C::XPL ? (c8 >> 2) * C::PIX * C::ROW + pix * C::ROW + (c8 & 3) * 16 : SINK;
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
                        *reinterpret_cast<u32x4 *>(lds + off[k]) = v;
                    }
                }
            }
            // ---- dy tile, channels co0..co0+63 at the tile's output pixels
            {
                constexpr int GC = 4 * C::COW;
                constexpr int NIT = C::BM * GC, ITS = (NIT + C::THREADS - 1) / C::THREADS, BATCH = 4;
    #pragma unroll 1
                for (int it0 = 0; it0 < ITS; it0 += BATCH) {
                    f32x4 r[BATCH][2];
                    u32x4 r16[BATCH];
                    int off[BATCH];
                    bool okv[BATCH];
    #pragma unroll
                    for (int k = 0; k < BATCH; ++k) {
                        const int item = tid + (it0 + k) * C::THREADS;
                        const int m = item / GC, c8 = item % GC;
                        const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                        const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                        const int oy = C::SUBPIX ? 2 * y + py : y, ox = C::SUBPIX ? 2 * x + px : x;
                        const int ch = co0 + c8 * 8;
                        const bool ok = item < NIT && ch < p.cout && n < p.N && y < p.LH && x < p.LW && oy < p.OH && ox < p.OW;
                        const size_t e = ok ? ((size_t)(n * p.OH + oy) * p.OW + ox) * p.gout_ld + ch : 0;
                        if constexpr (IO16) {
                            r16[k] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const __bf16 *>(p.gout) + e);
                        } else {
                            r[k][0] = *reinterpret_cast<const f32x4 *>(p.gout + e);
                            r[k][1] = *reinterpret_cast<const f32x4 *>(p.gout + e + 4);
                        }
                        okv[k] = ok;
                        off[k] = item < NIT ? C::LDS_X + (c8 >> 2) * C::BM * C::ROW + m * C::ROW + (c8 & 3) * 16 : SINK;
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
                        *reinterpret_cast<u32x4 *>(lds + off[k]) = v;
                    }
                }
            }
            __syncthreads();
            compute();
        }
    }

    // ---- one atomic per element: rows = input channels of this wave's quadrant, 32 lanes = 32 consecutive output channels
    const int co = co0 + wco * 32 + l31;
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);   // the two k halves
        if (hi == 0 && co < p.cout) atomicAdd(p.dbias + co, bsum);
    }
#pragma unroll
    for (int t = 0; t < C::TG; ++t) {
        const int tap = wtg * C::TG + t;
        if (C::CI32 && tap >= C::TAPS) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (ci < p.cin_pad && co < p.cout)
                atomicAdd(p.dw + ((size_t)(cls * C::TAPS + tap) * p.cin_pad + ci) * p.cout + co, acc[t][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct WbChoice {
    int th, tw, tn;
    int (*launch)(WgradBfParams &, int nclasses, hipStream_t);
};

template <class C>
static int launch_wb(WgradBfParams &p, int nclasses, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16_kernel<C, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16_kernel<C, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wgrad_bf16_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    constexpr bool kPairCfg = C::BM <= 128 && !(C::STRIDE == 2 && C::TN == 4);   // (the stride-2 4 x 4 x 4-sample tile is 2 registers short of carrying a second pair)
    if (p.N1 < p.N && (!kPairCfg || !p.io_bf16 || p.N1 % C::TN != 0 || p.N != 2 * p.N1)) return 1;   // the pair's tiles must not straddle the operand pairs
    p.tiles_x = (p.LW + C::TW - 1) / C::TW, p.tiles_y = (p.LH + C::TH - 1) / C::TH, p.tiles_n = (p.N + C::TN - 1) / C::TN;
    p.ntiles = p.tiles_x * p.tiles_y * p.tiles_n;
    p.ci_blocks = C::CI32 ? (p.cin_pad + 31) / 32 : (p.cin_pad + 63) / 64, p.co_blocks = (p.cout + C::CO_BLK - 1) / C::CO_BLK;
    const long other = (long)p.ci_blocks * p.co_blocks * nclasses * C::NGROUPS;
    // ~2 workgroups per CU overall: every workgroup ends with 64 x 64 x taps fp32 atomics on the same addresses, and with
    // the bf16 MFMA rate that tail is what a 1024-workgroup grid is bound by (64->64 @256x256 x8: 176 us -> 117 us at 512;
    // all weight-gradient launches of a batch-32 step: 256 / 512 / 768 / 1024 -> 7.7 / 7.2 / 8.5 / 9.1 ms)
    long ps = ((C::COW == 4 ? 256 : 512) + other - 1) / other;   // (an 8-wave workgroup fills a CU by itself)
    // The deep levels (few tiles, 2.4-16.8 MB of weights): every pixel split ends with 64 x 64 x taps atomics on the SAME addresses, so with 1-2 tiles
    // per workgroup the launch is the atomics (down_bottom6.conv_same at batch 64: 64 tiles over 32 splits = 18.9 M atomics for 0.6 M weights, 73 us at
    // 65 TFLOP/s).  At least `mint` tiles per workgroup (PWS_OPT_EXPERIMENT 190 + k: mint = k, 190 = no floor as in rounds 1-5).
    const long mint = g_experiment >= 190 && g_experiment < 200 ? g_experiment - 190 : 6;   // (tools/probes/r6e_wgrad.sh: 256 -> 256 @8x8 x 64: 71 / 45 / 39 / 41 us at 0 / 4 / 6 / 8)
    if (mint > 0 && ps > p.ntiles / mint) ps = p.ntiles / mint;
    if (ps > p.ntiles) ps = p.ntiles;
    if (ps < 1 || t_deterministic) ps = 1;   // deterministic: one adding workgroup per (channel block, class, tap group)
    dim3 grid((unsigned)ps, (unsigned)(p.ci_blocks * p.co_blocks), (unsigned)(nclasses * C::NGROUPS));
    if constexpr (kPairCfg) {
        if (p.N1 < p.N) {
            static PerDeviceFlag attr_pair_dev;
            bool &attr_pair = attr_pair_dev.cur();
            if (!attr_pair) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16_kernel<C, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
                if (e != hipSuccess) {
                    set_error("hipFuncSetAttribute(wgrad_bf16_kernel<pair>, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
                    return PWS_EHIP;
                }
                attr_pair = true;
            }
            hipLaunchKernelGGL((wgrad_bf16_kernel<C, true, true>), grid, dim3(C::THREADS), C::LDS_BYTES, st, p);
            return check_launch("wgrad_bf16_kernel");
        }
    }
    if (p.io_bf16)
        hipLaunchKernelGGL((wgrad_bf16_kernel<C, true>), grid, dim3(C::THREADS), C::LDS_BYTES, st, p);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<C, false>), grid, dim3(C::THREADS), C::LDS_BYTES, st, p);
    return check_launch("wgrad_bf16_kernel");
}

template <class C>
static constexpr WbChoice wbchoice() {
    return WbChoice{C::TH, C::TW, C::TN, &launch_wb<C>};
}

//                       KS S  P  subpix TH  TW  TN  TG
using WB_K3S1_T256 = WbCfg<3, 1, 1, 0, 16, 16, 1, 9>;
using WB_K3S1_T128 = WbCfg<3, 1, 1, 0, 8, 16, 1, 9>;
using WB_K3S1_T64 = WbCfg<3, 1, 1, 0, 8, 8, 1, 9>;
using WB_K3S1_T64N4 = WbCfg<3, 1, 1, 0, 4, 4, 4, 9>;
using WB_K3S1_T64N16 = WbCfg<3, 1, 1, 0, 2, 2, 16, 9>;
using WB_K3S2_T64 = WbCfg<3, 2, 1, 0, 8, 8, 1, 9>;
using WB_K3S2_T64N4 = WbCfg<3, 2, 1, 0, 4, 4, 4, 9>;
using WB_K3S2_T64N16 = WbCfg<3, 2, 1, 0, 2, 2, 16, 9>;
using WB_K3S2_T64_W = WbCfg<3, 2, 1, 0, 8, 8, 1, 9, false, 4>;     // 64 x 128 channels, 8 waves
using WB_K5S1_T128 = WbCfg<5, 1, 2, 0, 8, 16, 1, 5>;   // first layer: one kernel row of taps per workgroup
using WB_K5S1_T128H = WbCfg<5, 1, 2, 0, 8, 16, 1, 5, true>;    // ... <= 32 input channels: two kernel rows per workgroup (the 16x16
                                                                // tile of this kind runs out of registers: 272 B of scratch, 3.4x slower)
using WB_CT4_T256 = WbCfg<2, 1, 0, 1, 16, 16, 1, 4>;
using WB_CT4_T64 = WbCfg<2, 1, 0, 1, 8, 8, 1, 4>;
using WB_CT4_T64N4 = WbCfg<2, 1, 0, 1, 4, 4, 4, 4>;
using WB_CT4_T64N16 = WbCfg<2, 1, 0, 1, 2, 2, 16, 4>;

static const WbChoice kWbK3S1[] = {wbchoice<WB_K3S1_T256>(), wbchoice<WB_K3S1_T128>(), wbchoice<WB_K3S1_T64>(), wbchoice<WB_K3S1_T64N4>(),
                                   wbchoice<WB_K3S1_T64N16>()};
static const WbChoice kWbK3S2[] = {wbchoice<WB_K3S2_T64>(), wbchoice<WB_K3S2_T64N4>(), wbchoice<WB_K3S2_T64N16>()};
static const WbChoice kWbK3S2W = wbchoice<WB_K3S2_T64_W>();
static const WbChoice kWbK5[] = {wbchoice<WB_K5S1_T128>(), wbchoice<WB_K5S1_T128H>()};
static const WbChoice kWbCT4[] = {wbchoice<WB_CT4_T256>(), wbchoice<WB_CT4_T64>(), wbchoice<WB_CT4_T64N4>(),
                                  wbchoice<WB_CT4_T64N16>()};

static const WbChoice &pick(const WbChoice *c, int n, int LH, int LW, int N) {
    for (int i = 0; i < n; ++i) {
        const long tiles = (long)((LW + c[i].tw - 1) / c[i].tw) * ((LH + c[i].th - 1) / c[i].th) * ((N + c[i].tn - 1) / c[i].tn);
        const double useful = (double)N * LH * LW / ((double)tiles * c[i].th * c[i].tw * c[i].tn);
        if (useful >= 0.45 || i + 1 == n) return c[i];
    }
    return c[n - 1];
}

// Called by conv2d_bwd_weight_impl after its argument checks.  Returns 1 when the launch is not covered (first layer's
// NCHW window, sources that are not multiples of 32 channels): the caller then runs the fp32 kernel.
int wgrad_ring_try(const pws_conv_bwd_weight_args *a, int cin, hipStream_t st);   // wgrad_ring.hip; 1 = not covered

static int wgrad_bf16_launch_impl(const pws_conv_bwd_weight_args *a, hipStream_t st, bool pair);
int wgrad_bf16_launch(const pws_conv_bwd_weight_args *a, hipStream_t st) { return wgrad_bf16_launch_impl(a, st, false); }
// both operand pairs of a->gout2 in ONE launch of wgrad_bf16_kernel (bf16 storage; called by conv2d_bwd_weight_impl after wgrad_ring_try_pair
// declined); 1 = not covered (tile shapes that would straddle the pairs, fp32 storage): the caller launches the pairs one after the other.
// PWS_OPT_EXPERIMENT 188: never (A/B).
int wgrad_bf16_launch_pair(const pws_conv_bwd_weight_args *a, hipStream_t st) {
    if (g_experiment == 188 || a->store != PWS_STORE_BF16 || !a->gout2) return 1;
    for (int s = 0; s < a->nsrc; ++s)
        if (!a->src2_ptr[s] || (reinterpret_cast<size_t>(a->src2_ptr[s]) & 15)) return 1;
    return wgrad_bf16_launch_impl(a, st, true);
}
static int wgrad_bf16_launch_impl(const pws_conv_bwd_weight_args *a, hipStream_t st, bool pair) {
    if (a->src_nchw) return 1;
    WgradBfParams p{};
    p.nsrc = a->nsrc;
    int cin = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        if (a->src[s].channels % 32 != 0) return 1;
        p.src_ptr[s] = a->src[s].ptr, p.src_c[s] = a->src[s].channels, p.src_ld[s] = a->src[s].ld;
        cin += a->src[s].channels;
    }
    if (a->cout % 8 != 0) return 1;
    p.io_bf16 = a->store == PWS_STORE_BF16;
    if (p.io_bf16) {
        bool ok = a->gout_ld % 8 == 0;
        for (int s = 0; s < a->nsrc; ++s) ok = ok && a->src[s].ld % 8 == 0;
        if (!ok) {
            set_error("pws_conv2d_bwd_weight: bf16 storage needs ld %% 8 == 0 for every source and for gout");
            return PWS_EINVAL;
        }
    }
    if (p.io_bf16 && !pair) {   // the persistent LDS-ring kernel where it is covered (stride-1 kinds, whole 16 x 16 tiles, long tile streams)
        const int rc = wgrad_ring_try(a, cin, st);
        if (rc != 1) return rc;
    }
    p.cin = cin, p.cin_pad = (cin + 15) / 16 * 16, p.cout = a->cout;
    p.N = pair ? 2 * a->n : a->n, p.N1 = a->n, p.H = a->h, p.W = a->w;
    if (pair) {
        if (a->kind == PWS_CONV_K5S1) return 1;
        for (int s = 0; s < a->nsrc; ++s) p.src_ptr2[s] = static_cast<const float *>(a->src2_ptr[s]);
        p.gout2 = a->gout2;
    }
    p.gout = a->gout, p.gout_ld = a->gout_ld, p.dw = a->dw_packed, p.dbias = a->dbias;
    double k2 = 9;
    int nclasses = 1;
    const WbChoice *c = nullptr;
    switch (a->kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        p.OH = p.LH = a->h, p.OW = p.LW = a->w;
        c = &pick(kWbK3S1, 5, p.LH, p.LW, a->n);
        break;
    case PWS_CONV_K3S2:
        p.OH = p.LH = (a->h - 1) / 2 + 1, p.OW = p.LW = (a->w - 1) / 2 + 1;
        c = &pick(kWbK3S2, 3, p.LH, p.LW, a->n);
        // >= 128 output channels on maps that whole 8 x 8 tiles cover: 64 x 128 channels per (8-wave) workgroup.  PWS_OPT_EXPERIMENT 83
        // keeps the 64 x 64 workgroups (A/B)
        if (c == &kWbK3S2[0] && a->cout % 128 == 0 && p.io_bf16 && g_experiment != 83) c = &kWbK3S2W;
        break;
    case PWS_CONVT_K4S2:
        p.LH = a->h, p.LW = a->w, p.OH = 2 * a->h, p.OW = 2 * a->w, nclasses = 4, k2 = 4;
        c = &pick(kWbCT4, 4, p.LH, p.LW, a->n);
        // (a 64 x 128-channel variant as for the stride-2 kind was measured on the layers with >= 128 outputs: 33.2 vs 33.2 ms per step)
        break;
    case PWS_CONV_K5S1:
        p.OH = p.LH = a->h, p.OW = p.LW = a->w, k2 = 25;
        c = &kWbK5[p.cin_pad <= 32 ? 1 : 0];
        break;
    default: return 1;
    }
    if (pair && (a->n % c->tn != 0 || c->th * c->tw * c->tn > 128 || (a->kind == PWS_CONV_K3S2 && c->tn == 4))) return 1;   // (before the profiler scope opens) a tile would straddle the operand pairs / large tiles: no pair kernel
    const double out_pix = (double)p.N * p.OH * p.OW;
    ProfScope prof(KID_WGRAD_BF16, 2.0 * out_pix * a->cout * cin * k2,
                   4.0 * ((double)p.N * a->h * a->w * cin + out_pix * a->cout + k2 * cin * a->cout * (nclasses == 4 ? 4 : 1)), st);
    return c->launch(p, nclasses, st);
}

}  // namespace pws

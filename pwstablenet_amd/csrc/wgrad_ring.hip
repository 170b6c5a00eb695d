// bf16 weight gradient (K = pixels) on a persistent LDS ring: second generation of wgrad_bf16.hip for the launches that dominate
// BASELINE configs[2] / [3] (bf16 activation storage, 16 x 16-pixel tiles, at least 64 input and output channels).
//     dP[class][tap][ci][co] += sum_m x[m * S + tap - pad][ci] * dy[m][co]
//
// Why (profiles/r02_kernel_stats_configs2_bf16.csv): wgrad_bf16_kernel is 30 % of the configs[2] step at ~620 TFLOP/s.  Its tiles
// go global -> VGPR -> ds_write_b128 -> LDS; the 3x3 kind has no registers left to keep the next tile in flight during the matrix
// phase (144 accumulators + 76 staging registers), so every tile pays 2-3 exposed memory latencies, 19 LDS stores per lane and two
// barriers.  Here the tile image in LDS is byte for byte what lies in memory (rows [pixel][32 channels] of 64 bytes), so it is filled
// by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR stage, no LDS store instructions), one tile ahead, by 4 loader waves; the 8
// matrix waves only wait at ONE barrier per tile.  Unlike the forward / data-gradient ring (conv_ring.hip) there is no per-unit
// epilogue: a workgroup keeps its 64 x 64 x taps accumulators over all the tiles it walks and adds them to dW once, at the end.
//   * workgroup = 64 input x 64 output channels x ALL taps of one parity class, one per CU; matrix wave = (32 x 32 quadrant, tap
//     half): 5 / 4 (3x3) or 2 / 2 (2x2 class kernels) accumulators of 16 registers -> 168 registers leave room for 3 waves per SIMD
//     (8 matrix + 4 loader waves);
//   * operands by the transposing LDS read ds_read_b64_tr_b16, addresses as in wgrad_bf16.hip (the tile image is the same);
//   * zero padding / masked pixels: DMA offsets beyond the descriptor's num_records deliver zeros.
#include <type_traits>

#include "common.h"

namespace pws {

typedef float wr_f32x16 __attribute__((ext_vector_type(16)));
typedef short wr_s16x4 __attribute__((ext_vector_type(4)));

struct WgradRingParams {
    const void *src_ptr[4];   // bf16 NHWC sources of the forward layer's virtual concat (multiples of 32 channels)
    int src_c[4], src_ld[4];
    int nsrc;
    int cin, cin_pad, cout;
    int N, H, W;     // forward input extent
    int LH, LW;      // logical extent walked by tiles
    int OH, OW;      // forward output extent (extent of dy)
    const void *gout;
    int gout_ld;
    float *dw;
    int tiles_x, tiles_y, ntiles;
    int ci_blocks, co_blocks;
    float *dbias;
    int xcd_groups;   // 0: 3-D grid; else the number of pixel splits of the XCD-grouped 1-D grid (a multiple of 8)
    // second operand pair (pws_conv_bwd_weight_args.gout2): samples N1 .. N - 1 of the tile walk read these tensors (same geometry)
    const void *src_ptr2[4];
    const void *gout2;
    int N1;           // samples of the first pair (== N without a second one)
};

template <int KS_, int PAD_, int SUBPIX_, bool CI32_ = false, bool WIDE_ = false, int ABL_ = 0, bool S2_ = false, int TH_ = 0, int R_ = 0>
struct WrgCfg {
    static constexpr int ABL = ABL_;   // timing-only ablations (tools/wgrad_wide_ab.sh; results are wrong): 1 no DMA, 2 no operand reads, 4 no matrix instructions, 8 one atomic per accumulator instead of 16
    static constexpr int KS = KS_, PAD = PAD_, SUBPIX = SUBPIX_;
    // SUBPIX: 0 = dense kinds; 1 = one parity class of the transposed k4 s2 layers per workgroup (2x2 taps: a wave's dy fragment
    // feeds only two matrix instructions -- measured slower than wgrad_bf16_kernel); 2 = the class PAIR (py, 0) and (py, 1) per
    // workgroup: both read the same x tile (one more halo column), a matrix wave = (32 x 32 quadrant, class) owns all four taps of
    // its class, so a dy fragment feeds four matrix instructions and the x tile is staged once for eight taps.  The pair's tiles are
    // 8 x 16 pixels (x 9 x 18 + dy 2 x 8 x 16 pixels x 64 channels: 56 KB per tile, two in the ring).
    static constexpr bool PAIR = SUBPIX == 2;
    // S2 (round 4): the 3x3 STRIDE-2 layers (conv k3 s2 p1).  Tiles of 4 x 16 OUTPUT pixels; the x tile is the 9 x 33 input pixels under
    // them, each halo row stored as [17 even columns | 16 odd columns]: a k-step's 16 output pixels read the input columns 2 c + kx --
    // one parity -- i.e. 16 CONSECUTIVE 64-byte rows, as in the stride-1 kinds (tap (ky, kx) = row offset ky * 33 + (kx & 1) * 17 +
    // (kx >> 1)); tile row j sits 2 halo rows further.  wgrad_bf16_kernel stages this kind through registers without prefetch (its
    // 17 x 17 halo leaves no registers): 14 % matrix-pipe busy, 1.8 TB/s (profiles/r03_pmc_train_bf16_table.log).
    static constexpr bool S2 = S2_;
    static_assert(!S2 || (KS_ == 3 && PAD_ == 1 && SUBPIX_ == 0 && !CI32_ && !WIDE_), "S2: conv k3 s2 p1");
    // (TH_ / R_: shorter tiles in a deeper ring -- the tile stream is bound by the bytes IN FLIGHT per tile latency, not by the LDS-DMA rate)
    static constexpr int TH = TH_ ? TH_ : (S2 ? 4 : (PAIR ? 8 : 16)), TW = 16;     // tiles of one sample
    // CI32: the first layer (5x5, 31 -> 32 padded input channels): ONE 32-channel plane of x, a workgroup = 32 input x 64 output
    // channels x all 25 taps, matrix wave = (output-channel half, tap quarter): 7 / 7 / 7 / 4 accumulators.  (wgrad_bf16_kernel
    // walked the tensors three times, once per group of 10 taps: 2.7 GB of HBM traffic for 0.8 GB of operands.)
    static constexpr bool CI32 = CI32_;
    // WIDE (round 4, the class pairs of the transposed layers with cin % 128 == 0): a workgroup = 128 input x 64 output channels x the
    // 8 taps of a class pair, matrix wave = (32-channel plane of x, class) x BOTH output-channel halves x the 4 taps of its class:
    // 8 accumulators.  Why: (a) the dy tile is staged once for twice the matrix work (18 instead of 26 bytes per clock and CU through
    // the LDS-DMA path, which delivers ~20); (b) operand reads: a k-step of a wave is 4 x + 2 dy fragments for 8 matrix instructions
    // (0.75 fragments each) where the 64 x 64 workgroup's waves read 4 + 1 for 4 (1.25: 160 bytes per clock and CU at full matrix
    // rate against the LDS's 128); (c) the four workgroups that read the same tiles (2 channel blocks x 2 class rows at 256 input
    // channels) are issued next to each other on ONE XCD (wrg_launch), so that the re-reads are L2 hits instead of HBM traffic --
    // x and dy of the largest layer are 537 MB each, twice the Infinity Cache.
    static constexpr bool WIDE = WIDE_;
    static_assert(!WIDE || (SUBPIX_ == 2 && !CI32_), "WIDE: class pairs only");
    static constexpr int XPL = CI32 ? 1 : (WIDE ? 4 : 2); // 32-channel planes of the x tile
    static constexpr int NCO = WIDE ? 2 : 1;              // output-channel halves of a matrix wave
    static constexpr int CIB = 32 * XPL;                  // input channels of a workgroup
    static constexpr int NTG = CI32 ? 4 : (PAIR ? 1 : 2); // tap groups: matrix wave = (32 x 32 quadrant, tap group [or class of the pair])
    static constexpr int NG = PAIR ? 4 : 2;               // 32-channel planes of dy: (class of the pair,) output-channel half
    static constexpr int TAPS = KS * KS;
    static constexpr int NT0 = (TAPS + NTG - 1) / NTG;    // taps of a wave (the last group takes the rest)
    static constexpr int BM = TH * TW, KSTEPS = BM / 16;
    static constexpr int IH = S2 ? 2 * TH + 1 : TH + KS - 1, IW = S2 ? 2 * TW + 1 : TW + KS - 1 + (PAIR ? 1 : 0), PIX = IH * IW;
    static constexpr int EW = (IW + 1) / 2;               // S2: even columns of a halo row
    // column of the halo tile at position q of its LDS row; LDS row offset of a tap; LDS rows between consecutive tile rows
    __host__ __device__ static constexpr int col_at(int q) { return S2 ? (q < EW ? 2 * q : 2 * (q - EW) + 1) : q; }
    __host__ __device__ static constexpr int tap_rows(int tap) { return S2 ? (tap / KS) * IW + ((tap % KS) & 1) * EW + ((tap % KS) >> 1) : (tap / KS) * IW + (tap % KS); }
    static constexpr int JROWS = S2 ? 2 * IW : IW;
    static constexpr int ROW = 64;                        // bytes per LDS row: 32 bf16 channels
    static constexpr int XPP = (PIX * 4 + 63) / 64;       // DMA pieces (1 KB) per 32-channel plane of the x tile
    static constexpr int GPP = BM * 4 / 64;               // ... of the dy tile
    static constexpr int XP_BYTES = XPP * 1024, G_OFF = XPL * XP_BYTES, GP_BYTES = GPP * 1024;
    static constexpr int PIECES = XPL * XPP + NG * GPP;
    static constexpr int MWAVES = 8, LWAVES = 4, THREADS = 64 * (MWAVES + LWAVES);
    static constexpr int NL = (PIECES + LWAVES - 1) / LWAVES;
    static constexpr int IMG_BYTES = NL * LWAVES * 1024;
    // ring depth: 3 for the S2 kind -- its tiles are short (4 k-steps, 46 KB): with one tile in flight per tile latency (~2.7 us under load)
    // the stream delivered 8 bytes per clock and CU; two in flight (counted vmcnt) double that
    static constexpr int R = R_ ? R_ : (S2 ? 3 : 2), LDS_BYTES = R * IMG_BYTES;
    static_assert((R - 2) * NL <= 63, "vmcnt immediate");
    static_assert(LDS_BYTES <= 160 * 1024 && R <= 4, "LDS");
};

__device__ __forceinline__ void wrg_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
// the same inside a bracket that saved M0 and restores it (once per tile instead of per piece)
__device__ __forceinline__ void wrg_dma16_m0(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    // (s_nop 3 + the two instructions behind it = the 5 wait states between a VALU write of an SGPR -- v_readfirstlane, or hipcc reloading a spilled
    //  scalar with v_readlane right in front of this statement -- and a VMEM instruction that reads it as descriptor / offset: hipcc pads nothing for inline asm)
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
constexpr unsigned kWrgOob = 0x7ffffff0u;
__device__ __forceinline__ unsigned uniq(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *uniq(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)uniq((unsigned)(a >> 32)) << 32) | uniq((unsigned)a));
}
template <class T>
__device__ __forceinline__ T wrgsel4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}
__device__ __forceinline__ bf16x8 wrg_tr_pair(const unsigned char *lds, int off0, int off1) {
    typedef __attribute__((address_space(3))) wr_s16x4 *lptr;
    const wr_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off0));
    const wr_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + off1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <class C>
__global__ void __launch_bounds__(C::THREADS, 3) wgrad_ring_kernel(const WgradRingParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, li = lane & 15, lg = lane >> 4;

    // (pixel split, channel block, class [row]) of this workgroup.  Plain grids: blockIdx = (split, block, class).  p.xcd_groups: a
    // 1-D grid in which the `members` = blocks x class rows workgroups that walk the SAME tiles sit next to each other on one XCD
    // (consecutive workgroup ids go round the 8 XCDs): id = 8 * (group-in-XCD * members + member) + XCD, split = group-in-XCD * 8 + XCD
    int bx = blockIdx.x, cb = blockIdx.y, bz = blockIdx.z, nsplit = gridDim.x;
    if (p.xcd_groups) {
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int members = p.ci_blocks * p.co_blocks * 2, member = slot % members;
        bx = (slot / members) * 8 + xcd, cb = member >> 1, bz = member & 1, nsplit = p.xcd_groups;
    }
    const int ci0 = (cb / p.co_blocks) * C::CIB, co0 = (cb % p.co_blocks) * 64;
    // bz: the parity class (SUBPIX 1) or py of the class pair (SUBPIX 2: px is the matrix wave's / the dy plane's)
    const int cls = C::SUBPIX == 1 ? bz : 0;
    const int py = C::PAIR ? bz : cls >> 1, px = cls & 1;
    const int pad_y = C::SUBPIX ? 1 - py : C::PAD, pad_x = C::PAIR ? 1 : (C::SUBPIX ? 1 - px : C::PAD);
    const int my_tiles = (bx < p.ntiles) ? (p.ntiles - bx + nsplit - 1) / nsplit : 0;

    if (wv >= C::MWAVES) {
        // =========================================================================================== loader waves
        // the loader's code is instantiated once per loader wave (lw a compile-time constant): which plane / descriptor a DMA piece uses is then
        // known per piece -- with a run-time lw every piece chose its descriptor by chains of scalar selects (16 per piece)
        auto loader = [&](auto LWC) {
        constexpr int lw = decltype(LWC)::value;
        // piece pc = it * 4 + lw of a tile image: [x plane 0 | x plane 1 | dy plane 0 | dy plane 1], a plane = rows of 64 bytes (4 lanes)
        int desc[C::NL];   // kind << 28 | row-in-plane << 2 | 16-byte slot, or -1 (filler / past the plane)
#pragma unroll
        for (int it = 0; it < C::NL; ++it) {
            const int pc = it * C::LWAVES + lw;
            int kind, pl0;
            if (pc < C::XPL * C::XPP) kind = pc / C::XPP, pl0 = kind * C::XPP;
            else kind = C::XPL + (pc - C::XPL * C::XPP) / C::GPP, pl0 = C::XPL * C::XPP + (kind - C::XPL) * C::GPP;   // dy plane kind - XPL
            const int j = (pc - pl0) * 64 + lane;
            const int row = j >> 2, sp = j & 3;
            const bool ok = pc < C::PIECES && row < (kind < C::XPL ? C::PIX : C::BM);
            desc[it] = ok ? (kind << 27 | row << 2 | sp) : -1;
        }
        // channel block of the x planes inside the virtual concat (sources are multiples of 32 channels: a plane lies in one source)
        int xs[4], xch[4];
        bool xok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ch = ci0 + q * 32, s = 0;
            xok[q] = q < C::XPL && ch < p.cin;
            while (s < p.nsrc - 1 && ch >= wrgsel4(p.src_c, s)) ch -= wrgsel4(p.src_c, s), ++s;
            xs[q] = xok[q] ? s : 0, xch[q] = xok[q] ? ch : 0;
        }
        const bool gok[2] = {co0 < p.cout, co0 + 32 < p.cout};
        // Per-lane offsets from the tile's first halo pixel (x planes) / first pixel (dy planes): constants of the workgroup, so a tile
        // inside the image costs a loader ONE vector add per DMA piece (every vector instruction issued beside the matrix
        // instructions takes matrix-pipe cycles from its SIMD: conv_first.hip; the divisions and 32-bit multiplies that turned a
        // row number into an address were ~25 instructions per piece and tile).
        constexpr int S = C::SUBPIX ? 2 : 1;
        unsigned ldx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ldx[q] = (unsigned)wrgsel4(p.src_ld, xs[q]) * 2u;
        const unsigned ldg = (unsigned)p.gout_ld * 2u;
        unsigned loc[C::NL];
        int geo[C::NL];   // row-in-tile << 10 | column-in-tile
#pragma unroll
        for (int it = 0; it < C::NL; ++it) {
            const int pc = it * C::LWAVES + lw;
            const int row = (desc[it] >> 2) & 0x1ffffff, sp = desc[it] & 3;
            if (pc < C::XPL * C::XPP) {
                const int q = pc / C::XPP;
                const int lx = C::col_at(row % C::IW), ly = row / C::IW;
                loc[it] = (desc[it] >= 0 && wrgsel4(xok, q)) ? (unsigned)(ly * p.W + lx) * wrgsel4(ldx, q) + (unsigned)(sp * 16) : kWrgOob;
                geo[it] = ly << 10 | lx;
            } else {
                const int q = ((pc - C::XPL * C::XPP) / C::GPP) & 1;
                const int tx = row % C::TW, ty = row / C::TW;
                loc[it] = (desc[it] >= 0 && gok[q]) ? (unsigned)(S * ty * p.OW + S * tx) * ldg + (unsigned)(q * 64 + sp * 16) : kWrgOob;
                geo[it] = ty << 10 | tx;
            }
        }
        // ---- per-tile work of a loader (round 5).  Rounds 2-4 decoded the tile index with three integer divisions, rebuilt five buffer
        // descriptors from 64-bit pointer arithmetic and took every scalar through v_readfirstlane, per tile: ~2 600 lines of ISA per tile
        // with ~480 v_readlane / v_writelane of spilled scalar registers, while a tile is only 4 096 cycles of matrix instructions -- the
        // LOADERS' instruction stream, not the tile stream's bandwidth, is what the matrix waves waited for (57-63 % of their life at the
        // barrier).  Now: the tile coordinates advance by constant increments with carries; a descriptor = the tensor's base + the
        // sample's offset (two scalar adds per tensor and tile; num_records = ONE sample's extent: the out-of-range lanes' zero fill), nothing
        // goes through v_readfirstlane; M0 is saved / restored once per tile.
        const int tpx = p.tiles_x, tpy = p.tiles_y;
        int t_tx = bx % tpx, t_ty = (bx / tpx) % tpy, t_nn = bx / (tpx * tpy);
        const int d_tx = nsplit % tpx, d_ty = (nsplit / tpx) % tpy, d_nn = nsplit / (tpx * tpy);
        const char *xb1[4], *xb2[4];
        unsigned ximg[4], xrec[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ld = wrgsel4(p.src_ld, xs[q]);
            ximg[q] = (unsigned)((size_t)p.H * p.W * ld * 2);                 // bytes of one sample (< 2^31: wgrad_ring_try)
            xrec[q] = ximg[q] - (unsigned)xch[q] * 2u;
            xb1[q] = static_cast<const char *>(wrgsel4(p.src_ptr, xs[q])) + (size_t)xch[q] * 2;
            xb2[q] = p.N1 < p.N ? static_cast<const char *>(wrgsel4(p.src_ptr2, xs[q])) + (size_t)xch[q] * 2 : xb1[q];
        }
        const unsigned gimg = (unsigned)((size_t)p.OH * p.OW * p.gout_ld * 2), grec = gimg - (unsigned)co0 * 2u;
        const char *gb1 = static_cast<const char *>(p.gout) + (size_t)co0 * 2;
        const char *gb2 = p.N1 < p.N ? static_cast<const char *>(p.gout2) + (size_t)co0 * 2 : gb1;
        int tile = bx, pbuf = 0;
        auto stage = [&]() {
            const unsigned d_base = (unsigned)(pbuf * C::IMG_BYTES) + (unsigned)(lw * 1024);
            pbuf = pbuf + 1 == C::R ? 0 : pbuf + 1;
            if (tile >= p.ntiles || (C::ABL & 1)) return;   // past the last tile: nothing reads that buffer (the waits are vmcnt(0): no piece count to keep)
            const bool second = t_nn >= p.N1;            // (scalar) the tile belongs to the second operand pair
            const unsigned n0 = (unsigned)(second ? t_nn - p.N1 : t_nn);
            const int y0 = t_ty * C::TH, x0 = t_tx * C::TW;
            const int iy0 = C::S2 ? 2 * y0 - 1 : y0 - pad_y, ix0 = C::S2 ? 2 * x0 - 1 : x0 - pad_x;
            __amdgpu_buffer_rsrc_t rx[4];
            unsigned sxs[4], sxv[4];   // sample offset (the instruction's scalar offset) / offset of the tile's first halo pixel inside the sample (added to the lanes' offsets: may be negative at the border)
#pragma unroll
            for (int q = 0; q < C::XPL; ++q) {
                // (the sample goes into the descriptor's BASE -- two scalar adds -- not into the instruction's scalar offset: that offset takes part
                //  in the range check against num_records, which is ONE sample's extent here; measured: every sample but the first read zeros)
                rx[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>((second ? xb2[q] : xb1[q]) + (size_t)n0 * ximg[q]), 0, (int)xrec[q], 0x00020000);
                sxs[q] = 0u;
                sxv[q] = (unsigned)((iy0 * p.W + ix0) * (int)ldx[q]);
            }
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>((second ? gb2 : gb1) + (size_t)n0 * gimg), 0, (int)grec, 0x00020000);
            const unsigned sgs = 0u;
            const bool x_inside = iy0 >= 0 && iy0 + C::IH <= p.H && ix0 >= 0 && ix0 + C::IW <= p.W;   // scalar
            // dy: first pixel of the tile per class of the pair (px = 0 / 1), and whether the whole tile (both classes) lies inside
            const int fy = C::SUBPIX ? 2 * y0 + py : y0;
            const int fx0 = C::SUBPIX ? 2 * x0 + (C::PAIR ? 0 : px) : x0;
            const unsigned sgv0 = (unsigned)((fy * p.OW + fx0) * (int)ldg), sgv1 = sgv0 + ldg;   // (PAIR: class px = 1 starts one dy pixel to the right)
            const bool g_inside = y0 + C::TH <= p.LH && x0 + C::TW <= p.LW && fy + S * (C::TH - 1) < p.OH && fx0 + (C::PAIR ? 1 : 0) + S * (C::TW - 1) < p.OW;
            unsigned keep;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0" : "=s"(keep));
            if (x_inside && g_inside) {
                // ---- a tile inside the image (all but the border tiles): ONE vector add per piece, no predicate
#pragma unroll
                for (int it = 0; it < C::NL; ++it) {
                    const int pc = it * C::LWAVES + lw;   // wave-uniform: the kind of a piece is too
                    const unsigned dst = d_base + (unsigned)(it * C::LWAVES * 1024);
                    if (pc < C::XPL * C::XPP) {
                        const int q = pc / C::XPP;   // wave-uniform
                        const unsigned v = loc[it] + (C::XPL == 1 ? sxv[0] : (C::XPL == 2 ? (q ? sxv[1] : sxv[0]) : (q < 2 ? (q ? sxv[1] : sxv[0]) : (q == 2 ? sxv[2] : sxv[3]))));   // (a filler lane stays out of range: ~2^31 + an offset inside one sample)
                        wrg_dma16_m0(dst, v, C::XPL == 1 ? rx[0] : (C::XPL == 2 ? (q ? rx[1] : rx[0]) : (q < 2 ? (q ? rx[1] : rx[0]) : (q == 2 ? rx[2] : rx[3]))),
                                     C::XPL == 1 ? sxs[0] : (C::XPL == 2 ? (q ? sxs[1] : sxs[0]) : (q < 2 ? (q ? sxs[1] : sxs[0]) : (q == 2 ? sxs[2] : sxs[3]))));
                    } else {
                        const int g = (pc - C::XPL * C::XPP) / C::GPP;   // dy plane: (class of the pair,) output-channel half
                        wrg_dma16_m0(dst, loc[it] + ((C::PAIR && (g >> 1)) ? sgv1 : sgv0), rg, sgs);
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < C::NL; ++it) {
                    const int pc = it * C::LWAVES + lw;
                    const unsigned dst = d_base + (unsigned)(it * C::LWAVES * 1024);
                    if (pc < C::XPL * C::XPP) {
                        const int q = pc / C::XPP;
                        unsigned v = loc[it] + (C::XPL == 1 ? sxv[0] : (C::XPL == 2 ? (q ? sxv[1] : sxv[0]) : (q < 2 ? (q ? sxv[1] : sxv[0]) : (q == 2 ? sxv[2] : sxv[3]))));
                        const int iy = iy0 + (geo[it] >> 10), ix = ix0 + (geo[it] & 0x3ff);
                        v = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? v : kWrgOob;
                        wrg_dma16_m0(dst, v, C::XPL == 1 ? rx[0] : (C::XPL == 2 ? (q ? rx[1] : rx[0]) : (q < 2 ? (q ? rx[1] : rx[0]) : (q == 2 ? rx[2] : rx[3]))),
                                     C::XPL == 1 ? sxs[0] : (C::XPL == 2 ? (q ? sxs[1] : sxs[0]) : (q < 2 ? (q ? sxs[1] : sxs[0]) : (q == 2 ? sxs[2] : sxs[3]))));
                    } else {
                        const int g = (pc - C::XPL * C::XPP) / C::GPP;
                        const int gpx = C::PAIR ? g >> 1 : 0;
                        const int fx = fx0 + gpx;
                        unsigned v = loc[it] + (gpx ? sgv1 : sgv0);
                        const int y = y0 + (geo[it] >> 10), x = x0 + (geo[it] & 0x3ff);
                        const int oy = fy + S * (geo[it] >> 10), ox = fx + S * (geo[it] & 0x3ff);
                        v = (y < p.LH && x < p.LW && oy < p.OH && ox < p.OW) ? v : kWrgOob;
                        wrg_dma16_m0(dst, v, rg, sgs);
                    }
                }
            }
            asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
            // next tile of this workgroup: + nsplit, as increments of (column, row, sample) with carries
            tile += nsplit;
            t_tx += d_tx;
            const int cx = t_tx >= tpx ? 1 : 0;
            t_tx -= cx ? tpx : 0, t_ty += d_ty + cx;
            const int cy = t_ty >= tpy ? 1 : 0;
            t_ty -= cy ? tpy : 0, t_nn += d_nn + cy;
        };
        stage();
#pragma unroll
        for (int k = 2; k < C::R; ++k) stage();
        for (int s = 0; s < my_tiles; ++s) {
            // this wave's pieces of tile s have landed; a ring of R keeps the pieces of tiles s + 1 .. s + R - 2 in flight (in-order
            // return counting: NL per tile) -- as far as those tiles exist (stage() issues nothing past the end: the count is shorter)
            const int ahead = my_tiles - 1 - s < C::R - 2 ? my_tiles - 1 - s : C::R - 2;   // wave-uniform
            if (C::R >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NL) : "memory");
            else if (C::R >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // B_s: everybody's have; the matrix waves are done with tile s - 1
            stage();                                            // tile s + R - 1 into the buffer tile s - 1 occupied (nothing past the end)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the filler pieces still target this workgroup's LDS
        };
        const int lw_rt = wv - C::MWAVES;
        if (lw_rt == 0) loader(std::integral_constant<int, 0>{});
        else if (lw_rt == 1) loader(std::integral_constant<int, 1>{});
        else if (lw_rt == 2) loader(std::integral_constant<int, 2>{});
        else loader(std::integral_constant<int, 3>{});
        return;
    }

    // =============================================================================================== matrix waves
    const int quad = C::CI32 ? (wv & 1) : (wv & 3), th = C::CI32 ? (wv >> 1) : (wv >> 2);   // quadrant, tap group (PAIR: class px)
    // WIDE: wave = (x plane wv & 3, class wv >> 2), both output-channel halves
    const int wci = C::CI32 ? 0 : (C::WIDE ? quad : quad >> 1), wco = C::CI32 ? quad : (C::WIDE ? 0 : quad & 1);
    const int t0 = C::PAIR ? 0 : th * C::NT0, nt = C::TAPS - t0 < C::NT0 ? C::TAPS - t0 : C::NT0;   // wave-uniform
    const int wpx = C::PAIR ? th : 0;   // this wave's class of the pair: its x columns start one to the right for px = 1
    wr_f32x16 acc[C::NT0 * C::NCO];
#pragma unroll
    for (int t = 0; t < C::NT0 * C::NCO; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read addresses (bytes), as wgrad_bf16.hip: read q of a k-step covers pixels 8 * (lg >> 1) + 4 q + (li >> 2) of the step,
    // this lane supplies row (li >> 2) and the 4 columns 4 * (li & 3) .. of its group's 16 channels 16 * (lg & 1) ..
    const int colb = (16 * (lg & 1) + 4 * (li & 3)) * 2;
    int a_lane[2], b_lane[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = 8 * (lg >> 1) + 4 * q + (li >> 2);   // pixel of the k-step = tile pixel 16 j + c = (row j, column c)
        a_lane[q] = wci * C::XP_BYTES + c * C::ROW + colb;
        b_lane[q] = C::G_OFF + (wpx * 2 + wco) * C::GP_BYTES + c * C::ROW + colb;
    }
    const bool do_bias = p.dbias != nullptr && cb / p.co_blocks == 0 && (C::PAIR || th == 0) && wci == 0;   // PAIR: both classes' dy
    float bsum = 0.f, bsum1 = 0.f;

    int cbuf = 0;
    for (int s = 0; s < my_tiles; ++s) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_s
        asm volatile("" ::: "memory");
        const unsigned char *img = lds + (unsigned)(cbuf * C::IMG_BYTES);
        constexpr int UNR = C::CI32 || C::WIDE ? 2 : C::KSTEPS;   // (7 / 8 accumulators: fully unrolled, hipcc hoists operand reads until it spills)
#pragma unroll UNR
        for (int j = 0; j < C::KSTEPS; ++j) {   // k-step j = the 16 pixels of tile row j
            if (C::WIDE && (C::ABL & 6)) {   // timing only
                bf16x8 z = {};
                if (!(C::ABL & 2)) z = wrg_tr_pair(img, b_lane[0] + j * 16 * C::ROW, b_lane[1] + j * 16 * C::ROW);
                if (!(C::ABL & 4)) {
#pragma unroll
                    for (int t = 0; t < C::NT0 * C::NCO; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, acc[t], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < C::NT0; ++t) {
                        const int joff = j * C::IW * C::ROW + (((t / C::KS) * C::IW + (t % C::KS) + wpx) * C::ROW);
                        const bf16x8 a = wrg_tr_pair(img, a_lane[0] + joff, a_lane[1] + joff);
                        acc[t][0] += (float)a[0] + (float)z[0];
                    }
                }
                continue;
            }
            const bf16x8 b = wrg_tr_pair(img, b_lane[0] + j * 16 * C::ROW, b_lane[1] + j * 16 * C::ROW);
            bf16x8 b1 = b;
            if (C::WIDE) b1 = wrg_tr_pair(img, b_lane[0] + C::GP_BYTES + j * 16 * C::ROW, b_lane[1] + C::GP_BYTES + j * 16 * C::ROW);
            if (do_bias) {   // wave-uniform: this lane holds 8 pixels of dy column l31 (masked pixels are zeros)
                const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 0, 1), one2, bsum, false);
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 2, 3), one2, bsum, false);
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 4, 5), one2, bsum, false);
                bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b, b, 6, 7), one2, bsum, false);
                if (C::WIDE) {
                    bsum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b1, b1, 0, 1), one2, bsum1, false);
                    bsum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b1, b1, 2, 3), one2, bsum1, false);
                    bsum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b1, b1, 4, 5), one2, bsum1, false);
                    bsum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(b1, b1, 6, 7), one2, bsum1, false);
                }
            }
#pragma unroll
            for (int t = 0; t < C::NT0; ++t) {
                if (t < nt) {
                    const int tap = t0 + t;   // wave-uniform
                    const int toff = (C::tap_rows(tap) + wpx) * C::ROW;
                    const int joff = j * C::JROWS * C::ROW + toff;
                    const bf16x8 a = wrg_tr_pair(img, a_lane[0] + joff, a_lane[1] + joff);
                    acc[t * C::NCO] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t * C::NCO], 0, 0, 0);
                    if (C::WIDE) acc[t * C::NCO + C::NCO - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[t * C::NCO + C::NCO - 1], 0, 0, 0);
                }
            }
        }
        cbuf = cbuf + 1 == C::R ? 0 : cbuf + 1;
    }

    // ---- one atomic per element: rows = input channels of this wave's quadrant, 32 lanes = 32 consecutive output channels
#pragma unroll
    for (int h = 0; h < C::NCO; ++h) {
        const int co = co0 + (C::WIDE ? h : wco) * 32 + l31;
        if (do_bias) {
            float bs = h ? bsum1 : bsum;
            bs += __shfl_xor(bs, 32, 64);   // the two k halves
            if (hi == 0 && co < p.cout) atomicAdd(p.dbias + co, bs);
        }
#pragma unroll
        for (int t = 0; t < C::NT0; ++t) {
            if (t < nt) {
                const int tap = t0 + t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int ocls = C::PAIR ? py * 2 + wpx : cls;
                    if ((C::ABL & 8) && r) continue;
                    if (ci < p.cin_pad && co < p.cout)
                        atomicAdd(p.dw + ((size_t)(ocls * C::TAPS + tap) * p.cin_pad + ci) * p.cout + co, acc[t * C::NCO + h][r]);
                }
            }
        }
    }
}

template <class C>
static int wrg_launch(WgradRingParams &p, int nclasses, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_ring_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wgrad_ring_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    static PerDeviceInt ncu_dev;
    int &ncu = ncu_dev.cur();
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    p.tiles_x = (p.LW + C::TW - 1) / C::TW, p.tiles_y = (p.LH + C::TH - 1) / C::TH;
    p.ntiles = p.tiles_x * p.tiles_y * p.N;
    p.ci_blocks = (p.cin_pad + C::CIB - 1) / C::CIB, p.co_blocks = (p.cout + 63) / 64;
    if (C::PAIR) nclasses = 2;   // grid.z = py: a workgroup takes the classes (py, 0) and (py, 1)
    const long other = (long)p.ci_blocks * p.co_blocks * nclasses;
    // one workgroup per CU and round: the pixel tiles are split over as many workgroups as it takes to give every CU one
    long ps = (ncu + other - 1) / other;
    if (ps > p.ntiles) ps = p.ntiles;
    if (ps < 1 || t_deterministic) ps = 1;
    p.xcd_groups = 0;
    if (C::WIDE && ps >= 8 && g_experiment != 84) {
        // the workgroups that walk the same tiles next to each other on one XCD: a multiple of 8 pixel splits, 1-D grid (kernel head)
        ps = ps / 8 * 8;
        p.xcd_groups = (int)ps;
        hipLaunchKernelGGL((wgrad_ring_kernel<C>), dim3((unsigned)(ps * other)), dim3(C::THREADS), C::LDS_BYTES, st, p);
        return check_launch("wgrad_ring_kernel");
    }
    hipLaunchKernelGGL((wgrad_ring_kernel<C>), dim3((unsigned)ps, (unsigned)(p.ci_blocks * p.co_blocks), (unsigned)nclasses), dim3(C::THREADS), C::LDS_BYTES,
                       st, p);
    return check_launch("wgrad_ring_kernel");
}

// Called by wgrad_bf16_launch (wgrad_bf16.hip) with its checked arguments for the stride-1 kinds with bf16 storage.  Returns 1 when
// not covered: fewer than 64 (padded) input channels, maps that 16 x 16 tiles cover badly, too few tiles for a stream.
static int wgrad_ring_try_impl(const pws_conv_bwd_weight_args *a, int cin, hipStream_t st, bool pair);
int wgrad_ring_try(const pws_conv_bwd_weight_args *a, int cin, hipStream_t st) { return wgrad_ring_try_impl(a, cin, st, false); }
// both operand pairs of a->gout2 in one launch (conv2d_bwd_weight_impl); 1 = not covered (the caller then launches them one after the other)
int wgrad_ring_try_pair(const pws_conv_bwd_weight_args *a, hipStream_t st) {
    if (g_experiment == 89) return 1;   // A/B: never merge
    int cin = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        if (a->src[s].channels % 32 != 0 || a->src[s].ld % 8 != 0 || (reinterpret_cast<size_t>(a->src2_ptr[s]) & 15)) return 1;
        cin += a->src[s].channels;
    }
    if (a->cout % 8 != 0 || a->gout_ld % 8 != 0) return 1;
    return wgrad_ring_try_impl(a, cin, st, true);
}
static int wgrad_ring_try_impl(const pws_conv_bwd_weight_args *a, int cin, hipStream_t st, bool pair) {
    if (a->store != PWS_STORE_BF16 || g_experiment == 80) return 1;   // 80: never (A/B, tests)
    if (a->kind != PWS_CONV_K3S1 && a->kind != PWS_CONVT_K3S1 && a->kind != PWS_CONVT_K4S2 && a->kind != PWS_CONV_K5S1 && a->kind != PWS_CONV_K3S2) return 1;
    const bool first = a->kind == PWS_CONV_K5S1;   // the first layer: one source of 32 (31 + padding) channels
    if (first && (cin != 32 || a->nsrc != 1)) return 1;
    const bool ct4 = a->kind == PWS_CONVT_K4S2;
    const bool s2 = a->kind == PWS_CONV_K3S2;      // round 4: tiles of 4 x 16 output pixels (PWS_OPT_EXPERIMENT 87: never)
    if (s2 && (g_experiment == 87 || a->h % 8 != 0 || a->w % 32 != 0 || cin < 64)) return 1;
    if (!s2 && ((!first && cin < 64) || a->h % (ct4 ? 8 : 16) != 0 || a->w % 16 != 0)) return 1;
    if (a->cout < 32) return 1;
    // Measured (tools/wgrad_ring_bench.sh, batch 64, bf16 storage): the first layer (5x5, 32 -> 64 @256^2) 480 us against 1110 us;
    // 3x3 layers of >= 128 channels 329-343 us against 341-358 us of
    // wgrad_bf16_kernel (+4-5 %); 64 -> 64 @256^2 397 vs 372 us and the transposed kind 464-874 vs 410-767 us (its two-tap waves read
    // a dy fragment per two matrix instructions) -- so only the former is taken (PWS_OPT_EXPERIMENT 81 takes every covered launch).
    // Both kernels stage ~1.07 GB per launch at ~3 TB/s: the tile stream, not the matrix pipe, sets the pace of either.
    // The transposed kind as class pairs (SUBPIX 2): 780 / 416 / 421 / 217 us on the four decoder shapes against 807 / 436 / 423 / 207 us
    // of wgrad_bf16_kernel (one class per workgroup, SUBPIX 1: 910 / 488 / 485 / 248): 54 KB staged per 2 x 1024 matrix cycles is
    // more than the LDS-DMA delivers -- not taken either (81 takes the pairs, 82 the single classes).
    // Round 4: the class pairs on 128 x 64-channel workgroups (WrgCfg<..., WIDE>: dy staged once for twice the matrix work, 0.75 operand
    // fragments per matrix instruction, the workgroups that share tiles grouped on one XCD) -- taken for cin % 128 == 0;
    // PWS_OPT_EXPERIMENT 85: never (the previous selection).
    // (86: every covered launch as 81, but the 64 x 64 pairs)
    const bool wide = ct4 && cin % 128 == 0 && g_experiment != 85 && g_experiment != 86;   // (1300 + mask: timing-only ablations of the wide kernel)
    const bool force = g_experiment == 81 || g_experiment == 86;
    // Round 6: the 64-channel 3x3 layers too (after round 5's loader rewrite the ring wins there: 64 -> 64 @256^2 x 64 377 -> 348 us, @128^2 135 -> 104 us,
    // tools/probes/r6l_wgrad64.sh; PWS_OPT_EXPERIMENT 176: from 128 channels as in rounds 2-5)
    const int ring_min_cin = g_experiment == 176 ? 128 : 64;
    if (!first && !s2 && ((ct4 && !wide) || cin < ring_min_cin) && !force && g_experiment != 82) return 1;
    for (int s = 0; s < a->nsrc; ++s)
        if ((size_t)a->h * a->w * a->src[s].ld * 2 >= (1u << 31) || (reinterpret_cast<size_t>(a->src[s].ptr) & 15) || a->src[s].ld % 8 != 0) return 1;
    const int oh = ct4 ? 2 * a->h : (s2 ? a->h / 2 : a->h), ow = ct4 ? 2 * a->w : (s2 ? a->w / 2 : a->w);
    if ((size_t)oh * ow * a->gout_ld * 2 >= (1u << 31) || (reinterpret_cast<size_t>(a->gout) & 15) || a->gout_ld % 8 != 0) return 1;
    WgradRingParams p{};
    p.nsrc = a->nsrc;
    for (int s = 0; s < a->nsrc; ++s) p.src_ptr[s] = a->src[s].ptr, p.src_c[s] = a->src[s].channels, p.src_ld[s] = a->src[s].ld;
    p.cin = cin, p.cin_pad = (cin + 15) / 16 * 16, p.cout = a->cout;
    p.N = pair ? 2 * a->n : a->n, p.N1 = a->n, p.H = a->h, p.W = a->w, p.LH = s2 ? oh : a->h, p.LW = s2 ? ow : a->w, p.OH = oh, p.OW = ow;
    if (pair) {
        for (int s = 0; s < a->nsrc; ++s) p.src_ptr2[s] = a->src2_ptr[s];
        p.gout2 = a->gout2;
    }
    p.gout = a->gout, p.gout_ld = a->gout_ld, p.dw = a->dw_packed, p.dbias = a->dbias;
    const int nclasses = a->kind == PWS_CONVT_K4S2 ? 4 : 1;
    const long tiles = (s2 ? (long)(oh / 4) * (ow / 16) * a->n : (long)(a->h / (ct4 ? 8 : 16)) * (a->w / 16) * a->n) * (pair ? 2 : 1);
    const long other = (long)((p.cin_pad + (wide ? 127 : 63)) / (wide ? 128 : 64)) * ((a->cout + 63) / 64) * (ct4 ? 2 : 1);
    // a workgroup should stream at least a few tiles (its prologue is one exposed tile load, its tail the atomics)
    if (tiles * other < 256 * 4 && !force) return 1;
    const double k2 = a->kind == PWS_CONVT_K4S2 ? 4 : (first ? 25 : 9);
    const double out_pix = (double)p.N * oh * ow;
    ProfScope prof(KID_WGRAD_RING, 2.0 * out_pix * a->cout * cin * k2,
                   4.0 * ((double)p.N * a->h * a->w * cin + out_pix * a->cout + k2 * cin * a->cout * (nclasses == 4 ? 4 : 1)), st);
    if (ct4 && g_experiment == 82) return (a->h % 16 == 0) ? wrg_launch<WrgCfg<2, 0, 1>>(p, nclasses, st) : 1;   // one class per workgroup (A/B)
    if (ct4 && wide) {
        switch (g_experiment) {   // 1300 + mask: timing-only ablations
        case 1301: return wrg_launch<WrgCfg<2, 0, 2, false, true, 1>>(p, nclasses, st);
        case 1302: return wrg_launch<WrgCfg<2, 0, 2, false, true, 2>>(p, nclasses, st);
        case 1303: return wrg_launch<WrgCfg<2, 0, 2, false, true, 3>>(p, nclasses, st);
        case 1304: return wrg_launch<WrgCfg<2, 0, 2, false, true, 4>>(p, nclasses, st);
        case 1305: return wrg_launch<WrgCfg<2, 0, 2, false, true, 5>>(p, nclasses, st);
        case 1307: return wrg_launch<WrgCfg<2, 0, 2, false, true, 7>>(p, nclasses, st);
        case 1308: return wrg_launch<WrgCfg<2, 0, 2, false, true, 8>>(p, nclasses, st);
        case 1315: return wrg_launch<WrgCfg<2, 0, 2, false, true, 15>>(p, nclasses, st);
        // (4 x 16 tiles in a ring of 4 -- 120 instead of 76 KB in flight -- measured SLOWER on all six decoder shapes, 5-11 %: tools/wgrad_ring_ab2.sh)
        case 88: return wrg_launch<WrgCfg<2, 0, 2, false, true, 0, false, 4, 4>>(p, nclasses, st);
        default: return wrg_launch<WrgCfg<2, 0, 2, false, true>>(p, nclasses, st);
        }
    }
    if (ct4) return wrg_launch<WrgCfg<2, 0, 2>>(p, nclasses, st);
    if (first) return wrg_launch<WrgCfg<5, 2, 0, true>>(p, nclasses, st);
    if (s2) return wrg_launch<WrgCfg<3, 1, 0, false, false, 0, true>>(p, nclasses, st);
    // (8 x 16 tiles in a ring of 4: within +-3 % of the 16 x 16 tiles in a ring of 2 on five layer shapes, tools/wgrad_ring_ab2.sh: kept for A/B)
    if (g_experiment == 88) return wrg_launch<WrgCfg<3, 1, 0, false, false, 0, false, 8, 4>>(p, nclasses, st);
    return wrg_launch<WrgCfg<3, 1, 0>>(p, nclasses, st);
}

}  // namespace pws

// Winograd F(2x2, 3x3) for the 3x3 stride-1 layers (Conv2d k3 s1 p1 and ConvTranspose2d k3 s1 p1 = flipped correlation) in exact
// fp32 on the persistent LDS-ring structure of conv_ring.hip: second generation of conv_wino.hip for the launches that fill
// the chip (BASELINE configs[1], the headline `value`: the six 38.7-GFLOP conv_same layers are 21 % of the step).
//
// Why (profiles/r02_pmc.json): wino_kernel<0> keeps the fp32 matrix pipe 45-50 % busy.  Per 8-channel chunk a
// workgroup stages 32 KB of transformed weights for only 32 tiles x 64 channels (2 048 matrix cycles per wave), global -> VGPR ->
// LDS, writes the transformed input V to LDS and reads it back, and pays two barriers.  Here:
//   * a unit = 16 x 32 output pixels (8 x 16 Winograd tiles) x 32 output channels, walked by ONE persistent 8-wave workgroup per
//     CU; its K dimension streams through an LDS ring of R = 2 slots of 16 input channels, filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds) one slot ahead, one s_barrier per slot; consecutive units form one stream (the first slot of
//     the next unit is in flight under the last matrix phase and the epilogue of the current one).  A slot = raw halo tile
//     18 x 34 pixels x 64 bytes (one L2 request per pixel) + 32 KB of transformed weights for 128 tiles: 72 KB per 8 192 matrix
//     cycles per SIMD = 8.8 bytes per clock and CU (the first-generation kernel needs 18.5 at full rate);
//   * a wave owns ONE TILE ROW (16 tiles) x 32 channels x ALL 16 Winograd components on v_mfma_f32_16x16x4_f32 (128 accumulator
//     registers): lane (tile tx = lane & 15, kq = lane >> 4) reads the 4 x 4 patch of ITS tile for ITS four channels
//     (16 ds_read_b128), applies B^T d B in registers and feeds the 16 x 4 results straight into the matrix instructions as A
//     operands -- no V buffer, no transform pass, no second barrier; the B operand (transformed weights, four k-steps per
//     ds_read_b128) comes from a pack-time layout that is lane-linear in LDS and contiguous in memory (1 KB per DMA instruction);
//   * the whole output transform A^T M A happens inside a lane (it holds all 16 components of its 4 tiles x 2 channels): no
//     cross-wave reduction through LDS; bias + activation, 8-byte stores (channel pair) that are 128 bytes contiguous per 16 lanes;
//   * the matrix waves issue the DMA pieces themselves (9 per wave and slot, right behind the barrier): 128 accumulators + 64
//     operand registers need the 256-register budget of two waves per SIMD, which leaves no room for dedicated loader waves; at 32
//     cycles per fp32 matrix instruction the partner wave of the SIMD covers the ~50 cycles a piece blocks its wave;
//   * raw-tile LDS layout: pixel rows of 64 bytes; inside an image row the even-x pixels come first, then the odd ones, and the
//     four 16-byte slots of a pixel are permuted (slot' = P[s] ^ ((x / 2 / 4) & 3), P = {0, 3, 1, 2}), so that the 16 lanes the
//     hardware serves together in a ds_read_b128 (tiles x and channel slots mixed) hit 16 different 4-bank groups for the
//     patch columns 0 and 1 and at most two-way conflicts for columns 2 and 3.  Zero padding = DMA offsets beyond num_records.
// Numerics: as conv_wino.hip (fp32 Winograd, ~1e-6 relative to the direct form); the K order differs.
//
// MODE 1 -- ConvTranspose2d k4 s2 p1 on the same machinery: output parity class (py, px) is a 2x2 correlation of the input
// (pack.hip), run as Winograd F(2x2, 2x2): 3x3 patches, 9 multiplies for 4 outputs x 4 taps (1.78x fewer), transforms with
// coefficients 0 / +-1 only:   B^T = [[1,-1,0],[0,1,0],[0,-1,1]]   G = [[1,0],[1,1],[0,1]]   A^T = [[1,1,0],[0,1,1]].
// The four classes of a tile read the SAME 4x4 patch as F(2x2,3x3) (class (py, px) its 3x3 sub-patch at (py, px)), so the raw tile,
// its LDS layout and the DMA are unchanged; a unit = 16 x 32 class pixels x 32 channels x the two classes of one py (2 x 9 = 18
// components, 144 accumulator registers, 36 KB of weights per slot), and its outputs are the rows 2 y + py of a 32 x 64 region.
// Unlike F(3x3,2x2) (conv_wino.hip MODE 1: 3-pixel tiles waste 21-41 % of a 64- or 32-pixel map) the tiles divide the maps.
// MODE 2 -- the same with ONE class per unit (9 components, 72 accumulators, 18 KB of weights per slot out of the MODE 1 layout):
// twice the units for the launches whose MODE 1 units do not fill the chip, at 1.4x the staged bytes per matrix instruction.
#include <type_traits>

#include "conv_common.h"

namespace pws {

struct WringParams {
    const float *src_ptr[4];  // fp32 NHWC sources of the virtual concat, each a multiple of 16 channels
    int src_c[4], src_ld[4];
    int nsrc;
    int N, H, W, cout;
    const float *ur;          // ring layout of the transformed weights (wring_index, common.h)
    unsigned ur_bytes;
    const float *bias;
    int act;
    float *out;
    int out_ld;
    int tiles_x, tiles_y;
    unsigned ncob, ncls, nunits;
    int nchunks;              // sum(src_c) / 16
    int ksplit, cps;          // K split: a unit covers the cps = nchunks / ksplit chunks [ks * cps, (ks + 1) * cps) and writes its raw sums to
    size_t split_stride;      // out + ks * split_stride (dense [pixel][cout]); splitk_reduce_kernel adds them, bias + activation there
    int ablate;               // measurement only (PWS_OPT_EXPERIMENT 51..57, or 1000 + mask): 1 = DMA pieces fetch nothing after the
                              // first slot, 2 = no matrix phase, 4 = no epilogue stores, 8 = no A-operand reads, 16 = no B-operand
                              // reads, 32 = no barrier / DMA wait, 64 = no DMA instructions -- results are meaningless, only the
                              // timing is read
};

constexpr int WR_TH = 16, WR_TW = 32;                 // output (class) pixels of a unit: 8 x 16 Winograd tiles
constexpr int WR_RH = WR_TH + 2;                      // rows of the raw halo tile
constexpr int WR_WAVES = 8;
// G16 = 0: maps that are multiples of 32 pixels wide, a unit = 16 x 32 pixels of one sample, raw rows of 34 pixels.
// G16 = 1: 16-pixel-wide maps, a unit = 16 x 16 pixels of TWO samples (tiles 0..7 of a tile row = sample A, 8..15 = sample B), raw rows
//          = [18 pixels of A | 18 pixels of B]: the deep 16 x 16 levels of a batch run on the same code.
template <int MODE, int G16>
struct WrGeo {
    static constexpr int RWV = G16 ? 36 : 34;             // pixels per raw row
    static constexpr int HALF = RWV / 2;                  // even-x pixels come first in an LDS row, then the odd ones
    static constexpr int ROW_SLOTS = RWV * 4, ROW_BYTES = ROW_SLOTS * 16;
    static constexpr int RAW_SLOTS = WR_RH * ROW_SLOTS;
    static constexpr int RAW_PIECES = (RAW_SLOTS + 63) / 64;   // 39 / 41 DMA instructions (1 KB each)
    static constexpr int RAW_IT = (RAW_PIECES + WR_WAVES - 1) / WR_WAVES;
    static constexpr int U_OFF = RAW_PIECES * 1024;
    static constexpr int NC = MODE == 0 ? 16 : (MODE == 1 ? 18 : 9);   // components a wave accumulates
    static constexpr int NCLS = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);   // units per (tile, channel block): MODE 1 = the two py, MODE 2 = the four classes
    static constexpr int U_PIECES = NC * 2;              // NC components x 16 channels x 32 output channels x 4 bytes / 1 KB
    static constexpr int U_IT = (U_PIECES + WR_WAVES - 1) / WR_WAVES;
    static constexpr int NIT = RAW_IT + U_IT;            // DMA pieces per wave and slot (8 .. 11)
    static constexpr int GROUP_BYTES = (RAW_PIECES + U_PIECES) * 1024;
    static constexpr int LDS_BYTES = 2 * GROUP_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NIT <= NC + 2 && NIT <= 11, "DMA pieces are dealt to the first components");
};

// (the DMA / wait / uniformity helpers are those of conv_ring.hip; see the comments there)
__device__ __forceinline__ void wring_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
// timing-only variants of the statement above (ablation): the DMA without the M0 writes (lands wherever M0 points), the M0 writes alone
__device__ __forceinline__ void wring_dma16_nom0(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff));
}
__device__ __forceinline__ void wring_m0_only(unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds_addr));
}
constexpr unsigned kWringOob = 0x7ffffff0u;
__device__ __forceinline__ unsigned uniw(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *uniw(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)uniw((unsigned)(a >> 32)) << 32) | uniw((unsigned)a));
}
template <class T>
__device__ __forceinline__ T selw4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

struct WringUnit {
    int n, y0, x0, cob, py, ks;
};
__device__ __forceinline__ WringUnit wring_unit(const WringParams &p, unsigned u) {
    WringUnit r;
    const unsigned cls = u % p.ncls, u2 = u / p.ncls;
    const unsigned cob = u2 % p.ncob, u3 = u2 / p.ncob;
    const unsigned tile = u3 / (unsigned)p.ksplit;
    r.ks = (int)(u3 % (unsigned)p.ksplit);
    r.py = (int)cls;   // MODE 1: py; MODE 2: py * 2 + px
    const unsigned tx = tile % (unsigned)p.tiles_x, t2 = tile / (unsigned)p.tiles_x;
    r.cob = (int)cob, r.x0 = (int)tx, r.y0 = (int)(t2 % (unsigned)p.tiles_y), r.n = (int)(t2 / (unsigned)p.tiles_y);
    return r;
}

template <int MODE, int G16, int ABL>   // ABL: timing-only ablation mask (WringParams.ablate), 0 in the product
__global__ void __launch_bounds__(WR_WAVES * 64, 2) wino_ring_kernel(const WringParams p) {
    using G_ = WrGeo<MODE, G16>;
    constexpr int WR_ROW_BYTES = G_::ROW_BYTES, WR_U_OFF = G_::U_OFF, WR_RAW_IT = G_::RAW_IT, WR_RAW_PIECES = G_::RAW_PIECES;
    constexpr int NC = G_::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;

    // unit assignment: as conv_ring.hip (XCD-contiguous chunk, round-robin inside the XCD); the 32-channel blocks of one tile are
    // consecutive units, so the CUs of an XCD stage the same raw tile at about the same time and it comes from that L2 once
    const unsigned G = gridDim.x;
    const unsigned nxc = G < (unsigned)kXcds ? G : (unsigned)kXcds;
    const unsigned xcd = blockIdx.x % nxc, slot = blockIdx.x / nxc;
    const unsigned nx = G / nxc + (xcd < G % nxc ? 1u : 0u);
    const unsigned c_begin = (unsigned)((unsigned long long)xcd * p.nunits / nxc);
    const unsigned c_end = (unsigned)((unsigned long long)(xcd + 1) * p.nunits / nxc);
    if (c_begin + slot >= c_end) return;
    const unsigned u_begin = c_begin + slot, u_end = c_end, u_step = nx;
    const unsigned my_units = (u_end - u_begin + u_step - 1) / u_step;
    const int nchunks = p.cps;   // chunks of a unit (its K split)
    const unsigned total = my_units * (unsigned)nchunks;

    // ---- DMA offsets of this lane: piece pc = it * 8 + wv covers the 16-byte LDS slots pc * 64 + lane of a slot image.
    // Raw slot j -> (image row, pixel x, channel slot s): row = j / 136; inside the row 4 consecutive slots are one pixel, pixels
    // ordered even x first; the pixel's slots are permuted (header).  The byte offsets of a unit's raw pieces do not change from
    // slot to slot (the chunk goes into the scalar offset), so they are computed once per unit: every plain VALU instruction beside
    // the fp32 matrix instructions costs ~6 cycles of matrix time (tools/probes/mfma_f32_probe.hip: 155 TFLOP/s alone, 131 with one
    // v_add_f32 per matrix instruction, 119 with two) -- recomputing them per slot was a third of the phase's VALU work.
    constexpr int VRAW_N = WR_RAW_IT < 5 ? WR_RAW_IT : 5;   // (the 41st raw piece of the 16-wide geometry, half a piece of wave 0: on the fly)
    unsigned vraw[VRAW_N];
#pragma unroll
    for (int it = 0; it < VRAW_N; ++it) vraw[it] = kWringOob;
    auto raw_off = [&](int pc, int oy, int ox, unsigned ldb) -> unsigned {   // byte offset of this lane's 16 bytes of raw piece pc
        const int j = pc * 64 + lane;
        unsigned v = kWringOob;
        if (pc < WR_RAW_PIECES && j < G_::RAW_SLOTS) {
            const int row = j / G_::ROW_SLOTS, r = j - row * G_::ROW_SLOTS;
            const int pp = r >> 2, sp = r & 3;
            const int par = pp >= G_::HALF ? 1 : 0, q = pp - par * G_::HALF;
            int lx = 2 * q + par, smp = 0;   // G16: 0..17 = sample A, 18..35 = sample B
            if constexpr (G16) smp = lx >= G_::HALF ? 1 : 0, lx -= smp * G_::HALF;
            const int sl = (0x78 >> (2 * (sp ^ ((q >> 2) & 3)))) & 3;   // inverse of P = {0, 3, 1, 2}
            const int ry = oy + row, rx = ox + lx;
            if (ry >= 0 && ry < p.H && rx >= 0 && rx < p.W) v = (unsigned)((smp * p.H + ry) * p.W + rx) * ldb + (unsigned)(sl * 16);
        }
        return v;
    };
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ur), 0, (int)p.ur_bytes, 0x00020000);
    unsigned pu = u_begin;
    int ps = 0, pc0 = 0, pchunk = 0, pbuf = 0;
    bool pfirst = true;
    WringUnit PU = wring_unit(p, pu);
    int s_ld = p.src_ld[0], s_c = p.src_c[0];   // source ps
    const char *s_ptr = reinterpret_cast<const char *>(p.src_ptr[0]);
    auto seek = [&]() {   // cursor to the first chunk of unit PU's K split
        int c = PU.ks * nchunks * 16;
        ps = 0;
        while (ps < p.nsrc - 1 && c >= selw4(p.src_c, ps)) c -= selw4(p.src_c, ps), ++ps;
        pc0 = c, pchunk = 0;
        s_ld = selw4(p.src_ld, ps), s_c = selw4(p.src_c, ps), s_ptr = reinterpret_cast<const char *>(selw4(p.src_ptr, ps));
    };
    seek();
    // One slot's DMA = prep() (scalar: descriptors of the slot the cursor points at, then the cursor moves on) followed by
    // piece(0 .. WR_NIT - 1), which the matrix phase spreads over its first groups of matrix instructions.  Piece it of wave wv:
    // it = 0 .. 4 raw piece it * 8 + wv (it = 4: waves 0 .. 6 only), it = 5 .. 8 weight piece (it - 5) * 8 + wv.
    __amdgpu_buffer_rsrc_t d_rin = rsrc_u, d_rk = rsrc_u, d_ruk = rsrc_u;   // d_rk / d_ruk: what the pieces use (no records when killed)
    unsigned d_base = 0, d_sin = 0, d_su = 0, d_ldb = 0;
    int d_oy = 0, d_ox = 0;
    bool d_kill = false;   // past the last slot (or ablation): the pieces fetch nothing (offsets beyond num_records)
    unsigned d_su0 = 0;     // weight offset of the first chunk of unit pu
    bool p_new = true;      // the cursor entered a new unit or source: the descriptors below are recomputed (once per unit, typically)
    auto prep = [&]() {
        d_kill = pu >= u_end || ((ABL & 1) && !pfirst) || (ABL & 64);
        pfirst = false;
        const __amdgpu_buffer_rsrc_t r_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ur), 0, 0, 0x00020000);   // every offset out of range: zeros
        d_base = uniw((unsigned)(pbuf * G_::GROUP_BYTES));
        pbuf ^= 1;
        if (pu < u_end) {
            if (p_new) {
                p_new = false;
                const size_t img = (size_t)p.H * p.W * s_ld * 4;   // bytes of one sample
                const char *base_in = uniw(s_ptr + (size_t)(PU.n * (G16 ? 2 : 1)) * img);
                d_rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base_in), 0, (int)uniw((unsigned)(img * (G16 ? 2 : 1))), 0x00020000);
                d_ldb = (unsigned)s_ld * 4u;
                d_oy = PU.y0 * WR_TH - 1, d_ox = PU.x0 * WR_TW - 1;
#pragma unroll
                for (int it = 0; it < VRAW_N; ++it) vraw[it] = raw_off(it * WR_WAVES + wv, d_oy, d_ox, d_ldb);
                const unsigned chunk0 = (unsigned)(PU.ks * nchunks);   // of all p.nchunks chunks of the layer
                if constexpr (MODE == 2)   // the px half of the MODE 1 block of (py, channel block, chunk)
                    d_su0 = uniw((unsigned)(((((unsigned)(PU.py >> 1) * p.ncob + (unsigned)PU.cob) * (unsigned)p.nchunks + chunk0) * 2u + (unsigned)(PU.py & 1)) *
                                            (unsigned)(G_::U_PIECES * 1024)));
                else
                    d_su0 = uniw((unsigned)((((unsigned)PU.py * p.ncob + (unsigned)PU.cob) * (unsigned)p.nchunks + chunk0) * (unsigned)(G_::U_PIECES * 1024)));
            }
            d_sin = (unsigned)(pc0 * 4);
            d_su = d_su0 + (unsigned)pchunk * (unsigned)((MODE == 2 ? 2 : 1) * G_::U_PIECES * 1024);
            pc0 += 16, ++pchunk;
            if (pchunk == nchunks) {   // the unit's K split is staged: on to the next unit
                p_new = true;
                pu += u_step;
                if (pu < u_end) {
                    PU = wring_unit(p, pu);
                    seek();
                }
            } else if (pc0 >= s_c) {   // next source of the virtual concat
                p_new = true;
                pc0 = 0, ++ps;
                s_ld = selw4(p.src_ld, ps), s_c = selw4(p.src_c, ps), s_ptr = reinterpret_cast<const char *>(selw4(p.src_ptr, ps));
            }
        }
        d_ruk = d_kill ? r_none : rsrc_u;
        d_rk = d_kill ? r_none : d_rin;
    };
    const unsigned l16 = (unsigned)(lane * 16);
    auto piece = [&](auto it_c) {
        constexpr int it = decltype(it_c)::value;
        if constexpr (it < WR_RAW_IT) {
            const int pc = it * WR_WAVES + wv;   // wave-uniform
            if (it * WR_WAVES + WR_WAVES <= WR_RAW_PIECES || pc < WR_RAW_PIECES) {
                unsigned v;
                if constexpr (it < VRAW_N) v = vraw[it];   // no VALU work here (see vraw)
                else v = raw_off(pc, d_oy, d_ox, d_ldb);
                if constexpr (ABL & 128) wring_dma16_nom0(v, d_rk, d_sin);
                else if constexpr (ABL & 256) wring_m0_only(d_base + (unsigned)(pc * 1024));
                else wring_dma16(d_base + (unsigned)(pc * 1024), v, d_rk, d_sin);
            }
        } else {
            const int u = (it - WR_RAW_IT) * WR_WAVES + wv;
            if ((it - WR_RAW_IT) * WR_WAVES + WR_WAVES <= G_::U_PIECES || u < G_::U_PIECES) {   // wave-uniform
                if constexpr (ABL & 128) wring_dma16_nom0(l16, d_ruk, d_su + (unsigned)(u * 1024));
                else if constexpr (ABL & 256) wring_m0_only(d_base + (unsigned)(WR_U_OFF + u * 1024));
                else wring_dma16(d_base + (unsigned)(WR_U_OFF + u * 1024), l16, d_ruk, d_su + (unsigned)(u * 1024));
            }
        }
    };
    auto stage_all = [&]() {   // the whole slot at once (prologue)
        prep();
        piece(std::integral_constant<int, 0>{}), piece(std::integral_constant<int, 1>{}), piece(std::integral_constant<int, 2>{});
        piece(std::integral_constant<int, 3>{}), piece(std::integral_constant<int, 4>{}), piece(std::integral_constant<int, 5>{});
        piece(std::integral_constant<int, 6>{}), piece(std::integral_constant<int, 7>{});
        if constexpr (G_::NIT > 8) piece(std::integral_constant<int, 8>{});
        if constexpr (G_::NIT > 9) piece(std::integral_constant<int, 9>{});
        if constexpr (G_::NIT > 10) piece(std::integral_constant<int, 10>{});
    };
    static_assert(G_::NIT >= 8 && G_::NIT <= 11, "piece() calls");

    // ---- operand addresses.  A: patch pixel (a, b) of tile (wv, l15) = raw pixel (2 wv + a, 2 l15 + b), channel slot kq
    int offb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int q = (G16 ? (l15 >> 3) * (G_::HALF / 2) + (l15 & 7) : l15) + (b >> 1), par = b & 1;   // G16: tiles 8..15 = the second sample
        const int sp = ((0x9C >> (2 * kq)) & 3) ^ ((q >> 2) & 3);   // P[kq] ^ Q
        offb[b] = ((par * G_::HALF + q) * 4 + sp) * 16;
    }
    const int b_off = WR_U_OFF + lane * 16;   // + (xi * 2 + nt) * 1024
    const bool half1 = wv >= 4;

    f32x4 acc[NC][2];
    unsigned cu = u_begin;
    int cchunk = 0, cbuf = 0;
    WringUnit CU = wring_unit(p, cu);

    // static priority for the second-dispatched half: at equal priority the older wave of a SIMD wins every arbitration and its partner
    // takes the leftovers (MI355X_MICROARCH.md, two waves per SIMD); measured 189 -> 184.5 us on the 256 -> 256 @64^2 launch
    if (half1 && !(ABL & 512)) __builtin_amdgcn_s_setprio(1);
    stage_all();   // slot 0
    for (unsigned s = 0; s < total; ++s) {
        if (!(ABL & 32)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slot s have landed (and its epilogue stores are out)
            __builtin_amdgcn_s_barrier();                       // everybody's have; everybody is done reading slot s - 1
        }
        asm volatile("" ::: "memory");
        prep();                                             // slot s + 1 goes into the buffer slot s - 1 occupied (pieces: below)

        if (cchunk == 0) {
#pragma unroll
            for (int xi = 0; xi < NC; ++xi)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[xi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char *gbp = lds + (unsigned)(cbuf * G_::GROUP_BYTES);
        // first patch row of this wave's tiles in the raw tile (MODE 1: the classes py = 1 start one row further down)
        const unsigned char *gap = gbp + (unsigned)((2 * wv + (MODE == 1 ? CU.py : (MODE == 2 ? CU.py >> 1 : 0))) * WR_ROW_BYTES);
        int offc[4] = {offb[0], offb[1], offb[2], offb[3]};   // patch column offsets (MODE 2: the class's 3x3 sub-patch starts at column px)
        if constexpr (MODE == 2) {
            if (CU.py & 1) offc[0] = offb[1], offc[1] = offb[2], offc[2] = offb[3];
        }
        if (!(ABL & 2)) {
            // The matrix phase is laid out by hand in 64 groups of two matrix instructions (the two 16-channel halves of one
            // component and k-step: alternating accumulators, so no instruction waits for its predecessor's 40-cycle result),
            // fenced by sched_barrier: left to itself hipcc sinks every ds_read to its first use and waits for it there, and
            // chains the four k-steps of one accumulator back to back.  What else a wave has to do is dealt to the groups so that
            // nothing is needed long before it is there, and so that the two waves of a SIMD do not do the same thing at once:
            //   head        : the 4 patch pixels component 0 needs and its B operands -- the other 12 pixels follow in the groups
            //                 of components 0 .. 2 (ablation: all 16 reads in the head cost 33 us of a 208 us launch);
            //   group (xi, 0): the B operands of component xi + 1 (two ds_read_b128, a component = 256 matrix cycles ahead);
            //   group (xi, 3): B^T d B for component xi + 1 (hipcc shares the column-pass terms between components);
            //   group (xi, 0) for the waves 0 - 3, (xi, 2) for their SIMD partners 4 - 7, xi < 9: DMA piece xi of the next slot
            //                 (a wave-uniform branch; two instances of the phase with compile-time positions spill: the
            //                 accumulators of the two instances do not meet in the same registers).
            constexpr int NR = MODE == 0 ? 4 : 3;   // patch rows
            f32x4 d[NR][4];
            auto rd = [&](int a, int b) {
                d[a][b] = (ABL & 8) ? (f32x4){(float)lane, 1.f, (float)a, (float)b} : *reinterpret_cast<const f32x4 *>(gap + offc[b] + a * WR_ROW_BYTES);
            };
            auto colop = [&](int i, int b) -> f32x4 {
                if constexpr (MODE == 0) return i == 0 ? d[0][b] - d[2][b] : (i == 1 ? d[1][b] + d[2][b] : (i == 2 ? d[2][b] - d[1][b] : d[1][b] - d[3][b]));
                else return i == 0 ? d[0][b] - d[1][b] : (i == 1 ? d[1][b] : d[2][b] - d[1][b]);
            };
            auto vop = [&](int xi) -> f32x4 {
                f32x4 v;
                if constexpr (MODE == 0) {
                    const int i = xi >> 2, j = xi & 3;
                    v = j == 0 ? colop(i, 0) - colop(i, 2) : (j == 1 ? colop(i, 1) + colop(i, 2) : (j == 2 ? colop(i, 2) - colop(i, 1) : colop(i, 1) - colop(i, 3)));
                } else {   // component xi = (px class, i, j): the 3x3 sub-patch starts at column px
                    const int c0 = MODE == 2 ? 0 : xi / 9, i = (xi % 9) / 3, j = xi % 3;
                    v = j == 0 ? colop(i, c0) - colop(i, c0 + 1) : (j == 1 ? colop(i, c0 + 1) : colop(i, c0 + 2) - colop(i, c0 + 1));
                }
                asm volatile("" : "+v"(v));   // computed here, not where hipcc finds its first use
                return v;
            };
            // B operands of the components xi .. xi + BD - 1 (requested BD - 1 components ahead; MODE 1 has no registers for a third set)
            constexpr int BD = (MODE == 1 || (MODE == 0 && G16)) ? 2 : 3;
            f32x4 bq[BD][2];
            const bool nob = (ABL & 16) != 0;
            auto rdb = [&](int slot, int xi) {
                bq[slot][0] = nob ? (f32x4){1.f, 2.f, 3.f, (float)(lane + xi)} : *reinterpret_cast<const f32x4 *>(gbp + b_off + (xi * 2 + 0) * 1024);
                bq[slot][1] = nob ? (f32x4){1.f, 2.f, 3.f, (float)(lane - xi)} : *reinterpret_cast<const f32x4 *>(gbp + b_off + (xi * 2 + 1) * 1024);
            };
            if constexpr (MODE == 0) rd(0, 0), rd(0, 2), rd(2, 0), rd(2, 2);
            else rd(0, 0), rd(1, 0), rd(0, 1), rd(1, 1);
            rdb(0, 0);
            if constexpr (BD == 3) rdb(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 vq[2];
            vq[0] = vop(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < NC; ++xi) {
                const int cur = xi & 1, nxt = cur ^ 1;
                const int bc = xi % BD;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    if (st == 1 && xi < NC - (BD - 1)) rdb((xi + BD - 1) % BD, xi + BD - 1);
                    // the remaining patch pixels, each two groups or more ahead of the first component that needs it
                    if constexpr (MODE == 0) {
                        if (xi == 0 && st == 1) rd(0, 1), rd(2, 1);
                        if (xi == 0 && st == 2) rd(0, 3), rd(2, 3);
                        if (xi == 1 && st == 1) rd(1, 0), rd(1, 2);
                        if (xi == 1 && st == 2) rd(1, 1), rd(1, 3);
                        if (xi == 2 && st == 1) rd(3, 0), rd(3, 2);
                        if (xi == 2 && st == 2) rd(3, 1), rd(3, 3);
                    } else {
                        if (xi == 0 && st == 1) rd(0, 2), rd(1, 2);
                        if (xi == 1 && st == 1) rd(2, 0), rd(2, 1);
                        if (xi == 2 && st == 1) rd(2, 2);
                        if constexpr (MODE == 1) {
                            if (xi == 2 && st == 2) rd(0, 3);
                            if (xi == 3 && st == 1) rd(1, 3), rd(2, 3);
                        }
                    }
                    if (xi < G_::NIT && (st == 0 || st == 2) && (st == 2) == half1) {   // wave-uniform
                        if (xi == 0) piece(std::integral_constant<int, 0>{});
                        if (xi == 1) piece(std::integral_constant<int, 1>{});
                        if (xi == 2) piece(std::integral_constant<int, 2>{});
                        if (xi == 3) piece(std::integral_constant<int, 3>{});
                        if (xi == 4) piece(std::integral_constant<int, 4>{});
                        if (xi == 5) piece(std::integral_constant<int, 5>{});
                        if (xi == 6) piece(std::integral_constant<int, 6>{});
                        if (xi == 7) piece(std::integral_constant<int, 7>{});
                        if constexpr (G_::NIT > 8) {
                            if (xi == 8) piece(std::integral_constant<int, 8>{});
                        }
                        if constexpr (G_::NIT > 9) {
                            if (xi == 9) piece(std::integral_constant<int, 9>{});
                        }
                        if constexpr (G_::NIT > 10) {
                            if (xi == 10) piece(std::integral_constant<int, 10>{});
                        }
                    }
                    if (st == 3 && xi < NC - 1) vq[nxt] = vop(xi + 1);
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[cur][st], bq[bc][0][st], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[cur][st], bq[bc][1][st], acc[xi][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            piece(std::integral_constant<int, 0>{}), piece(std::integral_constant<int, 1>{}), piece(std::integral_constant<int, 2>{});
            piece(std::integral_constant<int, 3>{}), piece(std::integral_constant<int, 4>{}), piece(std::integral_constant<int, 5>{});
            piece(std::integral_constant<int, 6>{}), piece(std::integral_constant<int, 7>{});
            if constexpr (G_::NIT > 8) piece(std::integral_constant<int, 8>{});
            if constexpr (G_::NIT > 9) piece(std::integral_constant<int, 9>{});
            if constexpr (G_::NIT > 10) piece(std::integral_constant<int, 10>{});
        }

        if (cchunk == nchunks - 1) {
            // ---- epilogue of unit cu: lane (l15, kq) holds, for the tiles tx = 4 kq + r of its wave's tile row and the channel pair
            // (2 l15, 2 l15 + 1) of the unit's block, all components: Y = A^T M A in registers, bias, activation, 8-byte stores
            // K split: raw sums into this split's dense [pixel][cout] buffer (bias and activation in splitk_reduce_kernel)
            const int co = CU.cob * 32 + 2 * l15;
            const bool part = p.ksplit > 1;
            float bs0 = 0.f, bs1 = 0.f;
            if (p.bias && !part) bs0 = p.bias[co], bs1 = p.bias[co + 1];
            const int eact = part ? PWS_ACT_NONE : p.act;
            const size_t eld = part ? (size_t)p.cout : (size_t)p.out_ld;
            float *eout = p.out + (part ? (size_t)CU.ks * p.split_stride : 0);
            // tile tx = 4 kq + r of the wave's tile row: G16 -> sample 2 n + (tx >> 3), column 2 (tx & 7); else sample n, column 32 x0 + 2 tx
            auto tile_px = [&](int r, int &smp, int &cx) {
                const int tx = 4 * kq + r;
                if constexpr (G16) smp = CU.n * 2 + (tx >> 3), cx = 2 * (tx & 7);
                else smp = CU.n, cx = CU.x0 * WR_TW + 2 * tx;
            };
            if constexpr (MODE == 0) {
                const int oy = CU.y0 * WR_TH + 2 * wv;
                const size_t rs = (size_t)p.W * eld;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y[2][2][2];   // [nt][row][col]
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float tc[4][2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float m0 = acc[i * 4 + 0][nt][r], m1 = acc[i * 4 + 1][nt][r], m2 = acc[i * 4 + 2][nt][r], m3 = acc[i * 4 + 3][nt][r];
                            tc[i][0] = m0 + m1 + m2, tc[i][1] = m1 - m2 - m3;
                        }
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[nt][0][b] = tc[0][b] + tc[1][b] + tc[2][b];
                            y[nt][1][b] = tc[1][b] - tc[2][b] - tc[3][b];
                        }
                    }
                    if (!(ABL & 4)) {
                        int smp, cx;
                        tile_px(r, smp, cx);
                        float *o = eout + ((size_t)(smp * p.H + oy) * p.W + cx) * eld + co;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                *reinterpret_cast<float2 *>(o + a * rs + (size_t)b * eld) =
                                    make_float2(act_apply(y[0][a][b] + bs0, eact), act_apply(y[1][a][b] + bs1, eact));
                    } else if (y[0][0][0] == 12345.678f) {
                        p.out[0] = y[1][1][1];   // keep the accumulators live
                    }
                }
            } else {
                // class outputs (2 ty + a, 2 tx + b) of class (py, px) = output pixels (2 (16 y0 + 2 wv + a) + py, 2 (32 x0 + 2 tx + b) + px)
                const int OW = 2 * p.W;
                const int cpy = MODE == 2 ? CU.py >> 1 : CU.py;
                const int oy = 2 * (CU.y0 * WR_TH + 2 * wv) + cpy;
                const size_t rs = (size_t)2 * OW * eld;   // next class row = two output rows
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int pxc = 0; pxc < (MODE == 2 ? 1 : 2); ++pxc) {
                        const int cpx = MODE == 2 ? (CU.py & 1) : pxc;
                        float y[2][2][2];   // [nt][row][col]
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            float tc[3][2];
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const float m0 = acc[pxc * 9 + i * 3 + 0][nt][r], m1 = acc[pxc * 9 + i * 3 + 1][nt][r], m2 = acc[pxc * 9 + i * 3 + 2][nt][r];
                                tc[i][0] = m0 + m1, tc[i][1] = m1 + m2;
                            }
#pragma unroll
                            for (int b = 0; b < 2; ++b) y[nt][0][b] = tc[0][b] + tc[1][b], y[nt][1][b] = tc[1][b] + tc[2][b];
                        }
                        if (!(ABL & 4)) {
                            int smp, cx;
                            tile_px(r, smp, cx);
                            float *o = eout + ((size_t)(smp * 2 * p.H + oy) * OW + 2 * cx + cpx) * eld + co;
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int b = 0; b < 2; ++b)
                                    *reinterpret_cast<float2 *>(o + a * rs + (size_t)(2 * b) * eld) =
                                        make_float2(act_apply(y[0][a][b] + bs0, eact), act_apply(y[1][a][b] + bs1, eact));
                        } else if (y[0][0][0] == 12345.678f) {
                            p.out[0] = y[1][1][1];   // keep the accumulators live
                        }
                    }
                }
            }
            cchunk = 0, cu += u_step;
            if (cu < u_end) CU = wring_unit(p, cu);
        } else {
            ++cchunk;
        }
        cbuf ^= 1;
    }
}

// ---- weights.  K3S1 / CONVT_K3S1: U = G g G^T (F(2x2,3x3), 16 components) of the packed correlation kernel
// P[tap][cin_pad][cout]; CONVT_K4S2: per output parity class U = G g G^T (F(2x2,2x2), 9 components) of the 2x2 sub-pixel kernel
// P[cls * 4 + dy * 2 + dx][cin_pad][cout] (pack.hip), the two px classes of one py being the 18 components of a unit.
__global__ void wring_pack_kernel(const float *__restrict__ pk, float *__restrict__ ur, int cin_pad, int cout, int ct4) {
    const size_t plane = (size_t)cin_pad * cout;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    wring_pack_element(pk, ur, plane, i, cin_pad, cout, ct4);
}

int wring_pack(const float *pk, float *ur, int cin_pad, int cout, int ct4, hipStream_t st) {
    const size_t plane = (size_t)cin_pad * cout;
    hipLaunchKernelGGL(wring_pack_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, pk, ur, cin_pad, cout, ct4);
    return check_launch("wring_pack_kernel");
}

template <int MODE, int G16, int ABL>
static int wring_launch(const WringParams &p, unsigned grid, hipStream_t st) {
    using G = WrGeo<MODE, G16>;
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_ring_kernel<MODE, G16, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           G::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wino_ring_kernel, %d B LDS): %s", G::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((wino_ring_kernel<MODE, G16, ABL>), dim3(grid), dim3(WR_WAVES * 64), G::LDS_BYTES, st, p);
    return check_launch("wino_ring_kernel");
}

// Runs a 3x3 stride-1 (Winograd F(2x2,3x3)) or transposed k4 s2 (F(2x2,2x2) per parity class) forward launch on the Winograd ring
// kernel when it is covered: fp32 NHWC sources in multiples of 16 channels (16-byte aligned), an input map of whole units -- 16 x 32
// pixels of one sample, or 16 x 16 pixels of two samples for 16-pixel-wide maps --, cout a multiple of 32, 8-byte aligned output
// pixels, a->w_wring (pws_pack_conv_weight_wring) and enough units to occupy the chip, if need be by splitting K over units
// (partial sums in a->ws, added by splitk_reduce_kernel).
// Returns 1 when not covered (the caller goes on to the first-generation Winograd kernel / the direct kernels).
int wring_try(const pws_conv_args *a, const ProfHint &ph, hipStream_t st) {
    if (!a->w_wring || g_experiment == 50) return 1;
    const bool ct4 = a->kind == PWS_CONVT_K4S2;
    if (!ct4 && a->kind != PWS_CONV_K3S1 && a->kind != PWS_CONVT_K3S1) return 1;
    const bool g16 = a->w == 16 && a->n % 2 == 0 && g_experiment != 61;
    if (a->h % WR_TH != 0 || (!g16 && a->w % WR_TW != 0) || a->cout % 32 != 0 || a->out_ld % 2 != 0 || (reinterpret_cast<size_t>(a->out) & 7)) return 1;
    int cin = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        const pws_src &sr = a->src[s];
        if (sr.channels % 16 != 0 || sr.ld % 4 != 0 || (reinterpret_cast<size_t>(sr.ptr) & 15)) return 1;
        if ((size_t)a->h * a->w * 4 * sr.ld * 2 >= (1u << 31)) return 1;
        cin += sr.channels;
    }
    const size_t plane = (size_t)cin * a->cout;
    const size_t ur_bytes = (size_t)(ct4 ? 36 : 16) * plane * 4;
    if (ur_bytes >= (1u << 31)) return 1;
    WringParams p{};
    for (int s = 0; s < a->nsrc; ++s) p.src_ptr[s] = a->src[s].ptr, p.src_c[s] = a->src[s].channels, p.src_ld[s] = a->src[s].ld;
    p.nsrc = a->nsrc, p.N = a->n, p.H = a->h, p.W = a->w, p.cout = a->cout;
    p.ur = a->w_wring, p.ur_bytes = (unsigned)ur_bytes;
    p.bias = a->bias, p.act = a->act, p.out = a->out, p.out_ld = a->out_ld;
    p.tiles_x = g16 ? 1 : a->w / WR_TW, p.tiles_y = a->h / WR_TH;
    static PerDeviceInt ncu_dev;
    int &ncu = ncu_dev.cur();
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    p.ncob = (unsigned)(a->cout / 32), p.ncls = ct4 ? 2u : 1u;
    const unsigned tiles = (unsigned)(p.tiles_x * p.tiles_y) * (unsigned)(g16 ? a->n / 2 : a->n);
    unsigned units1 = tiles * p.ncob * p.ncls;
    p.nchunks = cin / 16;
    // K split over units when there are fewer units than CUs: the smallest divisor of the chunk count that fills the chip (at most 8
    // splits, at least 4 chunks per unit), partial sums through the caller's workspace (PWS_OPT_EXPERIMENT 62: never split)
    const size_t out_floats = (size_t)a->n * (ct4 ? 4 : 1) * a->h * a->w * a->cout;
    auto split_for = [&](unsigned units) {   // 1 = none needed / none possible
        int ks = 1;
        if (units < (unsigned)ncu && a->ws && g_experiment != 62) {
            for (int k = 2; k <= 8; ++k) {
                if (p.nchunks % k != 0 || p.nchunks / k < 4 || (size_t)k * out_floats * 4 > a->ws_bytes) continue;
                ks = k;
                if (units * (unsigned)k >= (unsigned)ncu) break;
            }
        }
        return ks;
    };
    // transposed kind with two-class units that leave CUs without work: one class per unit (MODE 2: twice the units, twice the raw-tile
    // DMA per matrix instruction) or a deeper K split of the two-class units.  Measured at batch 8 (tools/layer_profile.py): the split
    // wins where MODE 2 would have to split as well (1024 -> 256 @16^2: 115 -> 101 us) or where its units keep >= 32 chunks
    // (1024 -> 128 @32^2: 187 -> 178 us); MODE 2 without any split wins over a split to 16 chunks (512 -> 128 @32^2: 100 vs 103 us).
    // (PWS_OPT_EXPERIMENT 59 forces MODE 2, 60 forbids it, 63 = the rule before: MODE 2 whenever the two-class units do not fill the chip)
    bool mode2 = false;
    if (ct4 && units1 < (unsigned)ncu) {
        const int k1 = split_for(units1), k2 = split_for(units1 * 2);
        const bool split1_fills = units1 * (unsigned)k1 >= (unsigned)ncu;
        mode2 = g_experiment == 63 || !(split1_fills && (k2 > 1 || p.nchunks / k1 >= 32));
    }
    if (g_experiment == 60) mode2 = false;
    if (ct4 && g_experiment == 59) mode2 = true;
    if (mode2) p.ncls = 4, units1 *= 2;
    const int ksplit = split_for(units1);
    p.ksplit = ksplit, p.cps = p.nchunks / ksplit, p.split_stride = out_floats;
    p.nunits = units1 * (unsigned)ksplit;
    if (ksplit > 1) p.out = static_cast<float *>(a->ws);
    p.ablate = g_experiment >= 51 && g_experiment <= 57 ? g_experiment - 50 : (g_experiment >= 1000 && g_experiment < 2024 ? g_experiment - 1000 : 0);
    // a unit is cps x 8 192 (9 216) matrix cycles: fewer units than CUs leave CUs idle for the whole launch, and a non-integer
    // number of rounds costs its tail -- taken from 3/4 of the chip upwards (PWS_OPT_EXPERIMENT 58 / 59 force it for the tests)
    if (p.nunits < (unsigned)(ncu * 3 / 4) && g_experiment != 58 && g_experiment != 59) return 1;
    ProfScope prof(ct4 ? KID_CONV_WRING_CT4 : KID_CONV_WRING, ph.flops, ph.bytes, st);   // covers the split-K reduce as well
    const unsigned grid = p.nunits < (unsigned)ncu ? p.nunits : (unsigned)ncu;   // one persistent workgroup per CU
    int rc;
    const int mode = mode2 ? 2 : (ct4 ? 1 : 0);
    if (p.ablate == 0) {
        switch (mode * 2 + (g16 ? 1 : 0)) {
        case 0: rc = wring_launch<0, 0, 0>(p, grid, st); break;
        case 1: rc = wring_launch<0, 1, 0>(p, grid, st); break;
        case 2: rc = wring_launch<1, 0, 0>(p, grid, st); break;
        case 3: rc = wring_launch<1, 1, 0>(p, grid, st); break;
        case 4: rc = wring_launch<2, 0, 0>(p, grid, st); break;
        default: rc = wring_launch<2, 1, 0>(p, grid, st); break;
        }
    } else if (mode == 0 && !g16) {   // timing-only ablations (tools/wring_ablate.sh)
        switch (p.ablate) {
        case 1: rc = wring_launch<0, 0, 1>(p, grid, st); break;
        case 2: rc = wring_launch<0, 0, 2>(p, grid, st); break;
        case 4: rc = wring_launch<0, 0, 4>(p, grid, st); break;
        case 24: rc = wring_launch<0, 0, 24>(p, grid, st); break;
        case 88: rc = wring_launch<0, 0, 88>(p, grid, st); break;
        case 64: rc = wring_launch<0, 0, 64>(p, grid, st); break;
        case 96: rc = wring_launch<0, 0, 96>(p, grid, st); break;
        case 120: rc = wring_launch<0, 0, 120>(p, grid, st); break;
        case 512: rc = wring_launch<0, 0, 512>(p, grid, st); break;
        default: set_error("wino_ring_kernel<0>: ablation mask %d is not instantiated", p.ablate); return PWS_EINVAL;
        }
    } else {
        set_error("wino_ring_kernel: ablations are instantiated for F(2x2,3x3) on 32-pixel-wide maps only");
        return PWS_EINVAL;
    }
    if (rc != PWS_OK || ksplit == 1) return rc;
    ConvKParams kp{};
    kp.cout = a->cout, kp.bias = a->bias, kp.out = a->out, kp.out_ld = a->out_ld, kp.act = a->act;
    kp.ksplit = ksplit, kp.split_stride = out_floats;
    return launch_splitk_reduce(kp, static_cast<const float *>(a->ws), out_floats / 4, st);
}

}  // namespace pws

extern "C" size_t pws_packed_wring_floats(int kind, int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cout % 32 != 0) return 0;
    const size_t plane = (size_t)((cin + 15) / 16 * 16) * cout;
    if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) return 16 * plane;
    if (kind == PWS_CONVT_K4S2) return 36 * plane;
    if (kind == PWS_CONV_K5S1 && cin <= 32 && cin > 16 && cout == 64) return 36 * plane;   // F(2x2,5x5) of the first layer (conv_first_wino.hip)
    return 0;
}

extern "C" int pws_pack_conv_weight_wring(const float *w_packed, float *w_wring, int kind, int cin, int cout, pws_stream_t stream) {
    PWS_REQUIRE(w_packed && w_wring && pws_packed_wring_floats(kind, cin, cout) > 0, "pws_pack_conv_weight_wring: bad arguments (kind %d, cout %d)", kind,
                cout);
    return pws::wring_pack(w_packed, w_wring, (cin + 15) / 16 * 16, cout, kind == PWS_CONVT_K4S2 ? 1 : (kind == PWS_CONV_K5S1 ? 2 : 0), pws::as_stream(stream));
}

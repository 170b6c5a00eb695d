// bf16 implicit-GEMM convolution for gfx950, second generation ("ring" kernel): persistent workgroups that stream the K
// dimension of their tiles through an LDS ring filled by LDS-DMA (buffer_load_dwordx4 ... lds: global/L2 -> LDS with no VGPR
// stage), hand-counted s_waitcnt vmcnt(N) and ONE s_barrier per K group.  It replaces conv_bf16_kernel (conv_bf16.hip) for the
// launches that dominate BASELINE configs[2] / [3] (bf16 activation storage, maps >= 32 pixels wide); everything else
// (fp32 storage, small maps with split-K, the 5x5 first layer) stays on conv_bf16.hip.
//
// Why (profiles/r01_pmc_train_bf16.json): conv_bf16_kernel keeps the matrix pipe 6-32 % busy.  Its chunk
// pipeline is [barrier, wait for the chunk's global loads, 15 ds_write_b128, barrier, MFMAs] with the loads only one short matrix
// phase ahead of their use (hipcc's vmcnt(0) at the register hand-off drains any deeper prefetch), and every 256-pixel tile pays
// its own prologue (first-chunk latency) and epilogue (store burst) with nothing else to run on the CU but one more workgroup in
// the same state.  Here:
//   * a workgroup (one per CU: 8 matrix waves -- 4 TALL ones on 32-wide tiles since round 4, RgCfg::MT -- plus 4 loader waves) walks a contiguous range of
//     (tile, 64-channel block, parity class) units; the K groups
//     of consecutive units form ONE stream, so the DMAs of the next unit's first groups are in flight under the last matrix
//     phases and the epilogue stores of the current unit -- no per-tile prologue, and the store burst overlaps loads;
//   * a K group = 32 input channels, staged as [pixel][32 ch] rows of 64 bytes (+ the weight rows of the same 32 channels): a
//     row is ONE 64-byte L2 request fetched by 4 lanes.  (First cut: 16-channel rows.  rocprofv3: every 32-byte piece cost a whole
//     64-byte request and a texture-addresser cycle per 128-byte line touched -- TCP_TCC_READ_REQ 2x the useful bytes, and the
//     launch bound by the DMA side: with the matrix phase removed it ran 86 % as long.)  The ring holds R = 2 groups (2 x 80 KB
//     for the 3x3 kind); the counted waits keep the pieces of the next group and the epilogue's stores in flight across barriers;
//   * LDS-DMA writes lane-linear (M0 base + lane * 16; tools/probes/lds_dma_probe.hip), so rows cannot be padded against bank
//     conflicts: the four 16-byte slots of a row are permuted by XOR with bits 2..3 of the pixel's x (of the weight row's index)
//     -- applied to the per-lane SOURCE offset on the way in and to the ds_read_b128 address on the way out -- which puts the
//     16 lanes of every ds_read_b128 lane group on 16 different 4-bank slots (checked exhaustively);
//   * zero padding costs nothing: halo pixels outside the image get an offset beyond the buffer descriptor's num_records and
//     the DMA writes zeros (probe);
//   * stride 2 disappears: a 3x3 stride-2 convolution (and the 4x4 stride-2 data gradient of the transposed convolutions) is a
//     sum over the FOUR PARITY PLANES of its input of 2x2 stride-1 convolutions, and a plane is just a strided view for the
//     DMA's per-lane addresses.  So every kind runs the same stride-1 tile code with KS = 3 (3x3 s1) or KS = 2 (all others):
//     no 4x larger halo per matrix instruction, no strided (bank-conflicted) operand reads.
// Arithmetic is that of conv_bf16_kernel: bf16 x bf16 products exact in fp32, fp32 accumulation (v_mfma_f32_32x32x16_bf16), the
// K order differs (plane-major for the stride-2 kinds), one rounding to bf16 in the epilogue.
// Round 6: (a) units of 256 and 128 pixels (RgCfg::PIX) for the launches that give fewer than ~200 units of 512 -- the 32 x 32 ... 8 x 8 levels of a training
// batch, the 64 x 64 ... 16 x 16 levels of batch-8 inference -- which used to fall back to conv_bf16_kernel or leave half the chip idle (conv_ring_try picks the
// unit); (b) the first layer (RM_K5: 5x5 on the NHWC-32 copy of the window) with its 25 x 64 weight rows RESIDENT in LDS and input tiles only in the ring.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace pws {

enum RingMode {
    RM_K3S1 = 0,  // conv k3 s1 p1 (and transposed k3 s1 p1 = flipped taps, and their data gradients): dense in, dense out
    RM_CT4 = 1,   // transposed conv k4 s2 p1 forward: 4 output parity classes, each a 2x2 conv of the dense input
    RM_SP3 = 2,   // data gradient of conv k3 s2 p1: 4 output parity classes of 1/2/2/4 taps over the dense dy
    RM_K3S2 = 3,  // conv k3 s2 p1 forward: sum over the 4 input parity planes of 2x2 convs with 4/2/2/1 taps
    RM_K4S2 = 4,  // data gradient of transposed conv k4 s2 p1 (= conv k4 s2 p1 over dy): 4 input parity planes x 2x2 taps
    RM_K5 = 5     // round 6: the first layer (conv k5 s1 p2 on the NHWC-32 copy of the window, 32 -> 64 channels), forward only: ONE K group per
                  // unit, the 25 x 64 weight rows (100 KB) RESIDENT in LDS for the workgroup's whole life, the ring carries input tiles only
};

struct RingParams {
    const void *src_ptr[4];  // bf16 NHWC sources of the virtual concat
    int src_c[4], src_ld[4];
    int nsrc;
    int N, H, W;     // extent of the source tensors
    int LH, LW;      // logical output extent the tiles walk (class / plane grid)
    int OH, OW;      // extent of the output tensor
    int cout, kpad, npad;
    const void *w_bf;     // [plane][npad][kpad] bf16 (pws_pack_weight_bf16)
    size_t w_bytes;
    const float *bias;
    int act;
    void *out;            // bf16, ndst == 0
    int out_ld;
    int ndst;             // data-gradient scatter, as ConvKParams
    void *dst_ptr[4];
    int dst_c0[4], dst_c1[4], dst_ld[4], dst_acc[4];
    const void *dst_y[4];
    int dst_y_ld[4], dst_act[4];
    unsigned char *out_sign;            // forward: sign bits of `out` (ConvKParams.out_sign), NULL = none
    int out_sign_ld;
    const unsigned char *dst_sign[4];   // data gradient: sign bits of dst_y[s], read instead of it
    int dst_sign_ld[4];
    int tiles_x, tiles_y, tn;   // tn: samples per tile
    unsigned ncob, ncls, nunits;
    unsigned m_cls, m_cob, m_tx, m_ty;   // ceil(2^32 / d) of the four divisors of ring_unit (0: d == 1): u / d = umulhi(u, m), exact for u * d < 2^32 (ring_launch checks)
    int gpp;              // K groups per plane = sum(src_c) / 32
#ifdef PWS_RING_TIMERS
    unsigned long long *timers;   // [workgroup][matrix wave][8]
#endif
    int ablate;           // measurement only (PWS_OPT_EXPERIMENT 41..48): 1 = the DMA pieces fetch nothing after the first groups,
                          // 8 = weight pieces only for a workgroup's first unit (what resident weights would save: 5.5 % at 64 -> 64 @256^2),
                          // 2 = no matrix phase, 4 = no epilogue -- results are meaningless, only the timing is read
};

// MT_: 32-pixel operand rows per matrix wave.  2: eight matrix waves (two per SIMD), each 64 pixels x 64 channels -- every matrix instruction
// reads 1 KB of operands from LDS, and at 128 B per clock the CU's LDS delivers exactly the 8 waves x 4 KB a step of 4 x 8 matrix
// instructions (256 cycles per SIMD) consumes: the matrix phase cannot run faster than the LDS reads (measured: 55-60 % of the peak
// with the DMA and the epilogue switched off).  4 ("tall", 32-wide tiles): FOUR matrix waves (one per SIMD), each 4 tile rows x 64
// channels in 8 accumulators; the 6 halo rows a wave's 4 rows see through the 3 vertical taps are read ONCE per (tap column, channel
// half) and feed 24 matrix instructions with 6 weight reads: 0.5 KB per matrix instruction.
template <int MODE_, int TH_, int TW_, int TN_, int R_, int MT_ = 2>
struct RgCfg {
    static constexpr int MODE = MODE_, TH = TH_, TW = TW_, TN = TN_, R = R_, MT = MT_;
    static_assert(MT == 2 || (MT == 4 && TW == 32 && TN == 1), "tall matrix waves: consecutive operand rows = consecutive tile rows");
    // Pixels per unit.  512 is the unit the large maps run on: a weight row staged for a group feeds 512 pixels, and the matrix phase of a group is
    // as long as the DMA of the next.  Round 6: units of 256 and 128 pixels for the maps that give fewer than ~200 units of 512 (the 32 x 32 ... 8 x 8
    // levels of a training batch, the 64 x 64 ... 16 x 16 levels of batch-8 inference): those launches left half the chip idle or fell back to
    // conv_bf16_kernel's single-buffered chunk pipeline (16-19 % of the bf16 peak, profiles/r05_train_launches_bf16_b64.txt).  A small unit is
    // DMA-bound by construction (the 36 KB of a 3x3 group's weights feed 128-256 pixels instead of 512), so fewer matrix waves do: PIX / 64.
    static constexpr int PIX = TH * TW * TN;
    static_assert(PIX == 512 || PIX == 256 || PIX == 128, "pixels per unit");
    static constexpr int KS = MODE == RM_K3S1 ? 3 : (MODE == RM_K5 ? 5 : 2);
    static constexpr int TAPS = KS * KS;
    static constexpr bool WRES = MODE == RM_K5;          // weights resident behind the ring (one 64-channel block, one K group: every unit multiplies the same rows)
    static constexpr bool TALL = MT == 4 || MODE == RM_K5;   // matrix phase by tile rows (halo rows read once per tap column and channel half, reused over the vertical taps)
    static_assert(MODE != RM_K5 || (TW == 32 && TN == 1 && R_ == 2), "first layer: 32-wide tiles of one sample");
    static constexpr int NPLANES = (MODE == RM_K3S2 || MODE == RM_K4S2) ? 4 : 1;   // input parity planes
    static constexpr int NCLS = (MODE == RM_CT4 || MODE == RM_SP3) ? 4 : 1;   // output parity classes
    // 8 matrix waves (two per SIMD, each 2 tile rows x 64 channels) + 4 loader waves (one per SIMD) that do nothing but issue
    // the LDS-DMA pieces: a DMA piece blocks its wave for the ~30-70 cycles the texture addresser takes per piece, 75 pieces per
    // group -- issued by the matrix waves themselves that time ADDS to the matrix phase (measured: DMA-only, MFMA-only and
    // epilogue-only timings of the one-role kernel summed to its run time), issued by waves of their own it hides under it
    static constexpr int MWAVES = PIX / 32 / MT, LWAVES = 4, THREADS = 64 * (MWAVES + LWAVES);
    static constexpr int WAVES_PER_SIMD = (MWAVES + LWAVES + 3) / 4;   // what __launch_bounds__ promises: 3 -> 168 registers, 2 -> 256
    // PIX pixels per tile = TN samples x TH x TW; a matrix wave's operand is 32 consecutive tile pixels = 32 / TW tile rows
    static_assert((TW == 32 || TW == 16 || TW == 8 || TW == 4) && (TH & (TH - 1)) == 0 && MWAVES >= 1, "tile");
    static constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    static constexpr int CKG = 32;                                // input channels per K group
    static constexpr int ROWB = CKG * 2;                          // bytes per LDS row (pixel / weight row): one 64-byte L2 request
    static constexpr int SPP = ROWB / 16;                         // 16-byte slots per row
    static constexpr int IN_SLOTS = TN * IH * IW * SPP;           // 16-byte slots of the input image of one group
    static constexpr int IN_WI = (IN_SLOTS + 63) / 64;            // wave-instructions (64 slots each)
    static constexpr int WRES_WI = WRES ? TAPS * 64 * SPP / 64 : 0;   // resident weights: wave-instructions of the one-time preload
    static constexpr int WRES_BYTES = WRES_WI * 1024;
    static constexpr int W_WI = WRES ? 0 : TAPS * 64 * SPP / 64;  // TAPS x 64 rows x SPP slots
    static constexpr int NL = (IN_WI + W_WI + LWAVES - 1) / LWAVES; // DMA instructions per loader wave and group
    static constexpr int GROUP_BYTES = NL * LWAVES * 1024;
    static constexpr int W_OFF = IN_WI * 1024;
    static constexpr int LDS_RING_BYTES = R * GROUP_BYTES;
    static constexpr int W_RES_OFF = LDS_RING_BYTES;              // WRES: the resident weight rows
    // the layer's bias vector (conv_ring_try: cout <= BIAS_FLOATS) sits behind the ring -- or, where the ring fills the LDS (the
    // 3x3 kind on 16-wide maps: 2 x 80 KB), in the filler tail of the last ring buffer, which a ring of depth 2 need not write
    // (its waits are vmcnt(0): no piece count to keep constant)
    static constexpr int FILL_BYTES = (NL * LWAVES - IN_WI - W_WI) * 1024;
    static constexpr bool BIAS_IN_FILL = LDS_RING_BYTES + WRES_BYTES + 4096 > 160 * 1024;
    static constexpr int BIAS_FLOATS = BIAS_IN_FILL ? 512 : 1024;
    static constexpr int BIAS_OFF = BIAS_IN_FILL ? LDS_RING_BYTES - BIAS_FLOATS * 4 : LDS_RING_BYTES + WRES_BYTES;
    static constexpr int LDS_BYTES = BIAS_IN_FILL ? LDS_RING_BYTES : LDS_RING_BYTES + WRES_BYTES + BIAS_FLOATS * 4;
    static_assert(!(WRES && BIAS_IN_FILL), "resident weights: the bias slot sits behind them");
    static constexpr bool SKIP_FILL = R == 2;
    static_assert(!BIAS_IN_FILL || (SKIP_FILL && FILL_BYTES >= BIAS_FLOATS * 4), "bias slot");
    // the `it` whose 4 wave-instructions hold input pieces on the first loader waves and weight pieces on the others (-1: none)
    static constexpr int MIX_IT = IN_WI % LWAVES == 0 ? -1 : IN_WI / LWAVES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(R >= 2 && R <= 4, "ring depth");
    static_assert((R - 2) * NL + WRES_WI / LWAVES <= 63, "vmcnt is 6 bits");
};

// one LDS-DMA: 64 lanes x 16 bytes, lane l lands at lds_addr + 16 l; source = rsrc base + soff + voff (lanes beyond
// num_records deliver zeros).  M0 is written inside the statement (hipcc neither preserves nor models it); the leading s_nop
// covers a descriptor / offset SGPR written by a v_readfirstlane just ahead of the statement.
__device__ __forceinline__ void ring_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
    // no "memory" clobber: the statement is volatile, so it keeps its place among the other volatile statements (the waits and the
    // fenced barriers that order it against every LDS access to its destination), while hipcc stays free to schedule the
    // matrix phase's ds_reads and MFMAs around it
}
// the same inside a bracket that saved M0 and restores it (one save / restore per group instead of per piece).  Between the statements of a bracket
// M0 holds the last piece's LDS address while hipcc believes it unchanged; it cannot be told ("m0" on a clobber list is rejected as a reserved
// register), so the contract is the loaders' own: nothing M0-dependent (no s_movrel / s_sendmsg / GWS / v_readlane through m0, no LDS-direct) is
// ever emitted between save and restore -- true of everything a loader wave does (scalar adds, v_add offsets, the DMA instructions).
__device__ __forceinline__ void ring_dma16_m0(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    // (s_nop 3 + the two instructions behind it = the 5 wait states between a VALU write of an SGPR -- v_readfirstlane, or hipcc reloading a spilled
    //  scalar with v_readlane right in front of this statement -- and a VMEM instruction that reads it as descriptor / offset: hipcc pads nothing for inline asm)
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
template <int N>
__device__ __forceinline__ void ring_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ const u32x4 g_ring_zero16 = {0u, 0u, 0u, 0u};   // what the epilogue's loads read in place of an absent tensor

constexpr unsigned kRingOob = 0x7ffffff0u;   // voffset no descriptor of this kernel reaches (num_records < 2^31)

// wave-uniform values the compiler cannot prove uniform (an "s" asm operand needs the proof): through v_readfirstlane
__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *uni(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)uni((unsigned)(a >> 32)) << 32) | uni((unsigned)a));
}

// element i of a 4-entry kernel-argument array by selects: a runtime index would make hipcc copy the array to scratch
template <class T>
__device__ __forceinline__ T sel4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

struct RingUnit {   // decoded (tile, cout block, class)
    int n0, y0, x0, co0, py, px;
};
// (divisions by multiply-high with host-made reciprocals: scalar instructions.  hipcc's own expansion of a 32-bit division goes through
//  v_rcp_iflag_f32 -- vector registers holding wave-uniform reciprocals, hoisted in front of the role split and, in the 256-register
//  kernels, spilled round the matrix waves' whole loop)
__device__ __forceinline__ unsigned ring_div(unsigned u, unsigned m) { return m ? __umulhi(u, m) : u; }
__device__ __forceinline__ RingUnit ring_unit(const RingParams &p, unsigned u) {
    RingUnit r;
    const unsigned rest = ring_div(u, p.m_cls), cls = u - rest * p.ncls;
    const unsigned tile = ring_div(rest, p.m_cob), cob = rest - tile * p.ncob;
    const unsigned t2 = ring_div(tile, p.m_tx), tx = tile - t2 * (unsigned)p.tiles_x;
    const unsigned t3 = ring_div(t2, p.m_ty), ty = t2 - t3 * (unsigned)p.tiles_y;
    r.n0 = (int)t3 * p.tn;
    r.y0 = (int)ty, r.x0 = (int)tx, r.co0 = (int)cob * 64, r.py = (int)(cls >> 1), r.px = (int)(cls & 1);
    return r;
}

// SG (data gradient only): every act' operand is given as SIGN BITS (RingParams.dst_sign): the epilogue loads one byte per slot instead of
// 16 (a runtime choice would keep both in registers: the data-gradient kernels stand at the 168-register limit)
template <class C, bool DG, bool SG = false>   // DG: data-gradient epilogue (scatter over the forward layer's sources, accumulate, act')
__global__ void __launch_bounds__(C::THREADS, C::WAVES_PER_SIMD) conv_ring_kernel(const RingParams p) {
    static_assert(DG || !SG, "sign bits are an operand of the data-gradient epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // Units: XCD x (blocks x, x + 8, ...) takes the contiguous chunk [x U / 8, (x + 1) U / 8) and deals it round-robin to its
    // workgroups, so that at any time the ~32 workgroups of an XCD sit on ~32 CONSECUTIVE units: the 4 parity classes / the
    // 64-channel blocks of one tile (same input tile) and neighbouring tiles (shared halo) are staged by different CUs at about
    // the same time and come from that XCD's L2 once (rocprofv3: HBM reads = 1.2x the input tensor).
    const unsigned G = gridDim.x;
    const unsigned nxc = G < (unsigned)kXcds ? G : (unsigned)kXcds;
    const unsigned xcd = blockIdx.x % nxc, slot = blockIdx.x / nxc;
    const unsigned nx = G / nxc + (xcd < G % nxc ? 1u : 0u);   // workgroups on this XCD
    // (32-bit: nunits < 2^28, conv_ring_try; through uni(): both roles use these, they belong in scalar registers)
    const unsigned c_begin = uni(xcd * p.nunits / nxc);
    const unsigned c_end = uni((xcd + 1) * p.nunits / nxc);
    if (c_begin + slot >= c_end) return;
    const unsigned u_begin = c_begin + slot, u_end = c_end, u_step = nx;
    const unsigned my_units = (u_end - u_begin + u_step - 1) / u_step;
    const int ngroups = C::NPLANES * p.gpp;   // K groups of one unit
    const unsigned total = my_units * (unsigned)ngroups;

    // Both roles walk the same sequence of groups s = 0 .. total - 1 and meet at one barrier per group:
    //   B_s : group s has landed in buffer s % R (every loader waited for its own pieces first) and every matrix wave is done
    //         reading group s - 1, whose buffer the loaders now refill with group s + R - 1.
    // The epilogue of a unit uses no LDS (it stores straight from the accumulators), so nothing else needs ordering.
    if (wv >= C::MWAVES) {
        // =========================================================================================== loader waves
        const int lw = wv - C::MWAVES;
        // item `it` of this lane = 16-byte slot (it * 4 + lw) * 64 + lane of a group image; the wave-instruction (it * 4 + lw) is
        // wave-uniformly an input piece, a weight piece or filler.  A row (pixel / weight row) is 64 bytes = 32 channels = ONE
        // 64-byte L2 request, 4 lanes; its four 16-byte slots are permuted by XOR with bits 2..3 of the pixel's x (of the row
        // index): applied to the SOURCE offset here and to the ds_read_b128 address in the matrix phase.
        int ia[C::NL], ib[C::NL];   // input: (tn << 20 | ly << 10 | lx), c16  /  weights: tap slot, nn * kpad * 2 + c16  /  filler: -1
#pragma unroll
        for (int it = 0; it < C::NL; ++it) {
            const int wi = it * C::LWAVES + lw;
            const int j = wi * 64 + lane;
            if (wi < C::IN_WI) {
                const int q = j / C::SPP, sp = j % C::SPP;
                const int lx = q % C::IW, ly = (q / C::IW) % C::IH, tn = q / (C::IW * C::IH);
                ia[it] = j < C::IN_SLOTS ? (tn << 20 | ly << 10 | lx) : -1;
                ib[it] = (sp ^ ((lx >> 2) & 3)) * 16;
            } else if (wi < C::IN_WI + C::W_WI) {
                const int jj = j - C::IN_WI * 64;
                const int row = jj / C::SPP, sp = jj % C::SPP;
                const int t = row >> 6, rr = row & 63;
                // LDS row rr = nt * 32 + i feeds row i of matrix tile nt, whose results land in register r = 4 (i / 8) + i % 4 of
                // lane half (i / 4) % 2 (v_mfma_f32_32x32x16 result layout).  Row rr holds output channel
                // 16 q + 8 * half + r % 8 with q = 2 nt + r / 8, so that a lane ends up with four groups of 8 consecutive channels of its
                // pixel in acc[mt][0..1][0..15] and the epilogue stores straight from the accumulators (16-byte stores, no
                // transposition through LDS)
                const int ri = rr & 31;
                const int rreg = (ri >> 3) * 4 + (ri & 3), rhalf = (ri >> 2) & 1, rnt = rr >> 5;
                // 8-channel groups alternate between the lane halves (channel = 16 q + 8 half + k, q = 2 nt + r / 8), so that the
                // two halves' 16-byte stores of one instruction form whole 32-byte sectors (halves holding 32 consecutive
                // channels each, i.e. two half-written sectors per pixel and instruction, measured 8 % slower on the 64 -> 64
                // layer at 256 x 256)
                const int nn = (rnt * 2 + (rreg >> 3)) * 16 + rhalf * 8 + (rreg & 7);
                ia[it] = t;
                ib[it] = nn * p.kpad * 2 + (sp ^ ((rr >> 2) & 3)) * 16;
            } else {
                ia[it] = -1, ib[it] = 0;
            }
        }
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w_bf), 0, (int)p.w_bytes, 0x00020000);
        // The groups are walked by NESTED LOOPS (unit, plane, source, 32-channel group) -- round 5.  (Rounds 2-4 kept a cursor
        // (unit, plane, source, channel) that a stage() call advanced: hipcc turned its selects and nested ifs into ~900 lines of
        // branchy scalar code per group, and a loader wave spent ~0.6 us per group on that bookkeeping before its first DMA piece
        // (timing-only ablation "nothing but the loops": 163-193 us of the 440-710 us of the 2x2-tap kinds) -- for the stride-2 kinds,
        // whose matrix phase is 0.7-0.8 us per group, the loaders were the critical path: matrix waves 35-47 % of their life at B.)
        // Everything that depends on (unit, plane, source) -- descriptor, halo offset, per-lane offsets cv[] -- is set up once per
        // source; the innermost loop over a source's groups only bumps two scalar offsets by 64 bytes.  Protocol unchanged: group g
        // is issued behind barrier B_(g - R + 1); the loop's first R - 1 groups go out at once, R - 1 barriers remain at the end.
        unsigned loc[C::NL], cv[C::NL];   // cached per-lane source offsets
        int loc_ld = -1;
#pragma unroll
        for (int it = 0; it < C::NL; ++it) loc[it] = kRingOob, cv[it] = kRingOob;
        unsigned issued = 0;   // groups staged so far
        unsigned dslot = 0;    // ring slot of the next group
        const bool mix_in = C::MIX_IT >= 0 && C::MIX_IT * C::LWAVES + lw < C::IN_WI;   // the straddling wave-instruction: input piece on this wave?
        if constexpr (C::WRES) {
            // the resident weight rows, once: TAPS x 64 rows in the order and slot permutation of a group's weight rows (see the table above); they are
            // OLDER than every piece of group 0, so the counted wait in front of B_0 covers them
            unsigned keep0;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0" : "=s"(keep0));
#pragma unroll
            for (int i = 0; i < C::WRES_WI / C::LWAVES; ++i) {
                const int wi = i * C::LWAVES + lw;
                const int jj = wi * 64 + lane;
                const int row = jj / C::SPP, sp = jj % C::SPP;
                const int t = row >> 6, rr = row & 63;
                const int ri = rr & 31;
                const int rreg = (ri >> 3) * 4 + (ri & 3), rhalf = (ri >> 2) & 1, rnt = rr >> 5;
                const int nn = (rnt * 2 + (rreg >> 3)) * 16 + rhalf * 8 + (rreg & 7);
                const unsigned voff = (unsigned)t * (unsigned)(p.npad * p.kpad * 2) + (unsigned)(nn * p.kpad * 2 + (sp ^ ((rr >> 2) & 3)) * 16);
                ring_dma16_m0(uni((unsigned)(C::W_RES_OFF + wi * 1024)), voff, rsrc_w, 0u);
            }
            asm volatile("s_mov_b32 m0, %0" ::"s"(keep0));
        }
        for (unsigned pu = u_begin; pu < u_end; pu += u_step) {
            const RingUnit PU = ring_unit(p, pu);
            const bool dry = (p.ablate & 1) && pu != u_begin;     // TIMING ONLY: the pieces fetch nothing
            const bool no_w = C::SKIP_FILL && (p.ablate & 8) && pu != u_begin;   // TIMING ONLY: what resident weights would save
            for (int pplane = 0; pplane < C::NPLANES; ++pplane) {
                const int a = pplane >> 1, b = pplane & 1;
                int oy, ox;   // view coordinates of the halo's first pixel
                if constexpr (C::MODE == RM_K3S1) oy = PU.y0 * C::TH - 1, ox = PU.x0 * C::TW - 1;
                else if constexpr (C::MODE == RM_K5) oy = PU.y0 * C::TH - 2, ox = PU.x0 * C::TW - 2;
                else if constexpr (C::MODE == RM_CT4) oy = PU.y0 * C::TH - (1 - PU.py), ox = PU.x0 * C::TW - (1 - PU.px);
                else if constexpr (C::MODE == RM_SP3) oy = PU.y0 * C::TH, ox = PU.x0 * C::TW;
                else oy = PU.y0 * C::TH - a, ox = PU.x0 * C::TW - b;
                unsigned d_sw = uni((unsigned)((size_t)PU.co0 * p.kpad * 2));   // the unit's weight rows; + 64 bytes per group of the plane
                const unsigned wplane = (unsigned)(p.npad * p.kpad * 2);
                for (int ps = 0; ps < p.nsrc; ++ps) {
                    const int ld = sel4(p.src_ld, ps), srcc = sel4(p.src_c, ps);
                    const size_t img1 = (size_t)p.H * p.W * ld * 2;   // bytes of one sample
                    const size_t img = img1 * C::TN;                   // ... of the samples of a tile (N % TN == 0: conv_ring_try)
                    const char *base_in = uni(static_cast<const char *>(sel4(p.src_ptr, ps)) + (size_t)PU.n0 * img1);
                    const __amdgpu_buffer_rsrc_t d_rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base_in), 0, (int)uni((unsigned)img), 0x00020000);
                    const unsigned ldb = (unsigned)ld * 2u;
                    // descriptor of the straddling wave-instruction (wave-uniform selects)
                    const __amdgpu_buffer_rsrc_t d_rmix = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char *>(uni(mix_in ? base_in : static_cast<const char *>(p.w_bf))), 0, (int)uni(mix_in ? (unsigned)img : (unsigned)p.w_bytes),
                        0x00020000);
                    // Per-lane source offsets: every vector instruction a loader wave issues takes matrix-pipe cycles from its SIMD (~5 per
                    // instruction beside bf16 / fp32 matrix instructions: conv_first.hip, tools/probes/mfma_f32_probe.hip), and the 32-bit
                    // multiplies of the address arithmetic are quarter rate.  Offsets of a halo that lies inside the image are ONE add:
                    // loc[] (the lane's offset from the halo's first pixel, rebuilt only when the source's row stride changes) + the
                    // halo's scalar offset.
                    constexpr int S = C::NPLANES == 4 ? 2 : 1;
                    if (ld != loc_ld) {
                        loc_ld = ld;
#pragma unroll
                        for (int it = 0; it < C::NL; ++it) {
                            if (it * C::LWAVES >= C::IN_WI) continue;
                            const int tn = ia[it] >> 20, ly = (ia[it] >> 10) & 0x3ff, lx = ia[it] & 0x3ff;
                            loc[it] = ia[it] >= 0 ? (unsigned)((tn * p.H + S * ly) * p.W + S * lx) * ldb + (unsigned)ib[it] : kRingOob;
                        }
                    }
                    const int fy = S * oy + (S == 2 ? a : 0), fx = S * ox + (S == 2 ? b : 0);   // image coordinates of the halo's first pixel
                    const unsigned s_halo = uni((unsigned)((fy * p.W + fx) * (int)ldb));
                    const bool interior = fy >= 0 && fy + S * (C::IH - 1) < p.H && fx >= 0 && fx + S * (C::IW - 1) < p.W;   // scalar
#pragma unroll
                    for (int it = 0; it < C::NL; ++it) {
                        const bool in_ct = (it + 1) * C::LWAVES <= C::IN_WI, w_ct = it * C::LWAVES >= C::IN_WI;
                        unsigned v_in = kRingOob, v_w = kRingOob;
                        if (!w_ct) {
                            v_in = loc[it] + s_halo;   // (a filler lane stays out of range: ~2^31 + an offset inside one tile's samples)
                            if (!interior) {
                                const int ly = (ia[it] >> 10) & 0x3ff, lx = ia[it] & 0x3ff;
                                const int ry = fy + S * ly, rx = fx + S * lx;
                                v_in = (ia[it] >= 0 && ry >= 0 && ry < p.H && rx >= 0 && rx < p.W) ? v_in : kRingOob;
                            }
                        }
                        if (!in_ct) {
                            const int t = ia[it];
                            int wt = t;   // plane of the packed weights this tap slot reads
                            bool ok = t >= 0;
                            const int ty = t >> 1, tx = t & 1;
                            if constexpr (C::MODE == RM_CT4 || C::MODE == RM_SP3) wt = (PU.py * 2 + PU.px) * 4 + t;
                            if constexpr (C::MODE == RM_SP3) ok = ok && ty <= PU.py && tx <= PU.px;
                            if constexpr (C::MODE == RM_K3S2) wt = (a ? 2 * ty : 1) * 3 + (b ? 2 * tx : 1), ok = ok && ty <= a && tx <= b;
                            if constexpr (C::MODE == RM_K4S2) wt = (2 * ty + 1 - a) * 4 + (2 * tx + 1 - b);
                            v_w = ok ? (unsigned)wt * wplane + (unsigned)ib[it] : kRingOob;
                        }
                        cv[it] = dry ? kRingOob : (in_ct ? v_in : (w_ct ? v_w : (mix_in ? v_in : v_w)));
                    }
                    unsigned d_sin = 0;   // byte offset of the group's 32 channels inside a pixel of this source
                    for (int pc0 = 0; pc0 < srcc; pc0 += C::CKG) {
                        if (issued >= (unsigned)(C::R - 1)) {
                            ring_wait_vmcnt<(C::R - 2) * C::NL>();   // this wave's pieces of group issued - R + 1 have landed (R > 2: younger groups may still fly)
                            __builtin_amdgcn_s_barrier();            // B_(issued - R + 1): ... and nobody reads the slot this group goes to any more
                        }
                        const unsigned d_base = uni(dslot * (unsigned)C::GROUP_BYTES + (unsigned)(lw * 1024));
                        const unsigned d_smix = mix_in ? d_sin : d_sw;
                        unsigned keep;
                        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0" : "=s"(keep));   // (the s_nop covers descriptor / offset SGPRs fresh from a v_readfirstlane)
#pragma unroll
                        for (int it = 0; it < C::NL; ++it) {
                            // kind of this wave-instruction: compile-time except for the one `it` that straddles the input / weight boundary
                            const bool in_ct = (it + 1) * C::LWAVES <= C::IN_WI, w_ct = it * C::LWAVES >= C::IN_WI;
                            const unsigned dst = d_base + (unsigned)(it * C::LWAVES * 1024);
                            if (C::SKIP_FILL && it * C::LWAVES + lw >= C::IN_WI + C::W_WI) continue;   // filler (wave-uniform)
                            if (no_w && w_ct) continue;
                            if (in_ct) ring_dma16_m0(dst, cv[it], d_rin, d_sin);
                            else if (w_ct) ring_dma16_m0(dst, cv[it], rsrc_w, d_sw);
                            else ring_dma16_m0(dst, cv[it], d_rmix, d_smix);
                        }
                        asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
                        d_sin += (unsigned)(C::CKG * 2), d_sw += (unsigned)(C::CKG * 2);
                        ++issued;
                        dslot = dslot + 1 == (unsigned)C::R ? 0u : dslot + 1;
                    }
                }
            }
        }
        // the last min(total, R - 1) barriers: the groups still in flight land one by one (exact counts: no filler groups)
        const unsigned rem = issued < (unsigned)(C::R - 1) ? issued : (unsigned)(C::R - 1);
        for (unsigned j = 0; j < rem; ++j) {
            const unsigned left = rem - 1 - j;   // groups that may still fly behind the one this barrier publishes
            if (left >= 2) ring_wait_vmcnt<(C::R >= 4 ? 2 : 0) * C::NL>();
            else if (left == 1) ring_wait_vmcnt<(C::R >= 3 ? 1 : 0) * C::NL>();
            else ring_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =============================================================================================== matrix waves
    // operand read offsets inside a group image (bytes): lane (l31, hi) reads k-slot 2 ks + hi of its row
    // (the second 16-channel half, ks = 1, is the same address with bit 5 flipped: the slot index is (2 ks + hi) ^ f)
    constexpr int MT = C::MT;
    int a_off[MT][C::KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int dx = 0; dx < C::KS; ++dx) {
            const int m = (wv * MT + mt) * 32 + l31;   // tile pixel of this lane's operand row
            const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
            const int lx = tx + dx;
            a_off[mt][dx] = ((tn * C::IH + ty) * C::IW + lx) * C::ROWB + ((hi ^ ((lx >> 2) & 3)) << 4);
        }
    const int b_off = l31 * C::ROWB + ((hi ^ ((l31 >> 2) & 3)) << 4);   // inside the weight rows (wb below: a group's, or the resident ones)

    f32x16 acc[MT][2];

    // epilogue state.  The matrix instructions run transposed (weights as the A operand, pixels as B): lane (l31, hi) holds, for
    // each of its two pixels (mt), the 32 channels co0 + 32 hi + [0, 32) in acc[mt][0..1][0..15] -- four 16-byte stores per pixel
    // straight from the registers.  (First version: pixels as A, a [16 pixel][64 channel] fp32 round trip through LDS per wave and
    // pass to get 8 consecutive channels per lane; tools/ring_timers.py: that epilogue took 3.5 us per unit = 22-36 % of the
    // launch, bound by its ~700 vector instructions per wave and the LDS round trips, plus a workgroup barrier to free the LDS
    // it used.)
    // data-gradient: store slots whose old value / forward value are requested ahead of their use.  The sub-pixel data gradient
    // of the stride-2 layers on the 256 x 256 maps is a read-modify-write of 537 MB tensors with 1-4 taps of matrix work per
    // group: its pace is the number of 16-byte requests a wave keeps in flight (PWS_RING_PF overrides for A/B builds)
#ifndef PWS_RING_PF
#define PWS_RING_PF (SG ? (C::MODE == RM_K4S2 ? (C::MT == 4 ? 4 : 8) : (C::MODE == RM_SP3 ? 4 : 2)) : (C::MODE == RM_SP3 ? 2 : 1))   // (the sign-bit variants have the registers for more slots in flight)
#endif
    constexpr int PF = PWS_RING_PF;
    // tall waves: half as many waves issue the epilogue's requests, and the 96 operand registers of the matrix phase are free once it is over:
    // the requests of slots PF .. PFR - 1 go out right behind the last matrix instruction, and the rolling distance is PFR
#ifndef PWS_RING_PFR
#define PWS_RING_PFR (C::TALL && MT == 4 ? (SG ? 10 : 6) : PF)
#endif
    constexpr int PFR = PWS_RING_PFR < PF ? PF : PWS_RING_PFR;
    constexpr int NSLOT = MT * 4;  // slot = mt * 4 + q: pixel mt, 8-channel group q of the lane's 32 channels
    // destination of the unit's two 32-channel blocks (block b = channels co0 + 32 b + [0, 32) = store slots q = 2 b, 2 b + 1;
    // data gradient: a block lies in ONE source of the forward layer, sources being multiples of 32 channels): pointers already
    // at this lane's first group (+ 8 hi channels)
    struct EpiBlock {
        unsigned char *d;
        const unsigned char *y;   // the forward tensor (16 bytes per slot) or, with `sign`, its sign bits (1 byte per slot)
        unsigned dld2, yld2;   // pixel strides in bytes
        unsigned yo;           // SG: this lane's byte inside a pixel's sign bytes (y is then the tensor's base: wave-uniform, stays in scalar registers)
        bool ok, acc, hasy;
        float slope;
    } eb[2] = {};
    u32x4 e_old[DG ? NSLOT : 1], e_y[DG && !SG ? NSLOT : 1];
    unsigned e_m[SG ? NSLOT : 1];

    // The epilogue's loads are unconditional in control flow (a lane without the tensor reads 16 zero bytes instead): a load inside
    // an if leaves a control-flow merge behind, and hipcc waits vmcnt(0) at the first use
    // after every merge -- which serialised the rolling requests and the stores of the data-gradient epilogue.
    // (address space 1 spelled out: through a generic pointer these were flat_load instructions, which count on lgkmcnt as well as vmcnt and so sat
    //  in the way of the matrix phase's LDS reads)
    typedef const __attribute__((address_space(1))) u32x4 *gptr16;
    typedef const __attribute__((address_space(1))) unsigned char *gptr1;
    auto e_load = [&](const unsigned char *ptr, bool have) {
        return *(gptr16)(have ? ptr : reinterpret_cast<const unsigned char *>(&g_ring_zero16));
    };
    // act' operand of slot `sl` (channel half sl & 1 of its block) at pixel `pix` into register set `dst`: the 8 bf16 values of the forward
    // tensor, or -- SG -- ONE byte of sign bits (v > 0 is all the epilogue asks of them)
    auto e_load_y = [&](const EpiBlock &e, unsigned pix, int sl, bool have, int dst) {
        const unsigned char *z = reinterpret_cast<const unsigned char *>(&g_ring_zero16);
        if constexpr (SG) e_m[dst] = *(gptr1)(have ? e.y + (size_t)(pix * e.yld2 + e.yo + (unsigned)((sl & 1) * 2)) : z);
        else e_y[dst] = *(gptr16)(have ? e.y + (size_t)pix * e.yld2 + (sl & 1) * 32 : z);
    };
    constexpr int SO = C::NCLS == 4 ? 2 : 1;   // output stride of the parity classes
    unsigned cu = u_begin;
    int cg = 0, cplane = 0, cgp = 0, cbuf = 0;
    RingUnit CU = ring_unit(p, cu);
    // output pixel index of this lane's pixel mt of unit CU: tile pixel m = (2 wv + mt) * 32 + l31
    auto e_pix = [&](int mt) {
        const int m = (wv * MT + mt) * 32 + l31;
        const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
        const int y = CU.y0 * C::TH + ty, x = CU.x0 * C::TW + tx;
        const int oy = SO * y + (C::NCLS == 4 ? CU.py : 0), ox = SO * x + (C::NCLS == 4 ? CU.px : 0);
        return (unsigned)(((CU.n0 + tn) * p.OH + oy) * p.OW + ox);
    };
    // the bias vector lives in LDS behind the ring, staged once: a lane needs 32 different values per unit.  Every matrix wave
    // sees it after the first barrier B of the loop.
    const float *const lds_bias = reinterpret_cast<const float *>(lds + C::BIAS_OFF);
    if constexpr (!DG) {
        float *wb = reinterpret_cast<float *>(lds + C::BIAS_OFF);
        for (int c = tid; c < C::BIAS_FLOATS; c += C::MWAVES * 64) wb[c] = (p.bias && c < p.cout) ? p.bias[c] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the writes have landed before this wave reaches B_0 (hipcc puts no wait in front of a bare s_barrier)
    }

#ifdef PWS_RING_TIMERS   // diagnostic build (tools/ring_timers.sh): where a matrix wave's time goes, in s_memrealtime ticks (10 ns)
    unsigned long long t_wait = 0, t_mat = 0, t_epi = 0, t_pre = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#define PWS_RT() __builtin_amdgcn_s_memrealtime()
#endif
    for (unsigned s = 0; s < total; ++s) {
#ifdef PWS_RING_TIMERS
        const unsigned long long t0 = PWS_RT();
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_s
        asm volatile("" ::: "memory");
#ifdef PWS_RING_TIMERS
        const unsigned long long t1 = PWS_RT();
        t_wait += t1 - t0;
#endif

        const bool last_group = cg == ngroups - 1;
        if (last_group) {
            // ---- where this unit's channels go; data gradient: the old gradient values and the forward tensor (for act') of the
            // first store slot are requested ONE matrix phase ahead of their use
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int co = CU.co0 + 32 * b;
                EpiBlock e{};
                e.slope = 1.f;
                if constexpr (!DG) {
                    e.d = reinterpret_cast<unsigned char *>(reinterpret_cast<__bf16 *>(p.out) + co + hi * 8);
                    e.dld2 = (unsigned)p.out_ld * 2u, e.ok = co < p.cout;
                } else {
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        if (s_ < p.ndst && co >= p.dst_c0[s_] && co < p.dst_c1[s_]) {
                            e.d = reinterpret_cast<unsigned char *>(reinterpret_cast<__bf16 *>(p.dst_ptr[s_]) + (co - p.dst_c0[s_]) + hi * 8);
                            e.dld2 = (unsigned)p.dst_ld[s_] * 2u, e.acc = p.dst_acc[s_] != 0, e.ok = true;
                            if (p.dst_act[s_] != PWS_ACT_NONE) {
                                if constexpr (SG) {
                                    e.y = p.dst_sign[s_], e.yo = (unsigned)(((co - p.dst_c0[s_]) >> 3) + hi), e.yld2 = (unsigned)p.dst_sign_ld[s_];
                                } else {
                                    e.y = reinterpret_cast<const unsigned char *>(static_cast<const __bf16 *>(p.dst_y[s_]) + (co - p.dst_c0[s_]) + hi * 8);
                                    e.yld2 = (unsigned)p.dst_y_ld[s_] * 2u;
                                }
                                e.slope = p.dst_act[s_] == PWS_ACT_LRELU ? 0.2f : 0.f, e.hasy = true;
                            }
                        }
                    }
                }
                eb[b] = e;
            }
            if constexpr (DG) {
#pragma unroll
                for (int slot = 0; slot < PF; ++slot) {
                    const unsigned pix = e_pix(slot >> 2);
                    const EpiBlock &e = eb[(slot >> 1) & 1];
                    e_old[slot] = e_load(e.d + (size_t)pix * e.dld2 + (slot & 1) * 32, e.ok && e.acc);
                    e_load_y(e, pix, slot, e.ok && e.hasy, slot);
                }
            }
        }

        if (cg == 0) {
            if constexpr (!DG) {
                // the accumulators start at the BIAS (round 5): acc[mt][nt][r] is channel co0 + 32 nt + 16 (r / 8) + 8 hi + r % 8 of pixel mt, so a
                // lane's 32 bias values are eight 16-byte LDS reads per unit -- instead of two reads and eight adds in each of the 4 MT store
                // slots of the epilogue, which is bound by the instructions ONE wave per SIMD can issue (tools/probes/r5d_epilogue.sh: the
                // epilogue takes 3.5 us per unit with or without its stores)
                float bz[2][16];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 b0 = *reinterpret_cast<const float4 *>(lds_bias + CU.co0 + nt * 32 + h * 16 + hi * 8);
                        const float4 b1 = *reinterpret_cast<const float4 *>(lds_bias + CU.co0 + nt * 32 + h * 16 + hi * 8 + 4);
                        bz[nt][h * 8 + 0] = b0.x, bz[nt][h * 8 + 1] = b0.y, bz[nt][h * 8 + 2] = b0.z, bz[nt][h * 8 + 3] = b0.w;
                        bz[nt][h * 8 + 4] = b1.x, bz[nt][h * 8 + 5] = b1.y, bz[nt][h * 8 + 6] = b1.z, bz[nt][h * 8 + 7] = b1.w;
                    }
                // (one opaque v_mov per element: with plain copies -- element by element or acc[mt] = acc[0] -- hipcc gave the loop-carried
                //  accumulators other registers than the initialised ones and copied 96 registers in EVERY group: +6 ... +13 % on the tall kernels)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm("v_mov_b32 %0, %1" : "=v"(acc[mt][nt][r]) : "v"(bz[nt][r]));   // (opaque: a plain copy is not coalesced with the loop-carried registers)
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
        }
#ifdef PWS_RING_TIMERS
        const unsigned long long t2 = PWS_RT();
        t_pre += t2 - t1;
#endif
        unsigned tapmask = (1u << C::TAPS) - 1u;
        if constexpr (C::MODE == RM_SP3) tapmask = CU.py ? (CU.px ? 0xfu : 0x5u) : (CU.px ? 0x3u : 0x1u);
        if constexpr (C::MODE == RM_K3S2) tapmask = (cplane >> 1) ? ((cplane & 1) ? 0xfu : 0x5u) : ((cplane & 1) ? 0x3u : 0x1u);
        const unsigned gb = (unsigned)(cbuf * C::GROUP_BYTES);
        const unsigned wb = C::WRES ? (unsigned)C::W_RES_OFF : gb + (unsigned)C::W_OFF;
        if (!(p.ablate & 2)) {
            if constexpr (C::TALL) {
                // tall waves: step = (channel half ks, tap column tx).  The wave's 4 tile rows see MT + KS - 1 halo rows through the KS vertical
                // taps: each is read once per step; the operands of step + 1 are requested under the matrix instructions of the step, one
                // ds_read_b128 after every second matrix instruction (one wave per SIMD: nobody else fills a bubble in the matrix pipe)
                static_assert(C::MODE == RM_K3S1 || C::MODE == RM_CT4 || C::MODE == RM_K4S2 || C::MODE == RM_K5, "tall waves: the kinds without a tap mask");
                constexpr int NST = 2 * C::KS, NA = MT + C::KS - 1, NRD = NA + 2 * C::KS, NMM = C::KS * MT * 2;
                bf16x8 av[2][NA], bv[2][C::KS][2];
                auto rd = [&](int slot, int st) {
                    const int ks = st / C::KS, tx = st % C::KS;
#pragma unroll
                    for (int j = 0; j < NA; ++j)
                        av[slot][j] = *reinterpret_cast<const bf16x8 *>(lds + gb + (unsigned)((a_off[0][tx] ^ (ks * 32)) + j * C::IW * C::ROWB));
#pragma unroll
                    for (int ty = 0; ty < C::KS; ++ty)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            bv[slot][ty][nt] = *reinterpret_cast<const bf16x8 *>(lds + wb + (unsigned)((b_off ^ (ks * 32)) + ((ty * C::KS + tx) * 64 + nt * 32) * C::ROWB));
                };
                rd(0, 0);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int cur = st & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < NST) rd(cur ^ 1, st + 1);
#pragma unroll
                    for (int ty = 0; ty < C::KS; ++ty)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[cur][ty][nt], av[cur][mt + ty], acc[mt][nt], 0, 0, 0);
                    if (st + 1 < NST) {
                        if constexpr (NRD > NMM / 2) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMM / 2, 0);
#pragma unroll
                        for (int i = 0; i < (NRD < NMM / 2 ? NRD : NMM / 2); ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (C::MODE != RM_K3S1) {
                // sub-pixel / stride-2 kinds take 1, 2 or 4 of the taps by class / plane (wave-uniform mask); the k4s2 kinds measured
                // 1-2 % slower with the pipelined loop below: plain loop
#pragma unroll
                for (int tap = 0; tap < C::TAPS; ++tap) {
                    if (!((tapmask >> tap) & 1u)) continue;
                    const int ty = tap / C::KS, tx = tap % C::KS;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        bf16x8 av[2], bv[2];
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            av[mt] = *reinterpret_cast<const bf16x8 *>(lds + gb + (unsigned)((a_off[mt][tx] ^ (ks * 32)) + ty * C::IW * C::ROWB));
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            bv[nt] = *reinterpret_cast<const bf16x8 *>(lds + wb + (unsigned)((b_off ^ (ks * 32)) + (tap * 64 + nt * 32) * C::ROWB));
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[nt], av[mt], acc[mt][nt], 0, 0, 0);
                    }
                }
            } else {
                // 3x3 stride 1 (+4 to +7 % measured): the operands of step (tap, ks) + 2 are requested before the four matrix instructions of step (tap, ks) are
                // issued, fenced by sched_barrier -- left to itself hipcc sinks every ds_read_b128 to its first use and waits for it
                // there (49 s_waitcnt for 72 matrix instructions: every group of four paid an LDS round trip)
                constexpr int NS = C::TAPS * 2, PD = DG ? 1 : 2;   // the data-gradient epilogue needs the registers
                bf16x8 av[PD + 1][2], bv[PD + 1][2];
                auto rd = [&](int slot, int st) {
                    const int tap = st >> 1, ks = st & 1, ty = tap / C::KS, tx = tap % C::KS;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        av[slot][mt] = *reinterpret_cast<const bf16x8 *>(lds + gb + (unsigned)((a_off[mt][tx] ^ (ks * 32)) + ty * C::IW * C::ROWB));
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        bv[slot][nt] = *reinterpret_cast<const bf16x8 *>(lds + wb + (unsigned)((b_off ^ (ks * 32)) + (tap * 64 + nt * 32) * C::ROWB));
                };
#pragma unroll
                for (int st = 0; st < PD; ++st) rd(st, st);
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    if (st + PD < NS) rd((st + PD) % (PD + 1), st + PD);
                    __builtin_amdgcn_sched_barrier(0);
                    const int cur = st % (PD + 1);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[cur][nt], av[cur][mt], acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

#ifdef PWS_RING_TIMERS
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t3 = PWS_RT();
        t_mat += t3 - t2;
#endif
        if (last_group) {
            // ---- epilogue of unit cu, straight from the accumulators: slot (mt, q) = channels co0 + 32 hi + 8 q + [0, 8) of pixel mt
            // = acc[mt][q / 2][8 (q % 2) + k].  No LDS, no barrier: the waves run into the next unit's first group on their own.
            if (!(p.ablate & 4)) {
                if constexpr (DG && PFR > PF) {
#pragma unroll
                    for (int slot = PF; slot < PFR && slot < NSLOT; ++slot) {
                        const unsigned pix2 = e_pix(slot >> 2);
                        const EpiBlock &e2 = eb[(slot >> 1) & 1];
                        e_old[slot] = e_load(e2.d + (size_t)pix2 * e2.dld2 + (slot & 1) * 32, e2.ok && e2.acc);
                        e_load_y(e2, pix2, slot, e2.ok && e2.hasy && !(p.ablate & 16), slot);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned pix = e_pix(mt);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int slot = mt * 4 + q;
                        const EpiBlock &e = eb[q >> 1];
                        if constexpr (DG) {   // rolling request, PF slots ahead
                            const int nx2 = slot + PFR;
                            if (nx2 < NSLOT) {
                                const unsigned pix2 = e_pix(nx2 >> 2);
                                const EpiBlock &e2 = eb[(nx2 >> 1) & 1];
                                e_old[nx2] = e_load(e2.d + (size_t)pix2 * e2.dld2 + (nx2 & 1) * 32, e2.ok && e2.acc);
                                e_load_y(e2, pix2, nx2, e2.ok && e2.hasy && !(p.ablate & 16), nx2);
                            }
                        }
                        // straight-line on purpose: only the store itself is predicated (a block that short gets no skip branch), so
                        // hipcc counts the outstanding loads and stores exactly instead of waiting vmcnt(0) at every control-flow merge
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = acc[mt][q >> 1][(q & 1) * 8 + k];
                        if constexpr (!DG) {
                            // (the bias is in the accumulators already.)  LReLU / none: max(v, slope v) -- the multiplies pack two to an
                            // instruction; ReLU: max(v, +0) (slope v would leave -0 behind).  One scalar branch per slot.
                            if (p.act == PWS_ACT_RELU) {
#pragma unroll
                                for (int k = 0; k < 8; ++k) asm("v_max_f32 %0, 0, %1" : "=v"(v[k]) : "v"(v[k]));
                            } else {
                                const float slope = p.act == PWS_ACT_LRELU ? 0.2f : 1.f;
#pragma unroll
                                for (int k = 0; k < 8; k += 2) {
                                    const f32x2 t = (f32x2){v[k], v[k + 1]} * (f32x2){slope, slope};   // v_pk_mul_f32
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[k]) : "v"(v[k]), "v"(t.x));   // (the builtin would canonicalise first)
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[k + 1]) : "v"(v[k + 1]), "v"(t.y));
                                }
                            }
                        } else {
                            const u32x4 o = e_old[slot];   // zeros when this destination is overwritten
                            {   // (two values per instruction: v_pk_add_f32 -- written out, hipcc does not pair them across the asm statements below)
                                const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const f32x2 t = (f32x2){v[2 * k], v[2 * k + 1]} + (f32x2){bf16_lo(ow[k]), bf16_hi(ow[k])};
                                    v[2 * k] = t.x, v[2 * k + 1] = t.y;
                                }
                            }
                            // act'(y) of the tensor this destination is the gradient of (no such tensor: slope == 1)
                            const float sl = e.slope;
                            if constexpr (SG) {
                                // bit k of the mask set: v stays, else slope v -- as a bit select (v_bfe_i32 makes 0 / ~0 of the bit, v_bfi_b32
                                // picks v or slope v): 2.5 instructions per value instead of 4 (bit test, compare, select, multiply)
                                const unsigned m = e_m[slot];
#pragma unroll
                                for (int k = 0; k < 8; k += 2) {
                                    const f32x2 vs = (f32x2){v[k], v[k + 1]} * (f32x2){sl, sl};   // v_pk_mul_f32
                                    const unsigned keep0 = (unsigned)__builtin_amdgcn_sbfe((int)m, (unsigned)k, 1u), keep1 = (unsigned)__builtin_amdgcn_sbfe((int)m, (unsigned)(k + 1), 1u);
                                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(keep0), "v"(v[k]), "v"(vs.x));   // (keep & v) | (~keep & slope v); hipcc's own lowering of that expression is six instructions
                                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[k + 1]) : "v"(keep1), "v"(v[k + 1]), "v"(vs.y));
                                }
                            } else {
                                const u32x4 yv = e_y[slot];
                                v[0] *= bf16_lo(yv.x) > 0.f ? 1.f : sl, v[1] *= bf16_hi(yv.x) > 0.f ? 1.f : sl;
                                v[2] *= bf16_lo(yv.y) > 0.f ? 1.f : sl, v[3] *= bf16_hi(yv.y) > 0.f ? 1.f : sl;
                                v[4] *= bf16_lo(yv.z) > 0.f ? 1.f : sl, v[5] *= bf16_hi(yv.z) > 0.f ? 1.f : sl;
                                v[6] *= bf16_lo(yv.w) > 0.f ? 1.f : sl, v[7] *= bf16_hi(yv.w) > 0.f ? 1.f : sl;
                            }
                        }
                        u32x4 wq;
                        wq.x = cvt_pk_bf16(v[0], v[1]), wq.y = cvt_pk_bf16(v[2], v[3]);
                        wq.z = cvt_pk_bf16(v[4], v[5]), wq.w = cvt_pk_bf16(v[6], v[7]);
                        unsigned char *dptr = e.d + (size_t)pix * e.dld2 + (q & 1) * 32;
                        asm volatile("" ::"v"(wq.x), "v"(wq.y), "v"(wq.z), "v"(wq.w), "v"(dptr));   // nothing sinks into the if
                        const bool st_ok = DG ? e.ok : CU.co0 + q * 16 + hi * 8 < p.cout;   // (cout % 8 == 0: epi16)
#if defined(PWS_RING_TIMERS) && PWS_RING_TIMERS == 2   // diagnostic: no stores
                        if (st_ok && wq.x == 0x12345678u && wq.y == 0x9abcdef0u) *reinterpret_cast<u32x4 *>(dptr) = wq;
#else
                        if (st_ok) *reinterpret_cast<u32x4 *>(dptr) = wq;
#endif
                        if constexpr (!DG) {
                            if (p.out_sign) {   // (uniform) sign bits of the 8 rounded values: what a later act' needs of this tensor
                                // a bf16 is > 0 exactly when its bit pattern, read as a SIGNED 16-bit integer, is > 0 (a NaN with a clear sign bit
                                // counts as positive here; the float compare of rounds 2-4 said no): min(max(x, 0), 1) on both halves of a word
                                // at once (v_pk_max_i16 / v_pk_min_i16) leaves the two answers at bits 0 and 16 -- 13 instructions per slot
                                // instead of 35 (8 x unpack, compare, select, or)
                                const unsigned one2 = 0x00010001u;
                                auto pos2 = [&](unsigned w) {   // (asm: hipcc turns the elementwise min / max into two compares, two selects and a permute)
                                    unsigned r;
                                    asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_i16 %0, %0, %2" : "=v"(r) : "v"(w), "s"(one2));
                                    return r;
                                };
                                const unsigned x = pos2(wq.x) | (pos2(wq.y) << 2) | (pos2(wq.z) << 4) | (pos2(wq.w) << 6);   // values 0, 2, 4, 6 at bits 0, 2, 4, 6; 1, 3, 5, 7 at 16, 18, 20, 22
                                if (st_ok) p.out_sign[(size_t)pix * (unsigned)p.out_sign_ld + (unsigned)((CU.co0 + q * 16) >> 3) + (unsigned)hi] = (unsigned char)(x | (x >> 15));
                            }
                        }
                    }
                }
            }
            cg = 0, cplane = 0, cgp = 0, cu += u_step;
            if (cu < u_end) CU = ring_unit(p, cu);
#ifdef PWS_RING_TIMERS
            t_epi += PWS_RT() - t3;
#endif
        } else {
            ++cg;
            if (++cgp == p.gpp) cgp = 0, ++cplane;
        }
        cbuf = cbuf + 1 == C::R ? 0 : cbuf + 1;
    }
#ifdef PWS_RING_TIMERS
    if (lane == 0 && p.timers) {
        unsigned long long *t = p.timers + ((size_t)blockIdx.x * C::MWAVES + wv) * 8;
        t[0] = t_wait, t[1] = t_pre, t[2] = t_mat, t[3] = t_epi, t[4] = PWS_RT() - t_begin, t[5] = total, t[6] = my_units;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host side
template <class C, bool DG, bool SG = false>
static int ring_launch(RingParams &rp, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ring_kernel<C, DG, SG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_ring_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
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
    rp.tiles_x = rp.LW / C::TW, rp.tiles_y = rp.LH / C::TH, rp.tn = C::TN;   // whole tiles only (conv_ring_try)
    rp.ncob = (unsigned)((rp.cout + 63) / 64), rp.ncls = (unsigned)C::NCLS;
    rp.nunits = (unsigned)(rp.tiles_x * rp.tiles_y) * (unsigned)(rp.N / C::TN) * rp.ncob * rp.ncls;
    {
        auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); };
        unsigned dmax = rp.ncls > rp.ncob ? rp.ncls : rp.ncob;
        dmax = dmax > (unsigned)rp.tiles_x ? dmax : (unsigned)rp.tiles_x, dmax = dmax > (unsigned)rp.tiles_y ? dmax : (unsigned)rp.tiles_y;
        if ((unsigned long long)rp.nunits * dmax >= 0x100000000ull) return 1;   // (never at this generator's sizes) the multiply-high division would not be exact
        rp.m_cls = magic(rp.ncls), rp.m_cob = magic(rp.ncob), rp.m_tx = magic((unsigned)rp.tiles_x), rp.m_ty = magic((unsigned)rp.tiles_y);
    }
    int cin = 0;
    for (int s = 0; s < rp.nsrc; ++s) cin += rp.src_c[s];
    rp.gpp = cin / C::CKG;
    const unsigned grid = rp.nunits < (unsigned)ncu ? rp.nunits : (unsigned)ncu;   // one persistent workgroup per CU
    hipLaunchKernelGGL((conv_ring_kernel<C, DG, SG>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, st, rp);
    return check_launch("conv_ring_kernel");
}

// Tile shapes.  Units of 512 pixels: 16 x 32 pixels of one sample for maps at least 32 wide, 16 x 16 x 2 samples for 16-wide maps, 8 x 8 x 8 samples for
// 8 x 8 maps (2x2-tap kinds only: the 3x3 kind's group of 8 halo'd samples does not fit two ring buffers).  Units of 256 pixels (round 6): 16 x 16 of
// one sample, 8 x 8 x 4 samples; of 128 pixels: 8 x 8 x 2 samples, 4 x 4 x 8 samples.
template <int MODE, bool DG, bool SG = false>
static int ring_launch_tile(int pix, int tw, RingParams &rp, hipStream_t st) {
    constexpr int RS = MODE == RM_K3S1 ? 2 : 3;   // ring depth of the small units (the 3x3 kind's 36 KB of weights per group leave room for two)
    if (pix == 256) {
        using C16 = RgCfg<MODE, 16, 16, 1, RS>;
        using C8 = RgCfg<MODE, 8, 8, 4, RS>;
        static_assert(C16::BIAS_FLOATS == 1024 && C8::BIAS_FLOATS == 1024, "conv_ring_try's bias-slot check");
        if (tw == 16) return ring_launch<C16, DG, SG>(rp, st);
        if (tw == 8) return ring_launch<C8, DG, SG>(rp, st);
        return 1;
    }
    if (pix == 128) {
        using C8 = RgCfg<MODE, 8, 8, 2, RS>;
        using C4 = RgCfg<MODE, 4, 4, 8, RS>;
        static_assert(C8::BIAS_FLOATS == 1024 && C4::BIAS_FLOATS == 1024, "conv_ring_try's bias-slot check");
        if (tw == 8) return ring_launch<C8, DG, SG>(rp, st);
        if (tw == 4) return ring_launch<C4, DG, SG>(rp, st);
        return 1;
    }
    if constexpr (MODE == RM_K3S1 || MODE == RM_CT4 || MODE == RM_K4S2)
        if (tw == 32 && g_experiment != 105 && !(g_experiment == 106 && DG) && !(g_experiment == 107 && !DG)) return ring_launch<RgCfg<MODE, 16, 32, 1, MODE == RM_K3S1 ? 2 : 3, 4>, DG, SG>(rp, st);   // (A/B: 105 = the 8-wave kernel, 106 / 107 = for the data gradients / the forwards only)
    if (tw == 32) return ring_launch<RgCfg<MODE, 16, 32, 1, MODE == RM_K3S1 ? 2 : 3>, DG, SG>(rp, st);
    if (tw == 16) return ring_launch<RgCfg<MODE, 16, 16, 2, 2>, DG, SG>(rp, st);
    if constexpr (MODE != RM_K3S1) return ring_launch<RgCfg<MODE, 8, 8, 8, 2>, DG, SG>(rp, st);
    return 1;
}

// Runs the launch described by kp (prepared by conv2d_fwd_impl / conv2d_bwd_data_impl, conv_mfma.hip) on the ring kernel when it
// is covered: bf16 storage with 16-byte epilogue stores, every source a multiple of 32 channels, a logical map of at least
// 16 x 32 pixels and enough units to give every CU one.  Returns 1 when not covered (the caller runs conv_bf16_kernel).
int conv_ring_try(int kind, bool dgrad, const ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    if (!kp.io_bf16 || !kp.epi16 || g_experiment == 20) return 1;
#ifdef PWS_INTERFERENCE_PROBE
    if (g_experiment >= 2100 && g_experiment < 2140) return 1;   // (2100 ..: the probe variants of conv_bf16_kernel)
#endif
    for (int s = 0; s < kp.nsrc; ++s)
        if (kp.src_c[s] % 32 != 0 || kp.src_ld[s] % 8 != 0 || (reinterpret_cast<size_t>(kp.src_ptr[s]) & 15)) return 1;
    int mode;
    int planes;
    if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) mode = RM_K3S1, planes = 9;
    else if (kind == PWS_CONV_K3S2) mode = dgrad ? RM_SP3 : RM_K3S2, planes = dgrad ? 16 : 9;
    else if (kind == PWS_CONVT_K4S2) mode = dgrad ? RM_K4S2 : RM_CT4, planes = 16;
    else if (kind == PWS_CONV_K5S1 && !dgrad) mode = RM_K5, planes = 25;
    else return 1;
    // the first layer (round 6): ONE source of 32 channels (the NHWC-32 copy of the window), ONE 64-channel output block -- what makes its weights resident;
    // 8 x 32 tiles (PWS_OPT_EXPERIMENT 189: never -- conv_bf16_k5_kernel as in rounds 1-5)
    if (mode == RM_K5 && (g_experiment == 189 || kp.nsrc != 1 || kp.src_c[0] != 32 || kp.cout > 64 || kp.kpad_bf != 32 || kp.LW % 32 != 0 || kp.LH % 8 != 0)) return 1;
    const size_t sample_bytes = (size_t)kp.H * kp.W * 2;
    for (int s = 0; s < kp.nsrc; ++s)
        if (sample_bytes * kp.src_ld[s] >= (1u << 31)) return 1;
    RingParams rp{};
    for (int s = 0; s < 4; ++s) {
        rp.src_ptr[s] = kp.src_ptr[s], rp.src_c[s] = kp.src_c[s], rp.src_ld[s] = kp.src_ld[s];
        rp.dst_ptr[s] = kp.dst_ptr[s], rp.dst_c0[s] = kp.dst_c0[s], rp.dst_c1[s] = kp.dst_c1[s], rp.dst_ld[s] = kp.dst_ld[s];
        rp.dst_acc[s] = kp.dst_acc[s], rp.dst_y[s] = kp.dst_y[s], rp.dst_y_ld[s] = kp.dst_y_ld[s], rp.dst_act[s] = kp.dst_act[s];
    }
    rp.nsrc = kp.nsrc, rp.N = kp.N, rp.H = kp.H, rp.W = kp.W, rp.LH = kp.LH, rp.LW = kp.LW, rp.OH = kp.OH, rp.OW = kp.OW;
    rp.cout = kp.cout, rp.kpad = kp.kpad_bf, rp.npad = kp.npad_bf, rp.w_bf = kp.w_bf;
    rp.w_bytes = (size_t)planes * kp.npad_bf * kp.kpad_bf * 2;
    if (rp.w_bytes >= (1u << 31)) return 1;
    rp.bias = kp.bias, rp.act = kp.act, rp.out = kp.out, rp.out_ld = kp.out_ld, rp.ndst = kp.ndst;
    rp.out_sign = dgrad ? nullptr : static_cast<unsigned char *>(kp.out_sign), rp.out_sign_ld = kp.out_sign_ld;
    for (int s = 0; s < 4; ++s) rp.dst_sign[s] = static_cast<const unsigned char *>(kp.dst_sign[s]), rp.dst_sign_ld[s] = kp.dst_sign_ld[s];
#ifdef PWS_RING_TIMERS
    {   // tools/ring_timers.py passes the buffer's address through the environment
        const char *e = getenv("PWS_RING_TIMERS_PTR");
        rp.timers = e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0)) : nullptr;
    }
#endif
    rp.ablate = g_experiment >= 41 && g_experiment <= 48 ? g_experiment - 40 : (g_experiment == 49 ? 16 : 0);   // (49 = mask 16: no act' loads)   // (48 = mask 8: weight pieces only for a workgroup's first unit)
    // Unit size and tile shape: whole tiles only.  512-pixel units when they give every CU (most of) one; else 256, else 128 -- a smaller unit stages
    // the same weight rows for fewer pixels, so it is taken only where the large one would leave the chip idle.  Below `min_small` units of 128 pixels
    // the launch is the split-K / one-shot kernels' (PWS_OPT_EXPERIMENT 181 .. 184: min_small 64 / 96 / 192 / 256; 185: no small units, rounds 2-5; 186 / 187: units of 256 / 128 pixels whatever the count).
    const long per_px = (long)((kp.cout + 63) / 64) * ((mode == RM_CT4 || mode == RM_SP3) ? 4 : 1);
    const long px = (long)kp.LW * kp.LH * kp.N;
    int pix = 0, tw = 0;
    if (mode == RM_K5) {
        pix = 256, tw = 32;
        if (px / 256 < 192 && g_experiment != 21) return 1;
    } else {
        int tw512 = 0;
        if (kp.LW % 32 == 0 && kp.LH % 16 == 0) tw512 = 32;
        else if (kp.LW % 16 == 0 && kp.LH % 16 == 0 && kp.N % 2 == 0) tw512 = 16;
        else if (kp.LW % 8 == 0 && kp.LH % 8 == 0 && kp.N % 8 == 0 && mode != RM_K3S1) tw512 = 8;
        int tw256 = 0;
        if (kp.LW % 16 == 0 && kp.LH % 16 == 0) tw256 = 16;
        else if (kp.LW % 8 == 0 && kp.LH % 8 == 0 && kp.N % 4 == 0) tw256 = 8;
        int tw128 = 0;
        if (kp.LW % 8 == 0 && kp.LH % 8 == 0 && kp.N % 2 == 0) tw128 = 8;
        else if (kp.LW % 4 == 0 && kp.LH % 4 == 0 && kp.N % 8 == 0) tw128 = 4;
        const bool small_ok = g_experiment != 185;
        // (measured, tools/probes/r6c / r6d: batch-8 inference 6471 -> 6600 f/s with 64 instead of 128 -- its alternatives are launch-latency-bound split-K
        //  pairs; the batch-64 training step 25.88 -> 25.94 ms -- there the alternative is a conv_bf16_kernel launch that fills the chip: by batch)
        const long min_dflt = kp.N <= 16 ? 64 : 128;
        const long min_small = g_experiment == 181 ? 64 : (g_experiment == 182 ? 96 : (g_experiment == 183 ? 192 : (g_experiment == 184 ? 256 : min_dflt)));
        const int force_pix = g_experiment == 186 ? 256 : (g_experiment == 187 ? 128 : 0);   // tests: that unit size or nothing
        if (force_pix == 256 && tw256) pix = 256, tw = tw256;
        else if (force_pix == 128 && tw128) pix = 128, tw = tw128;
        else if (force_pix) return 1;
        else if (tw512 && (px / 512 * per_px >= 192 || g_experiment == 21)) pix = 512, tw = tw512;
        else if (small_ok && tw256 && px / 256 * per_px >= 192) pix = 256, tw = tw256;
        else if (small_ok && tw128 && px / 128 * per_px >= min_small) pix = 128, tw = tw128;
        else return 1;   // too few units for 256 persistent workgroups: the split-K kernels do better
    }
    if (kp.cout > (pix == 512 && tw == 16 && mode == RM_K3S1 ? 512 : 1024)) return 1;   // the bias vector's LDS slot (RgCfg::BIAS_FLOATS)
    if (px / pix * per_px >= (1l << 28)) return 1;   // 32-bit unit arithmetic in the kernel
    // the sign-bit variant of the data-gradient epilogue: when EVERY destination with an act' has its sign bits (PWS_OPT_EXPERIMENT 12: never)
    bool sg = dgrad && g_experiment != 12, any_act = false;
    for (int s = 0; s < kp.ndst; ++s)
        if (kp.dst_act[s] != PWS_ACT_NONE) any_act = true, sg = sg && kp.dst_sign[s] != nullptr;
    sg = sg && any_act;
    ProfScope prof(KID_CONV_RING, pi.flops, pi.bytes, st);
    switch (mode) {
    case RM_K5:   // (one tile row per wave on 8 matrix waves -- the epilogue of one wave under the matrix phase of the other -- measured the same: 480-510 against 490 us)
        return ring_launch<RgCfg<RM_K5, 8, 32, 1, 2, 2>, false, false>(rp, st);
    case RM_K3S1: return !dgrad ? ring_launch_tile<RM_K3S1, false>(pix, tw, rp, st) : (sg ? ring_launch_tile<RM_K3S1, true, true>(pix, tw, rp, st) : ring_launch_tile<RM_K3S1, true>(pix, tw, rp, st));
    case RM_CT4: return ring_launch_tile<RM_CT4, false>(pix, tw, rp, st);
    case RM_SP3: return sg ? ring_launch_tile<RM_SP3, true, true>(pix, tw, rp, st) : ring_launch_tile<RM_SP3, true>(pix, tw, rp, st);
    case RM_K3S2: return ring_launch_tile<RM_K3S2, false>(pix, tw, rp, st);
    default: return sg ? ring_launch_tile<RM_K4S2, true, true>(pix, tw, rp, st) : ring_launch_tile<RM_K4S2, true>(pix, tw, rp, st);
    }
}

}  // namespace pws

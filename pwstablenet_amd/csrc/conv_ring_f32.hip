// Exact-fp32 implicit-GEMM convolution for gfx950 on the persistent LDS-ring structure of conv_ring.hip (read its header first):
// 8 matrix waves + 4 loader waves per CU, K groups streamed through an LDS ring by LDS-DMA, counted waits, one barrier per group,
// the stride-2 kind as a sum over input parity planes and the transposed kind as output parity classes of 2x2 stride-1
// convolutions.  This is the fp32 inference path of BASELINE configs[1] (the headline `value`).
// v_mfma_f32_32x32x2_f32 is 16x slower than the bf16 instruction, so a K group of 16 channels is ~18 000 matrix cycles per wave
// against ~5 000 cycles of DMA: the staging hides completely and the kernel is matrix-bound.  What the first-generation kernel
// (conv_mfma_kernel: register-staged chunks, two barriers per chunk, a prologue / epilogue per 256-pixel tile, ~75 % of the fp32
// peak on its best layers, 50-60 % on the stride-2 kind) loses to pipeline bubbles is what this one recovers.
//   * a K group = 16 input channels: input rows [pixel][16 fp32] of 64 bytes (one L2 request; 16-byte slots XOR-permuted as in
//     conv_ring.hip), weight rows [tap][channel][64 cout fp32] of 256 bytes straight from the packed layout [tap][cin][cout];
//   * operands: the MFMA takes ONE fp32 per lane and k-step, and which channel a k-step means is free as long as A and B agree.
//     A lane's ds_read_b128 of its pixel row therefore delivers FOUR k-steps of A: k-step (t, i) of lane half `hi` is channel
//     4 (2 t + hi) + i; B is the matching weight row, read with ds_read_b32 (32 consecutive cout: conflict-free);
//   * numerics: exact fp32 products and sums as conv_mfma_kernel; the K order differs (channel order inside a group; plane-major
//     for the stride-2 kind), so the two agree to fp32 summation order, not bit for bit;
//   * epilogue: bias + activation on the accumulators, one 128-byte (32-channel) store per pixel and lane half; forward only
//     (fp32 training stays on conv_mfma_kernel / conv_wgrad.hip).
#include "conv_common.h"

namespace pws {

typedef float rf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rf_u32x2 __attribute__((ext_vector_type(2)));

enum RingfMode {
    RF_K3S1 = 0,  // conv k3 s1 p1 (and transposed k3 s1 p1 = flipped taps): dense in, dense out
    RF_CT4 = 1,   // transposed conv k4 s2 p1: 4 output parity classes, each a 2x2 conv of the dense input
    RF_K3S2 = 2   // conv k3 s2 p1: sum over the 4 input parity planes of 2x2 convs with 4/2/2/1 taps
};

struct RingfParams {
    const float *src_ptr[4];  // fp32 NHWC sources of the virtual concat
    int src_c[4], src_ld[4];
    int nsrc;
    int N, H, W;     // extent of the source tensors
    int LH, LW;      // logical output extent the tiles walk (class / plane grid)
    int OH, OW;      // extent of the output tensor
    int cout, cin_pad;
    const float *w;       // packed [plane][cin_pad][cout] fp32 (pws_pack_conv_weight)
    size_t w_bytes;
    const float *bias;
    int act;
    float *out;
    int out_ld;
    int tiles_x, tiles_y;
    unsigned ncob, ncls, nunits;
    int gpp;              // K groups per plane = sum(src_c) / 16
};

template <int MODE_, int TH_, int R_, int NT_ = 2>
struct RfCfg {
    static constexpr int MODE = MODE_, TH = TH_, TW = 32, R = R_;
    // NT: 32-channel output blocks per unit.  2: a matrix wave owns 64 output channels.  1: units of 32 output channels, twice as
    // many of them -- for launches whose 64-channel units would leave half the chip idle (the stride-2 layers on 64^2 inputs)
    static constexpr int NT = NT_, CO_UNIT = 32 * NT, WROWB = 128 * NT, WSL = 8 * NT;   // weight row: bytes, 16-byte slots
    static_assert(NT == 1 || NT == 2, "output-channel blocks");
    static constexpr int KS = MODE == RF_K3S1 ? 3 : 2;
    static constexpr int TAPS = KS * KS;
    static constexpr int NPLANES = MODE == RF_K3S2 ? 4 : 1;   // input parity planes
    static constexpr int NCLS = MODE == RF_CT4 ? 4 : 1;       // output parity classes
    static constexpr int MWAVES = 8, LWAVES = 4, THREADS = 64 * (MWAVES + LWAVES);
    // tile = TH rows of 32 pixels of one sample; a matrix wave owns MT = TH / 8 rows (32-pixel operands) x 64 output channels:
    // 16 x 32 tiles where the launch has enough of them for 256 persistent workgroups, 8 x 32 tiles otherwise
    static_assert(TH == 16 || TH == 8, "tile");
    static constexpr int MT = TH / 8;
    static constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    static constexpr int CKG = 16;                                // input channels per K group
    static constexpr int ROWB = CKG * 4;                          // bytes per input row: one 64-byte L2 request
    static constexpr int SPP = ROWB / 16;                         // 16-byte slots per input row
    static constexpr int IN_SLOTS = IH * IW * SPP;
    static constexpr int IN_WI = (IN_SLOTS + 63) / 64;            // wave-instructions (64 slots each)
    static constexpr int W_WI = TAPS * CKG * WROWB / 1024;        // TAPS x 16 channel rows x 128 NT bytes
    static constexpr int NL = (IN_WI + W_WI + LWAVES - 1) / LWAVES; // DMA instructions per loader wave and group
    static constexpr int GROUP_BYTES = NL * LWAVES * 1024;
    static constexpr int W_OFF = IN_WI * 1024;
    static constexpr int LDS_BYTES = R * GROUP_BYTES;
    static constexpr int MIX_IT = IN_WI % LWAVES == 0 ? -1 : IN_WI / LWAVES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(R >= 2 && R <= 4, "ring depth");
    static_assert((R - 2) * NL <= 63, "vmcnt is 6 bits");
};

// (the DMA / wait / uniformity helpers are those of conv_ring.hip; see the comments there)
__device__ __forceinline__ void ringf_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
// the same inside a bracket that saved M0 and restores it; s_nop 3 + the two instructions behind it = the 5 wait states a VALU-written
// SGPR needs before a VMEM instruction reads it (conv_ring.hip)
__device__ __forceinline__ void ringf_dma16_m0(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
template <int N>
__device__ __forceinline__ void ringf_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
constexpr unsigned kRingfOob = 0x7ffffff0u;
__device__ __forceinline__ unsigned unif(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *unif(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)unif((unsigned)(a >> 32)) << 32) | unif((unsigned)a));
}
template <class T>
__device__ __forceinline__ T selq4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

struct RingfUnit {   // decoded (tile, cout block, class)
    int n0, y0, x0, co0, py, px;
};
__device__ __forceinline__ RingfUnit ringf_unit(const RingfParams &p, unsigned u, int co_unit) {
    RingfUnit r;
    const unsigned cls = u % p.ncls, rest = u / p.ncls;
    const unsigned cob = rest % p.ncob, tile = rest / p.ncob;
    const unsigned tx = tile % (unsigned)p.tiles_x, t2 = tile / (unsigned)p.tiles_x;
    const unsigned ty = t2 % (unsigned)p.tiles_y;
    r.n0 = (int)(t2 / (unsigned)p.tiles_y);
    r.y0 = (int)ty, r.x0 = (int)tx, r.co0 = (int)cob * co_unit, r.py = (int)(cls >> 1), r.px = (int)(cls & 1);
    return r;
}

template <class C>
__global__ void __launch_bounds__(C::THREADS, 3) conv_ringf_kernel(const RingfParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // unit assignment: as conv_ring.hip (XCD-contiguous chunk, round-robin inside the XCD)
    const unsigned G = gridDim.x;
    const unsigned nxc = G < (unsigned)kXcds ? G : (unsigned)kXcds;
    const unsigned xcd = blockIdx.x % nxc, slot = blockIdx.x / nxc;
    const unsigned nx = G / nxc + (xcd < G % nxc ? 1u : 0u);
    const unsigned c_begin = (unsigned)((unsigned long long)xcd * p.nunits / nxc);
    const unsigned c_end = (unsigned)((unsigned long long)(xcd + 1) * p.nunits / nxc);
    if (c_begin + slot >= c_end) return;
    const unsigned u_begin = c_begin + slot, u_end = c_end, u_step = nx;
    const unsigned my_units = (u_end - u_begin + u_step - 1) / u_step;
    const int ngroups = C::NPLANES * p.gpp;
    const unsigned total = my_units * (unsigned)ngroups;

    if (wv >= C::MWAVES) {
        // =========================================================================================== loader waves
        const int lw = wv - C::MWAVES;
        // item `it` of this lane = 16-byte slot (it * 4 + lw) * 64 + lane of a group image.  Input: row = pixel, 4 slots of 4
        // channels, permuted by XOR with bits 2..3 of the pixel's x.  Weights: row = (tap, channel), 16 slots of 4 output channels.
        int ia[C::NL], ib[C::NL];   // input: (ly << 10 | lx), c16  /  weights: tap * 16 + k, cout slot  /  filler: -1
#pragma unroll
        for (int it = 0; it < C::NL; ++it) {
            const int wi = it * C::LWAVES + lw;
            const int j = wi * 64 + lane;
            if (wi < C::IN_WI) {
                const int q = j / C::SPP, sp = j % C::SPP;
                const int lx = q % C::IW, ly = q / C::IW;
                ia[it] = j < C::IN_SLOTS ? (ly << 10 | lx) : -1;
                ib[it] = (sp ^ ((lx >> 2) & 3)) * 16;
            } else if (wi < C::IN_WI + C::W_WI) {
                const int jj = j - C::IN_WI * 64;
                ia[it] = jj / C::WSL;   // tap * 16 + k
                ib[it] = jj % C::WSL;   // 4-cout slot
            } else {
                ia[it] = -1, ib[it] = 0;
            }
        }
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.w_bytes, 0x00020000);
        // The groups are walked by nested loops (unit, plane, source, 16-channel group), as in conv_ring.hip since round 5: what depends on
        // (unit, plane, source) -- descriptor, halo offset, per-lane offsets cv[] -- is set up once per source, the innermost loop bumps two
        // scalar offsets.  (The cursor + stage() form of rounds 2-4 compiled to several hundred scalar instructions per group.)  Protocol
        // unchanged: group g is issued behind barrier B_(g - R + 1); R - 1 barriers remain at the end.
        unsigned loc[C::NL], cv[C::NL];   // cached per-lane source offsets
        int loc_ld = -1;
#pragma unroll
        for (int it = 0; it < C::NL; ++it) loc[it] = kRingfOob, cv[it] = kRingfOob;
        unsigned issued = 0, dslot = 0;
        const bool mix_in = C::MIX_IT >= 0 && C::MIX_IT * C::LWAVES + lw < C::IN_WI;
        for (unsigned pu = u_begin; pu < u_end; pu += u_step) {
            const RingfUnit PU = ringf_unit(p, pu, C::CO_UNIT);
            for (int pplane = 0; pplane < C::NPLANES; ++pplane) {
                const int a = pplane >> 1, b = pplane & 1;
                int oy, ox;   // view coordinates of the halo's first pixel
                if constexpr (C::MODE == RF_K3S1) oy = PU.y0 * C::TH - 1, ox = PU.x0 * C::TW - 1;
                else if constexpr (C::MODE == RF_CT4) oy = PU.y0 * C::TH - (1 - PU.py), ox = PU.x0 * C::TW - (1 - PU.px);
                else oy = PU.y0 * C::TH - a, ox = PU.x0 * C::TW - b;
                unsigned d_sw = unif((unsigned)((size_t)PU.co0 * 4));   // + 16 channel rows of cout floats per group of the plane
                const unsigned wplane = (unsigned)(p.cin_pad * p.cout * 4);
                const unsigned wstep = (unsigned)(C::CKG * p.cout * 4);
                for (int ps = 0; ps < p.nsrc; ++ps) {
                    const int ld = selq4(p.src_ld, ps), srcc = selq4(p.src_c, ps);
                    const size_t img = (size_t)p.H * p.W * ld * 4;   // bytes of one sample
                    const char *base_in = unif(reinterpret_cast<const char *>(selq4(p.src_ptr, ps)) + (size_t)PU.n0 * img);
                    const __amdgpu_buffer_rsrc_t d_rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base_in), 0, (int)unif((unsigned)img), 0x00020000);
                    const unsigned ldb = (unsigned)ld * 4u;
                    const __amdgpu_buffer_rsrc_t d_rmix = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char *>(unif(mix_in ? base_in : reinterpret_cast<const char *>(p.w))), 0,
                        (int)unif(mix_in ? (unsigned)img : (unsigned)p.w_bytes), 0x00020000);
                    constexpr int S = C::NPLANES == 4 ? 2 : 1;
                    if (ld != loc_ld) {
                        loc_ld = ld;
#pragma unroll
                        for (int it = 0; it < C::NL; ++it) {
                            if (it * C::LWAVES >= C::IN_WI) continue;
                            const int ly = ia[it] >> 10, lx = ia[it] & 0x3ff;
                            loc[it] = ia[it] >= 0 ? (unsigned)(S * ly * p.W + S * lx) * ldb + (unsigned)ib[it] : kRingfOob;
                        }
                    }
                    const int fy = S * oy + (S == 2 ? a : 0), fx = S * ox + (S == 2 ? b : 0);   // image coordinates of the halo's first pixel
                    const unsigned s_halo = unif((unsigned)((fy * p.W + fx) * (int)ldb));
                    const bool interior = fy >= 0 && fy + S * (C::IH - 1) < p.H && fx >= 0 && fx + S * (C::IW - 1) < p.W;   // scalar
#pragma unroll
                    for (int it = 0; it < C::NL; ++it) {
                        const bool in_ct = (it + 1) * C::LWAVES <= C::IN_WI, w_ct = it * C::LWAVES >= C::IN_WI;
                        unsigned v_in = kRingfOob, v_w = kRingfOob;
                        if (!w_ct) {
                            v_in = loc[it] + s_halo;
                            if (!interior) {
                                const int ry = fy + S * (ia[it] >> 10), rx = fx + S * (ia[it] & 0x3ff);
                                v_in = (ia[it] >= 0 && ry >= 0 && ry < p.H && rx >= 0 && rx < p.W) ? v_in : kRingfOob;
                            }
                        }
                        if (!in_ct) {
                            const int t = ia[it] >> 4, k = ia[it] & 15;
                            int wt = t;   // plane of the packed weights this tap slot reads
                            bool ok = ia[it] >= 0 && PU.co0 + ib[it] * 4 < p.cout;
                            const int ty = t >> 1, tx = t & 1;
                            if constexpr (C::MODE == RF_CT4) wt = (PU.py * 2 + PU.px) * 4 + t;
                            if constexpr (C::MODE == RF_K3S2) wt = (a ? 2 * ty : 1) * 3 + (b ? 2 * tx : 1), ok = ok && ty <= a && tx <= b;
                            v_w = ok ? (unsigned)wt * wplane + (unsigned)(k * p.cout + ib[it] * 4) * 4u : kRingfOob;
                        }
                        cv[it] = in_ct ? v_in : (w_ct ? v_w : (mix_in ? v_in : v_w));
                    }
                    unsigned d_sin = 0;   // byte offset of the group's 16 channels inside a pixel of this source
                    for (int pc0 = 0; pc0 < srcc; pc0 += C::CKG) {
                        if (issued >= (unsigned)(C::R - 1)) {
                            ringf_wait_vmcnt<(C::R - 2) * C::NL>();   // this wave's pieces of group issued - R + 1 have landed
                            __builtin_amdgcn_s_barrier();             // B_(issued - R + 1)
                        }
                        const unsigned d_base = unif(dslot * (unsigned)C::GROUP_BYTES + (unsigned)(lw * 1024));
                        const unsigned d_smix = mix_in ? d_sin : d_sw;
                        unsigned keep;
                        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
                        for (int it = 0; it < C::NL; ++it) {
                            const bool in_ct = (it + 1) * C::LWAVES <= C::IN_WI, w_ct = it * C::LWAVES >= C::IN_WI;
                            const unsigned dst = d_base + (unsigned)(it * C::LWAVES * 1024);
                            if (in_ct) ringf_dma16_m0(dst, cv[it], d_rin, d_sin);
                            else if (w_ct) ringf_dma16_m0(dst, cv[it], rsrc_w, d_sw);
                            else ringf_dma16_m0(dst, cv[it], d_rmix, d_smix);
                        }
                        asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
                        d_sin += (unsigned)(C::CKG * 4), d_sw += wstep;
                        ++issued;
                        dslot = dslot + 1 == (unsigned)C::R ? 0u : dslot + 1;
                    }
                }
            }
        }
        // the last min(total, R - 1) barriers: the groups still in flight land one by one (exact counts: no filler groups)
        const unsigned rem = issued < (unsigned)(C::R - 1) ? issued : (unsigned)(C::R - 1);
        for (unsigned j = 0; j < rem; ++j) {
            const unsigned left = rem - 1 - j;
            if (left >= 2) ringf_wait_vmcnt<(C::R >= 4 ? 2 : 0) * C::NL>();
            else if (left == 1) ringf_wait_vmcnt<(C::R >= 3 ? 1 : 0) * C::NL>();
            else ringf_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =============================================================================================== matrix waves
    // A: lane (l31, hi) reads 16-byte slot 2 t + hi of its pixel row (XOR-permuted; t = 1 is the address with bit 5 flipped)
    int a_off[C::MT][C::KS];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int dx = 0; dx < C::KS; ++dx) {
            const int lx = l31 + dx;
            a_off[mt][dx] = ((wv * C::MT + mt) * C::IW + lx) * C::ROWB + ((hi ^ ((lx >> 2) & 3)) << 4);
        }
    // B: weight row of channel 4 (2 t + hi) + i, output channel nt * 32 + l31
    // (NT == 2: output channels split even / odd over the two accumulators -- a lane's B operands of both are ONE 8-byte read and its
    //  results 8-byte stores; every LDS / vector instruction beside the fp32 matrix instructions costs matrix time: conv_first.hip)
    const int b_off = C::W_OFF + hi * 4 * C::WROWB + l31 * 4 * C::NT;

    f32x16 acc[C::MT][C::NT];
    unsigned cu = u_begin;
    int cg = 0, cplane = 0, cgp = 0, cbuf = 0;
    RingfUnit CU = ringf_unit(p, cu, C::CO_UNIT);

    for (unsigned s = 0; s < total; ++s) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_s
        asm volatile("" ::: "memory");
        if (cg == 0) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        }
        unsigned tapmask = (1u << C::TAPS) - 1u;
        if constexpr (C::MODE == RF_K3S2) tapmask = (cplane >> 1) ? ((cplane & 1) ? 0xfu : 0x5u) : ((cplane & 1) ? 0x3u : 0x1u);
        const unsigned gb = (unsigned)(cbuf * C::GROUP_BYTES);
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            if constexpr (C::MODE == RF_K3S2) {
                if (!((tapmask >> tap) & 1u)) continue;   // wave-uniform: 1, 2 or 4 taps per plane
            }
            const int ty = tap / C::KS, tx = tap % C::KS;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 a4[C::MT];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
                    a4[mt] = *reinterpret_cast<const f32x4 *>(lds + gb + (unsigned)((a_off[mt][tx] ^ (t * 32)) + ty * C::IW * C::ROWB));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float bv[C::NT];
                    if constexpr (C::NT == 2) {
                        const rf_f32x2 b2 = *reinterpret_cast<const rf_f32x2 *>(lds + gb + (unsigned)(b_off + (tap * 16 + 8 * t + i) * C::WROWB));
                        bv[0] = b2[0], bv[1] = b2[1];
                    } else {
                        bv[0] = *reinterpret_cast<const float *>(lds + gb + (unsigned)(b_off + (tap * 16 + 8 * t + i) * C::WROWB));
                    }
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < C::NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mt][i], bv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }

        if (cg == ngroups - 1) {
            // ---- epilogue of unit cu: bias + activation on the accumulators; lane (l31, hi) holds channels NT l31 (+ 1) of the pixels
            // (r & 3) + 8 (r >> 2) + 4 hi of its 32-pixel rows.  Stores are buffer stores with the pixel as SCALAR offset: the
            // epilogue's vector work is the bias and the activation alone (NT == 2: 8-byte stores, 256 bytes per pixel and half-wave)
            constexpr int SO = C::NCLS == 4 ? 2 : 1;
            const int co = CU.co0 + C::NT * l31;
            const bool co_ok = co < p.cout;
            float bs[C::NT];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) bs[nt] = (p.bias && co_ok) ? p.bias[co + nt] : 0.f;
            const size_t oimg = (size_t)p.OH * p.OW * p.out_ld;
            const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)CU.n0 * oimg, 0, (int)(oimg * 4), 0x00020000);
            const unsigned pxb = (unsigned)(SO * p.out_ld * 4);   // bytes between consecutive tile pixels in the output
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const int y = CU.y0 * C::TH + wv * C::MT + mt;
                const int oy = SO * y + (C::NCLS == 4 ? CU.py : 0);
                const int ox0 = SO * (CU.x0 * C::TW + 4 * hi) + (C::NCLS == 4 ? CU.px : 0);
                const unsigned vo = co_ok ? (unsigned)(((oy * p.OW + ox0) * p.out_ld + co) * 4) : kRingfOob;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = (int)(((r & 3) + 8 * (r >> 2)) * pxb);
                    if constexpr (C::NT == 2) {
                        const rf_f32x2 v = {act_apply(acc[mt][0][r] + bs[0], p.act), act_apply(acc[mt][1][r] + bs[1], p.act)};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rf_u32x2, v), rsrc_o, (int)vo, soff, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, act_apply(acc[mt][0][r] + bs[0], p.act)), rsrc_o, (int)vo, soff, 0);
                    }
                }
            }
            cg = 0, cplane = 0, cgp = 0, cu += u_step;
            if (cu < u_end) CU = ringf_unit(p, cu, C::CO_UNIT);
        } else {
            ++cg;
            if (++cgp == p.gpp) cgp = 0, ++cplane;
        }
        cbuf = cbuf + 1 == C::R ? 0 : cbuf + 1;
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <class C>
static int ringf_launch(RingfParams &rp, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ringf_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_ringf_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
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
    rp.tiles_x = rp.LW / C::TW, rp.tiles_y = rp.LH / C::TH;   // whole tiles only (conv_ringf_try)
    rp.ncob = (unsigned)((rp.cout + C::CO_UNIT - 1) / C::CO_UNIT), rp.ncls = (unsigned)C::NCLS;
    rp.nunits = (unsigned)(rp.tiles_x * rp.tiles_y) * (unsigned)rp.N * rp.ncob * rp.ncls;
    int cin = 0;
    for (int s = 0; s < rp.nsrc; ++s) cin += rp.src_c[s];
    rp.gpp = cin / C::CKG;
    const unsigned grid = rp.nunits < (unsigned)ncu ? rp.nunits : (unsigned)ncu;   // one persistent workgroup per CU
    hipLaunchKernelGGL((conv_ringf_kernel<C>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, st, rp);
    return check_launch("conv_ringf_kernel");
}

template <int MODE>
static int ringf_launch_tile(int th, int nt, RingfParams &rp, hipStream_t st) {
    constexpr int KS2 = MODE != RF_K3S1;
    if constexpr (MODE == RF_K3S2) {
        if (th == 8 && nt == 1) return ringf_launch<RfCfg<MODE, 8, 4, 1>>(rp, st);
    }
    if (th == 16) return ringf_launch<RfCfg<MODE, 16, KS2 ? 3 : 2>>(rp, st);
    return ringf_launch<RfCfg<MODE, 8, KS2 ? 4 : 2>>(rp, st);
}

// Runs the fp32 forward launch described by kp (prepared by conv2d_fwd_impl, conv_mfma.hip) on the fp32 ring kernel when it is
// covered: NHWC fp32 sources (multiples of 16 channels, 16-byte aligned), maps at least 32 wide in whole tiles, enough units to
// occupy the chip.  `prefer_other`: the caller has a faster specialised kernel for this launch when the map is large (Winograd for
// the 3x3 kind).  Returns 1 when not covered.
int conv_ringf_try(int kind, const ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    if (kp.io_bf16 || kp.ndst != 0 || g_experiment == 22) return 1;
    for (int s = 0; s < kp.nsrc; ++s)
        if (kp.src_ld[s] == 0 || kp.src_c[s] % 16 != 0 || kp.src_ld[s] % 4 != 0 || (reinterpret_cast<size_t>(kp.src_ptr[s]) & 15)) return 1;
    if (kp.LW % 32 != 0 || kp.LH % 8 != 0 || kp.cout % 4 != 0) return 1;
    // (8-byte stores of channel pairs through one buffer descriptor per output sample)
    if (kp.out_ld % 2 != 0 || (reinterpret_cast<size_t>(kp.out) & 7) || (size_t)kp.OH * kp.OW * kp.out_ld * 4 >= (1u << 31)) return 1;
    int mode, planes;
    if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) mode = RF_K3S1, planes = 9;
    else if (kind == PWS_CONV_K3S2) mode = RF_K3S2, planes = 9;
    else if (kind == PWS_CONVT_K4S2) mode = RF_CT4, planes = 16;
    else return 1;
    for (int s = 0; s < kp.nsrc; ++s)
        if ((size_t)kp.H * kp.W * 4 * kp.src_ld[s] >= (1u << 31)) return 1;
    RingfParams rp{};
    for (int s = 0; s < 4; ++s) rp.src_ptr[s] = kp.src_ptr[s], rp.src_c[s] = kp.src_c[s], rp.src_ld[s] = kp.src_ld[s];
    rp.nsrc = kp.nsrc, rp.N = kp.N, rp.H = kp.H, rp.W = kp.W, rp.LH = kp.LH, rp.LW = kp.LW, rp.OH = kp.OH, rp.OW = kp.OW;
    rp.cout = kp.cout, rp.cin_pad = kp.cin_pad, rp.w = kp.w, rp.w_bytes = (size_t)planes * kp.cin_pad * kp.cout * 4;
    if (rp.w_bytes >= (1u << 31)) return 1;
    rp.bias = kp.bias, rp.act = kp.act, rp.out = kp.out, rp.out_ld = kp.out_ld;
    const long per256 = (long)kp.LW * kp.LH * kp.N / 256 * ((kp.cout + 63) / 64) * (mode == RF_CT4 ? 4 : 1);   // units of 8 x 32 tiles
    // 16 x 32 tiles when they still give every CU two units, else 8 x 32; too few even then: the split-K kernels do better
    // Measured (tools/layer_profile.py, batch 8, one queue): the fp32 MFMA path is matrix-bound either way -- the transposed kind
    // runs 118-126 TFLOP/s on both kernels (the chip clocks down under sustained fp32 MFMA load: ~2.0 GHz, i.e. ~131 TFLOP/s of
    // peak at that clock) -- so this kernel is taken where its missing per-tile prologue / epilogue shows: the stride-2 kind and the
    // direct 3x3 kind on maps large enough for 16 x 32 tiles (256^2 stride-2: 123 -> 114 us, 3x3 128->128 @128^2: 100 -> 88 us);
    // the 8 x 32-tile variant and the transposed kind stay available to the tests (PWS_OPT_EXPERIMENT 23 / 24).
    const long per256_32 = (long)kp.LW * kp.LH * kp.N / 256 * ((kp.cout + 31) / 32);   // ... with 32 output channels each
    int th, nt = 2;
    const bool forced = g_experiment == 23 || g_experiment == 24;
    // (stride-2 kind: from one 16 x 32 unit per CU upwards -- 64 -> 64 @256^2 x 8 = 256 units: 113 vs 136 us of conv_mfma_kernel; with
    //  fewer units than CUs the persistent kernel loses: 128 -> 128 @128^2 x 8 = 128 units 188 vs 109 us)
    if (g_experiment == 27 && mode == RF_K3S2) th = 8, nt = 1;   // (the 32-channel units forced for the tests' small launches)
    else if (kp.LH % 16 == 0 && (per256 / 2 >= (mode == RF_K3S2 ? 256 : 512) || g_experiment == 24)) th = 16;
    // (stride-2 kind with one 8 x 32 unit per CU, measured after the loader's offsets moved into registers: 128 -> 128 @128^2 x 8
    //  91 vs 98 us, 64 -> 128 @128^2 52 vs 60 us of conv_mfma_kernel; half a chip of units loses: 256 -> 256 @64^2 158 vs 98 us)
    else if (forced || (mode == RF_K3S2 && per256 >= 256)) th = 8;
    // (units of 32 output channels where 64-channel units fill only half the chip: 256 -> 256 @64^2 x 8 = 256 such units)
    else if (mode == RF_K3S2 && per256_32 >= 256 && g_experiment != 26) th = 8, nt = 1;
    else return 1;
    if (mode == RF_CT4 && !forced) return 1;
    ProfScope prof(KID_CONV_RINGF, pi.flops, pi.bytes, st);
    switch (mode) {
    case RF_K3S1: return ringf_launch_tile<RF_K3S1>(th, nt, rp, st);
    case RF_CT4: return ringf_launch_tile<RF_CT4>(th, nt, rp, st);
    default: return ringf_launch_tile<RF_K3S2>(th, nt, rp, st);
    }
}

}  // namespace pws

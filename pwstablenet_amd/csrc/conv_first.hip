// The generator's first layer (Conv2d(31, 64, k5, p2) on the reference's NCHW window, lib/networks_cascading.py:21-23 `inconv`) as a
// persistent exact-fp32 MFMA kernel for gfx950 -- the fp32 inference path of BASELINE configs[1].
// conv_mfma_kernel<k5s1, NCHW> ran this launch at 68 % matrix-pipe busy (profiles/r03_pmc.json: register-staged chunks of 8 channels,
// two barriers per chunk, a prologue / epilogue per 256-pixel tile, 30 % of its LDS cycles bank conflicts on the [pixel][9] rows).
// What that costs was measured on this kernel's own ablations (tools/first_ablate.sh): every vector / LDS instruction issued on a
// SIMD beside v_mfma_f32_32x32x2_f32 takes ~5 cycles of matrix time (tools/probes/mfma_f32_probe.hip), whichever wave issues it --
// so the design rule here is INSTRUCTION COUNT, not bytes:
//   * the input stays PLANAR in LDS exactly as it lies in memory: a unit = 8 rows x 32 pixels of one sample, its halo 12 rows x 40
//     columns (16-byte aligned: x0 - 4 .. x0 + 35) per channel plane, copied by 16-byte LDS-DMA; image border and padding channel
//     (31 -> 32) are zeroed by out-of-range offsets.  The A operand of the matrix instruction is ONE fp32 per lane: lane (pixel
//     l31, half hi) reads plane 2 kk + hi at its pixel -- 32 consecutive floats per half-wave, conflict-free, and a filter tap is an
//     address offset: no im2col, no transposition.  Planes are 1 920 bytes apart, so ONE ds_read2st64_b32 serves two k-steps;
//   * output channels are split even / odd over the two accumulators of a wave: a lane's B operands of both are one 8-byte read
//     (two k-steps: one ds_read2_b64) and its results are 8-byte stores (buffer_store_dwordx2 with the pixel as scalar offset);
//   * weights stream through a 2-deep LDS ring in groups of (filter row ky, 16 input channels) = 5 taps x 16 rows x 64 cout = 20 KB
//     straight from the packed layout [tap][32][cout]; a group is 80 matrix instructions per wave (5 120 cycles) between barriers;
//   * the NEXT unit's input planes are copied into the second input buffer alongside the weight groups, so a unit starts without
//     waiting for memory; the 4 loader waves keep every per-lane offset of their 15 + 5 DMA instructions in registers (the group
//     loop is unrolled): an interior unit costs them one vector add per DMA instruction;
//   * numerics: exact fp32 products and sums as conv_mfma_kernel; the K order differs (ky, channel half, kx, channel pair), so the
//     two agree to fp32 summation order, not bit for bit.  Forward only (no gradient scatter epilogue).
#include "conv_common.h"

namespace pws {

struct FirstParams {
    const float *src;      // NCHW
    size_t sstride;        // floats between samples
    int C, N, H, W;
    const float *w;        // packed [25][32][cout] fp32
    unsigned w_bytes;
    int cout;
    const float *bias;
    int act;
    float *out;
    int out_ld;
    int tiles_x, tiles_y;
    unsigned ncob, nunits;
    unsigned long long *clk;   // PWS_OPT_EXPERIMENT 1200: per workgroup (shader cycles, 100 MHz ticks) of the whole kernel
    int ablate;   // timing-only ablations (PWS_OPT_EXPERIMENT 1100 + mask): 1 no operand reads, 2 no DMA, 4 no stores, 8 no barriers
};

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int FT_TH = 8, FT_TW = 32, FT_KS = 5, FT_CP = 32;
constexpr int FT_IH = FT_TH + FT_KS - 1, FT_IW = 40;                     // 12 rows x 40 columns (x0 - 4 .. x0 + 35)
constexpr int FT_PLANE = FT_IH * FT_IW;                                  // floats per channel plane: 1 920 bytes
constexpr int FT_IN_BYTES = FT_CP * FT_PLANE * 4;
constexpr int FT_ROW_SLOTS = FT_IW / 4, FT_PLANE_SLOTS = FT_IH * FT_ROW_SLOTS;
constexpr int FT_IN_WI = FT_CP * FT_PLANE_SLOTS / 64;                    // 16-byte wave-instructions per unit: 60
static_assert(FT_CP * FT_PLANE_SLOTS % 64 == 0, "input image in whole wave-instructions");
constexpr int FT_NG = 10;                                                // groups per unit: (ky, channel half)
constexpr int FT_MW = 8, FT_LW = 4, FT_THREADS = 64 * (FT_MW + FT_LW);
constexpr int FT_IN_K = FT_IN_WI / FT_LW;                                // input DMA instructions per loader wave and unit: 15
static_assert(FT_IN_WI % FT_LW == 0 && FT_IN_K <= 2 * FT_NG, "two input pieces per group at most");
constexpr int FT_W_BYTES = FT_KS * 16 * 256;                             // one weight group
constexpr int FT_W_IT = FT_W_BYTES / 1024 / FT_LW;                       // weight DMA instructions per loader wave and group: 5
constexpr int FT_W_BASE = 2 * FT_IN_BYTES;
constexpr int FT_LDS = FT_W_BASE + 2 * FT_W_BYTES;
static_assert(FT_LDS <= 160 * 1024, "LDS");
constexpr unsigned kFirstOob = 0x7ffffff0u;

__device__ __forceinline__ void first_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
__device__ __forceinline__ unsigned funif(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *funif(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)funif((unsigned)(a >> 32)) << 32) | funif((unsigned)a));
}

struct FirstUnit {
    int n, y0, x0, co0;
};
__device__ __forceinline__ FirstUnit first_unit(const FirstParams &p, unsigned u) {
    FirstUnit r;
    const unsigned cob = u % p.ncob, tile = u / p.ncob;
    const unsigned tx = tile % (unsigned)p.tiles_x, t2 = tile / (unsigned)p.tiles_x;
    r.y0 = (int)(t2 % (unsigned)p.tiles_y) * FT_TH, r.n = (int)(t2 / (unsigned)p.tiles_y);
    r.x0 = (int)tx * FT_TW, r.co0 = (int)cob * 64;
    return r;
}
}  // namespace

__global__ void __launch_bounds__(FT_THREADS, 3) conv_first_kernel(const FirstParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // unit assignment: as conv_ring.hip (XCD-contiguous chunk of the unit list, round-robin inside the XCD)
    const unsigned G = gridDim.x;
    const unsigned nxc = G < (unsigned)kXcds ? G : (unsigned)kXcds;
    const unsigned xcd = blockIdx.x % nxc, slot = blockIdx.x / nxc;
    const unsigned nx = G / nxc + (xcd < G % nxc ? 1u : 0u);
    const unsigned c_begin = (unsigned)((unsigned long long)xcd * p.nunits / nxc);
    const unsigned c_end = (unsigned)((unsigned long long)(xcd + 1) * p.nunits / nxc);
    if (c_begin + slot >= c_end) return;
    const unsigned long long clk_c0 = p.clk ? __builtin_amdgcn_s_memtime() : 0ull, clk_r0 = p.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned u_begin = c_begin + slot, u_end = c_end, u_step = nx;
    const bool no_dma = p.ablate & 2, no_bar = p.ablate & 8;

    if (wv >= FT_MW) {
        // =========================================================================================== loader waves
        const int lw = wv - FT_MW;
        // weight item `it` of this lane = 16-byte slot (it * 4 + lw) * 64 + lane of a group: row (kx, k) of 16 slots of 4 cout
        unsigned w_voff[FT_W_IT];
        int w_q[FT_W_IT];
#pragma unroll
        for (int it = 0; it < FT_W_IT; ++it) {
            const int j = (it * FT_LW + lw) * 64 + lane;
            const int row = j >> 4, q = j & 15;
            const int kx = row >> 4, k = row & 15;
            w_voff[it] = (unsigned)((kx * FT_CP + k) * p.cout + q * 4) * 4u;
            w_q[it] = q * 4;
        }
        // input item k of this lane = 16-byte slot (k * 4 + lw) * 64 + lane of the unit's planes: (plane c, row ly, 4 columns q)
        unsigned i_voff[FT_IN_K];   // byte offset of the slot from the halo's first element; out of range for the padding planes
        int i_geo[FT_IN_K];         // ly | q << 8
#pragma unroll
        for (int k = 0; k < FT_IN_K; ++k) {
            const int j = (k * FT_LW + lw) * 64 + lane;
            const int c = j / FT_PLANE_SLOTS, rem = j - c * FT_PLANE_SLOTS;
            const int ly = rem / FT_ROW_SLOTS, q = rem - ly * FT_ROW_SLOTS;
            i_voff[k] = c < p.C ? (unsigned)((c * p.H + ly) * p.W + q * 4) * 4u : kFirstOob;
            i_geo[k] = ly | q << 8;
        }
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.w_bytes, 0x00020000);
        const unsigned img_bytes = (unsigned)p.C * (unsigned)p.H * (unsigned)p.W * 4u;

        auto stage_w = [&](int wbuf, const FirstUnit &U, int g) {
            if (no_dma) return;
            const int ky = g >> 1, ch = g & 1;
            const unsigned soff = funif((unsigned)(((ky * FT_KS * FT_CP + ch * 16) * p.cout + U.co0) * 4));
            const unsigned dst = funif((unsigned)(FT_W_BASE + wbuf * FT_W_BYTES + lw * 1024));
            if (U.co0 + 64 <= p.cout) {   // (scalar branch) a whole block of output channels: nothing to mask
#pragma unroll
                for (int it = 0; it < FT_W_IT; ++it) first_dma16(dst + (unsigned)(it * FT_LW * 1024), w_voff[it], rsrc_w, soff);
            } else {
#pragma unroll
                for (int it = 0; it < FT_W_IT; ++it)
                    first_dma16(dst + (unsigned)(it * FT_LW * 1024), U.co0 + w_q[it] < p.cout ? w_voff[it] : kFirstOob, rsrc_w, soff);
            }
        };
        // input slot k of unit U into input buffer ibuf
        auto stage_in = [&](int ibuf, const FirstUnit &U, int k, unsigned voff_k, int geo_k) {
            if (no_dma) return;
            const char *base = funif(reinterpret_cast<const char *>(p.src + (size_t)U.n * p.sstride));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)img_bytes, 0x00020000);
            const int oy = U.y0 - 2, ox = U.x0 - 4;   // image coordinates of the halo's first element
            const unsigned s_unit = funif((unsigned)((oy * p.W + ox) * 4));
            const unsigned dst = funif((unsigned)(ibuf * FT_IN_BYTES + (k * FT_LW + lw) * 1024));
            const bool interior = oy >= 0 && oy + FT_IH <= p.H && ox >= 0 && ox + FT_IW <= p.W;   // scalar
            unsigned v = voff_k + s_unit;   // (a padding plane stays out of range: its offset is ~2^31 and |s_unit| < 2^31)
            if (!interior) {
                const int iy = oy + (geo_k & 0xff), ix = ox + (geo_k >> 8) * 4;
                v = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? v : kFirstOob;
            }
            first_dma16(dst, v, rsrc, 0u);
        };

        unsigned pu = u_begin;
        FirstUnit PU = first_unit(p, pu);
        FirstUnit NU = PU;
        bool have_next = pu + u_step < u_end;
        if (have_next) NU = first_unit(p, pu + u_step);
#pragma unroll
        for (int k = 0; k < FT_IN_K; ++k) stage_in(0, PU, k, i_voff[k], i_geo[k]);
        stage_w(0, PU, 0);
        int ibuf = 0;
        while (pu < u_end) {
#pragma unroll
            for (int g = 0; g < FT_NG; ++g) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // group g (and every input piece issued so far) has landed
                if (!no_bar) __builtin_amdgcn_s_barrier();         // B_s
                // two pieces of the next unit's planes into the other input buffer; group g + 1 into the weight buffer group g - 1 left
                if (have_next) {
                    constexpr int kz = 0;
                    const int k0 = 2 * g < FT_IN_K ? 2 * g : kz, k1 = 2 * g + 1 < FT_IN_K ? 2 * g + 1 : kz;
                    if (2 * g < FT_IN_K) stage_in(ibuf ^ 1, NU, k0, i_voff[k0], i_geo[k0]);
                    if (2 * g + 1 < FT_IN_K) stage_in(ibuf ^ 1, NU, k1, i_voff[k1], i_geo[k1]);
                }
                if (g + 1 < FT_NG) {
                    stage_w((g + 1) & 1, PU, g + 1);
                } else {
                    pu += u_step, ibuf ^= 1;
                    if (pu < u_end) {
                        PU = NU;
                        stage_w(0, PU, 0);
                        have_next = pu + u_step < u_end;
                        if (have_next) NU = first_unit(p, pu + u_step);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =============================================================================================== matrix waves
    // wave wv owns tile row wv: 32 pixels x 64 output channels, even channels in acc[0], odd ones in acc[1].
    // A: plane 2 kk + hi at (row wv + ky, column l31 + kx + 2); B: weight row (kx, 2 kk + hi), channels 2 l31, 2 l31 + 1.
    const unsigned a_lane = (unsigned)((hi * FT_PLANE + wv * FT_IW + l31 + 2) * 4);
    const unsigned b_lane = (unsigned)(FT_W_BASE + (hi * 64 + 2 * l31) * 4);
    f32x16 acc[2];
    unsigned cu = u_begin;
    FirstUnit CU = first_unit(p, cu);
    int g = 0, ibuf = 0;
    const unsigned my_units = (u_end - u_begin + u_step - 1) / u_step;
    const unsigned total = my_units * (unsigned)FT_NG;
    for (unsigned s = 0; s < total; ++s) {
        asm volatile("" ::: "memory");
        if (!no_bar) __builtin_amdgcn_s_barrier();   // B_s
        asm volatile("" ::: "memory");
        if (g == 0) {
            const int co = CU.co0 + 2 * l31;
            f32x2 bs = {0.f, 0.f};
            if (p.bias && co < p.cout) bs = *reinterpret_cast<const f32x2 *>(p.bias + co);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = bs[0], acc[1][r] = bs[1];
        }
        const int ky = g >> 1, ch = g & 1;
        if (p.ablate & 1) {
#pragma unroll 8
            for (int st = 0; st < FT_KS * 8; ++st) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)lane, (float)hi, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)lane, (float)l31, acc[1], 0, 0, 0);
            }
        } else {
            const unsigned ab = a_lane + (unsigned)(ibuf * FT_IN_BYTES + (ch * 16 * FT_PLANE + ky * FT_IW) * 4);
            const unsigned bb = b_lane + (unsigned)((int)(s & 1u) * FT_W_BYTES);
            // a block = the two k-steps (kx, 2 kp) and (kx, 2 kp + 1): one LDS instruction each for A and for B.  Operands are read
            // FT_PF blocks ahead of the matrix instructions that consume them (the asm pins the order: the accumulators tie the matrix
            // instructions to it, the memory clobber the reads)
            constexpr int NB = FT_KS * 4, FT_PF = 2;
            float a0[NB], a1[NB];
            f32x2 b0[NB], b1[NB];
            auto rd = [&](int bl) {
                const int kx = bl / 4, kp = bl % 4;
                a0[bl] = *reinterpret_cast<const float *>(lds + ab + (unsigned)((4 * kp * FT_PLANE + kx) * 4));
                a1[bl] = *reinterpret_cast<const float *>(lds + ab + (unsigned)(((4 * kp + 2) * FT_PLANE + kx) * 4));
                b0[bl] = *reinterpret_cast<const f32x2 *>(lds + bb + (unsigned)((kx * 16 + 4 * kp) * 256));
                b1[bl] = *reinterpret_cast<const f32x2 *>(lds + bb + (unsigned)((kx * 16 + 4 * kp + 2) * 256));
            };
#pragma unroll
            for (int bl = 0; bl < FT_PF; ++bl) rd(bl);
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) {
                if (bl + FT_PF < NB) rd(bl + FT_PF);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : "memory");
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[bl], b0[bl][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[bl], b0[bl][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[bl], b1[bl][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[bl], b1[bl][1], acc[1], 0, 0, 0);
            }
        }
        if (g == FT_NG - 1) {
            // ---- epilogue of unit cu: lane (l31, hi) holds channels 2 l31, 2 l31 + 1 of the pixels (r & 3) + 8 (r >> 2) + 4 hi of its
            // row (the bias is already in the sums): one 8-byte store per pixel, 256 bytes per pixel and half-wave; the pixel is the
            // scalar offset of the store, so the epilogue's vector work is the activation alone
            const int co = CU.co0 + 2 * l31;
            const size_t img = (size_t)p.H * p.W * p.out_ld;
            const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)CU.n * img, 0, (int)(img * 4), 0x00020000);
            const unsigned vo = (co < p.cout && !(p.ablate & 4)) ? (unsigned)((((CU.y0 + wv) * p.W + CU.x0 + 4 * hi) * p.out_ld + co) * 4) : kFirstOob;
            const unsigned ldb = (unsigned)p.out_ld * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x2 v = {act_apply(acc[0][r], p.act), act_apply(acc[1][r], p.act)};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc_o, (int)vo, (int)(((r & 3) + 8 * (r >> 2)) * ldb), 0);
            }
            g = 0, ibuf ^= 1, cu += u_step;
            if (cu < u_end) CU = first_unit(p, cu);
        } else {
            ++g;
        }
    }
    if (p.clk && tid == 0) {
        p.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_c0;
        p.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}

// Runs the first-layer launch described by kp (prepared by pws_conv2d_fwd, conv_mfma.hip) when it is covered: fp32, NCHW source of 17
// to 32 channels with 16-byte aligned rows, whole 8 x 32 tiles, an even cout / out_ld, enough units for half the chip
// (PWS_OPT_EXPERIMENT 25: never; 29: also for the tests' small launches).  Returns 1 when not covered.
int conv_first_try(const ConvKParams &kp, hipStream_t st, const ProfInfo &pi) {
    if (kp.io_bf16 || kp.ndst != 0 || kp.nsrc != 1 || g_experiment == 25) return 1;
    if (kp.src_c[0] > FT_CP || kp.cin_pad != FT_CP || kp.W % FT_TW != 0 || kp.H % FT_TH != 0 || kp.cout % 4 != 0 || kp.out_ld % 2 != 0) return 1;
    const size_t sstride = kp.src_ld[0] ? (size_t)kp.src_ld[0] : (size_t)kp.src_c[0] * kp.H * kp.W;
    if ((reinterpret_cast<size_t>(kp.src_ptr[0]) & 15) || sstride % 4 != 0 || (reinterpret_cast<size_t>(kp.out) & 7) ||
        (kp.bias && (reinterpret_cast<size_t>(kp.bias) & 7)))
        return 1;
    if ((size_t)kp.src_c[0] * kp.H * kp.W * 4 >= (1u << 31) || (size_t)25 * FT_CP * kp.cout * 4 >= (1u << 31) ||
        (size_t)kp.H * kp.W * kp.out_ld * 4 >= (1u << 31))
        return 1;
    FirstParams p{};
    p.src = kp.src_ptr[0], p.C = kp.src_c[0], p.N = kp.N, p.H = kp.H, p.W = kp.W, p.sstride = sstride;
    p.w = kp.w, p.w_bytes = (unsigned)(25 * FT_CP * kp.cout * 4), p.cout = kp.cout;
    p.bias = kp.bias, p.act = kp.act, p.out = kp.out, p.out_ld = kp.out_ld;
    p.tiles_x = kp.W / FT_TW, p.tiles_y = kp.H / FT_TH;
    p.ncob = (unsigned)((kp.cout + 63) / 64);
    p.nunits = (unsigned)(p.tiles_x * p.tiles_y) * (unsigned)kp.N * p.ncob;
    p.ablate = g_experiment >= 1100 && g_experiment < 1116 ? g_experiment - 1100 : 0;
    static PerDeviceInt ncu_dev;
    int &ncu = ncu_dev.cur();
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (p.nunits < (unsigned)(ncu / 2) && g_experiment != 29) return 1;
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_first_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FT_LDS);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_first_kernel, %d B LDS): %s", FT_LDS, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    const unsigned grid = p.nunits < (unsigned)ncu ? p.nunits : (unsigned)ncu;   // one persistent workgroup per CU
    if (g_experiment == 1200) {   // measurement aid (tools/first_ablate.sh): the shader clock the chip holds inside this kernel
        static unsigned long long *clk = nullptr;
        if (!clk && hipHostMalloc(&clk, 2 * 1024 * sizeof(unsigned long long)) != hipSuccess) return PWS_EHIP;
        p.clk = clk;
        hipLaunchKernelGGL(conv_first_kernel, dim3(grid), dim3(FT_THREADS), FT_LDS, st, p);
        if (hipStreamSynchronize(st) != hipSuccess) return PWS_EHIP;
        double cyc = 0, ticks = 0;
        for (unsigned i = 0; i < grid; ++i) cyc += (double)clk[2 * i], ticks += (double)clk[2 * i + 1];
        fprintf(stderr, "conv_first_kernel: %.0f shader cycles and %.1f us per workgroup (mean of %u): %.0f MHz\n", cyc / grid, ticks / grid / 100.0, grid,
                cyc / ticks * 100.0);
        return check_launch("conv_first_kernel");
    }
    ProfScope prof(KID_CONV_FIRST, pi.flops, pi.bytes, st);
    hipLaunchKernelGGL(conv_first_kernel, dim3(grid), dim3(FT_THREADS), FT_LDS, st, p);
    return check_launch("conv_first_kernel");
}

}  // namespace pws

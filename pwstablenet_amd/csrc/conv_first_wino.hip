// The generator's first layer (Conv2d(31, 64, k5, p2) on the reference's NCHW window, lib/networks_cascading.py:21-23 `inconv`) as
// Winograd F(2x2,5x5) in exact fp32 on the matrix cores of gfx950 -- round 5, the fp32 inference path of BASELINE configs[1].
// conv_first_kernel (conv_first.hip) runs the direct convolution at 82 % matrix-pipe busy: nothing left but the multiply count.
// F(2x2,5x5) takes 36 multiplies per 2 x 2 outputs where the direct form takes 100 (2.78x), on the six points {0, 1, -1, 2, -1/2, inf}
// (fp32 error of this layer 1.8e-6 against 1.4e-6 of the direct sum at |y| ~ 1; the textbook points {0, +-1, +-2, inf}: 4.5e-6).
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A,  d = the 6 x 6 input patch of a 2 x 2 output tile.
// With only 32 input channels the transforms are the problem, not the products: every vector instruction beside
// v_mfma_f32_16x16x4_f32 costs ~5 cycles of its SIMD's matrix time (tools/probes/mfma_f32_probe.hip), so
//   * the input transform is computed ONCE per (tile, channel) by 4 helper waves and shared through LDS -- in the matrix waves'
//     registers (conv_wring.hip's way) every 16-channel block of outputs would repeat it;
//   * K runs OUTSIDE the 36 components: a k-step = 4 input channels x all 36 components, which is exactly what one 6 x 6 transform
//     produces; a matrix wave keeps the 36 component accumulators of its 16 channels x 16 tiles (144 registers) for the whole unit and
//     transforms them back in registers at the end: no second trip through LDS;
//   * unit = 4 x 8 tiles (8 x 16 output pixels) x 64 channels: 8 matrix waves (4 channel blocks x 2 tile blocks); per k-step the
//     helper waves copy the next raw planes (LDS-DMA, 4 channels x 12 x 24 floats, zero padding by out-of-range offsets) and the next
//     36 KB of transformed weights U (packed [k-step][4 components][channel block][lane][4]: a lane's A operands of four matrix instructions are one 16-byte LDS read), and transform
//     the planes that landed one step earlier: 6 x 6 -> rows i of B^T d B split over two lanes' waves (rows 0-2 / 3-5);
//   * one s_barrier per k-step; rings: raw planes 3 deep, V (transformed input) 2 deep, U 3 deep = 159 KB of LDS.
// Numerics: exact fp32 products, fp32 sums in Winograd order (not the direct kernel's bits; same tolerance in the tests).
#include "conv_common.h"

namespace pws {

struct Wino5Params {
    const float *src;      // NCHW
    size_t sstride;        // floats between samples
    int C, N, H, W;
    const float *u;        // [8 k-steps][9 groups of 4 components][4 channel blocks][kq * 16 + channel % 16][4] fp32 (wring_pack_element, mode 2)
    const float *bias;
    int act;
    float *out;
    int out_ld;
    int tiles_x, tiles_y;
    unsigned nunits;
    int ablate;            // timing-only (PWS_OPT_EXPERIMENT 1300 + mask): 1 no matrix phase, 2 no transform, 4 no DMA, 8 no epilogue; 16 (not timing-only): every wave in the same order
};

namespace {
constexpr int W5_TR = 4, W5_TC = 8;                      // tiles per unit
constexpr int W5_IH = 2 * W5_TR + 4, W5_IW = 24;         // halo rows y0 - 2 .. y0 + 9, columns x0 - 4 .. x0 + 19 (16-byte slots)
constexpr int W5_PLANE = W5_IH * W5_IW;                  // 288 floats per channel plane
constexpr int W5_CHUNK_SLOTS = 4 * W5_PLANE / 4;         // 16-byte slots of a k-step's 4 planes: 288
constexpr int W5_RAW_WI = (W5_CHUNK_SLOTS + 63) / 64;    // 5 wave-instructions
constexpr int W5_RAW_SLOT = W5_RAW_WI * 1024;
constexpr int W5_RR = 3, W5_RV = 2, W5_RU = 3;           // ring depths
constexpr int W5_V_BYTES = 36 * 512, W5_U_BYTES = 36 * 1024;
constexpr int W5_V_OFF = W5_RR * W5_RAW_SLOT, W5_U_OFF = W5_V_OFF + W5_RV * W5_V_BYTES;
constexpr int W5_LDS = W5_U_OFF + W5_RU * W5_U_BYTES;
static_assert(W5_LDS <= 160 * 1024, "LDS");
constexpr int W5_MW = 8, W5_THREADS = 64 * W5_MW;
constexpr int W5_NK = 8;                                 // k-steps per unit (32 channels)
constexpr unsigned kW5Oob = 0x7ffffff0u;

__device__ __forceinline__ void w5_dma16_m0(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    // inside a bracket that saved M0 (s_nop 3 + the two instructions behind it: the 5 wait states between a VALU write of a scalar register and the VMEM read of it)
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
__device__ __forceinline__ unsigned w5u(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *w5u(const char *ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<const char *>(((unsigned long long)w5u((unsigned)(a >> 32)) << 32) | w5u((unsigned)a));
}
template <int N>
__device__ __forceinline__ void w5_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct W5Unit {
    int n, y0, x0;
};
__device__ __forceinline__ W5Unit w5_unit(const Wino5Params &p, unsigned u) {
    W5Unit r;
    const unsigned tx = u % (unsigned)p.tiles_x, t2 = u / (unsigned)p.tiles_x;
    r.y0 = (int)(t2 % (unsigned)p.tiles_y) * (2 * W5_TR), r.n = (int)(t2 / (unsigned)p.tiles_y);
    r.x0 = (int)tx * (2 * W5_TC);
    return r;
}

// The 1-D input transform B^T (6 -> 6) on the points {0, 1, -1, 2, -1/2, inf} (17 operations; the textbook points {0, +-1, +-2, inf} take 12 but
// leave 4.5e-6 of error on this layer where these leave 1.8e-6 and the direct fp32 sum 1.4e-6 -- the whole-network bound of 1e-3 on the warped
// frames is sensitive to that: tests/test_hip_timed_path.py):
//   B^T = [1 3/2 -2 -3/2 1 0; 0 -1 -5/2 -1/2 1 0; 0 1 1/2 -5/2 1 0; 0 -1/2 -1 1/2 1 0; 0 2 -1 -2 1 0; 0 1 3/2 -2 -3/2 1]
template <class T>
__device__ __forceinline__ void w5_bt_rows012(const T &d0, const T &d1, const T &d2, const T &d3, const T &d4, T &r0, T &r1, T &r2) {
    r0 = (d0 + (d4 - 2.f * d2)) + 1.5f * (d1 - d3);
    r1 = ((d4 - d1) - 2.5f * d2) - 0.5f * d3;
    r2 = ((d4 + d1) + 0.5f * d2) - 2.5f * d3;
}
template <class T>
__device__ __forceinline__ void w5_bt_rows345(const T &d1, const T &d2, const T &d3, const T &d4, const T &d5, T &r3, T &r4, T &r5) {
    const T cc = d4 - d2, s1 = d1 - d3;
    r3 = cc - 0.5f * s1, r4 = cc + 2.f * s1;
    r5 = (d1 + (d5 - 2.f * d3)) + 1.5f * (d2 - d4);
}
__device__ __forceinline__ void w5_bt6(const float c[6], float v[6]) {
    w5_bt_rows012(c[0], c[1], c[2], c[3], c[4], v[0], v[1], v[2]);
    w5_bt_rows345(c[1], c[2], c[3], c[4], c[5], v[3], v[4], v[5]);
}
}  // namespace

__global__ void __launch_bounds__(W5_THREADS, 2) wino5_first_kernel(const Wino5Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // unit assignment: as conv_ring.hip (XCD-contiguous chunk of the unit list, round-robin inside the XCD)
    const unsigned G = gridDim.x;
    const unsigned nxc = G < (unsigned)kXcds ? G : (unsigned)kXcds;
    const unsigned xcd = blockIdx.x % nxc, slot = blockIdx.x / nxc;
    const unsigned nx = G / nxc + (xcd < G % nxc ? 1u : 0u);
    const unsigned c_begin = w5u((unsigned)((unsigned long long)xcd * p.nunits / nxc));
    const unsigned c_end = w5u((unsigned)((unsigned long long)(xcd + 1) * p.nunits / nxc));
    if (c_begin + slot >= c_end) return;
    const unsigned u_begin = c_begin + slot, u_end = c_end, u_step = nx;
    const unsigned my_units = (u_end - u_begin + u_step - 1) / u_step;
    const unsigned total = my_units * (unsigned)W5_NK;
    const bool no_dma = p.ablate & 4, no_tf = p.ablate & 2;

    // The 8 waves walk the k-steps s = 0 .. total - 1 and meet at one barrier B_s in front of matrix phase s:
    //   before B_s : V_s written (transformed in interval s - 1), U_s and the raw planes of s + 1 landed;
    //   after  B_s : every wave issues its pieces of U_(s+2) and raw_(s+3) into the slots steps s - 1 / s left; four waves (one per SIMD,
    //                alternating with their SIMD's other wave from step to step) transform raw_(s+1) into V_(s+1); all multiply step s,
    //                wait for their own pieces of U_(s+1) / raw_(s+2) and arrive at B_(s+1).
    // ---- DMA pieces of this wave.  U: 36 wave-instructions per k-step, q = wv + 8 i (5 on waves 0-3, 4 on waves 4-7); raw planes: 5 per k-step,
    // q = wv - 4 on waves 4-7 and the half-filled fifth on wave 7
    const int n_u = wv < 4 ? 5 : 4;
    const int n_raw = wv < 4 ? 0 : (wv == 7 ? 2 : 1);
    unsigned r_off[2];   // lane's byte offset from the halo's first element of the chunk's first plane
    int r_geo[2];        // row | column slot << 8 | plane << 16, or -1 (slot beyond the chunk)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (i == 0 ? (wv & 3) : 4) * 64 + lane;
        const int c4 = j / (W5_PLANE / 4), rem = j - c4 * (W5_PLANE / 4);
        const int row = rem / (W5_IW / 4), cs = rem - row * (W5_IW / 4);
        const bool ok = j < W5_CHUNK_SLOTS;
        r_off[i] = ok ? (unsigned)(((c4 * p.H + row) * p.W + cs * 4) * 4) : kW5Oob;
        r_geo[i] = ok ? (row | cs << 8 | c4 << 16) : -1;
    }
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, W5_NK * W5_U_BYTES, 0x00020000);
    const unsigned img_bytes = (unsigned)p.C * (unsigned)p.H * (unsigned)p.W * 4u;
    const unsigned lane16 = (unsigned)lane * 16u;

    // raw cursor (runs 3 steps ahead of the matrix phase)
    unsigned ru = u_begin;
    int rk = 0;
    W5Unit RU = w5_unit(p, ru);
    auto issue_raw = [&](unsigned rslot) {   // the chunk of cursor (RU, rk) into raw slot rslot; advances the cursor
        if (!no_dma && n_raw) {
            const char *base = w5u(reinterpret_cast<const char *>(p.src + (size_t)RU.n * p.sstride));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)img_bytes, 0x00020000);
            const int oy = RU.y0 - 2, ox = RU.x0 - 4, c0 = rk * 4;
            const unsigned s_unit = w5u((unsigned)(((c0 * p.H + oy) * p.W + ox) * 4));
            const bool interior = oy >= 0 && oy + W5_IH <= p.H && ox >= 0 && ox + W5_IW <= p.W && c0 + 4 <= p.C;   // scalar
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && n_raw < 2) continue;
                unsigned v = r_off[i] + s_unit;   // (a slot beyond the chunk stays out of range: ~2^31 + an offset inside one sample)
                if (!interior) {
                    const int iy = oy + (r_geo[i] & 0xff), ix = ox + ((r_geo[i] >> 8) & 0xff) * 4, c = c0 + (r_geo[i] >> 16);
                    v = (r_geo[i] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.C) ? v : kW5Oob;
                }
                w5_dma16_m0(w5u((unsigned)(rslot * W5_RAW_SLOT + (i == 0 ? (wv & 3) : 4) * 1024)), v, rsrc, 0u);
            }
        }
        if (++rk == W5_NK) {
            rk = 0, ru += u_step;
            if (ru < u_end) RU = w5_unit(p, ru);
        }
    };
    auto issue_u = [&](unsigned uslot, int k) {
        if (no_dma) return;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && n_u < 5) continue;
            const int q = wv + i * W5_MW;
            w5_dma16_m0(w5u((unsigned)(W5_U_OFF + uslot * W5_U_BYTES + q * 1024)), lane16, rsrc_u, w5u((unsigned)(k * W5_U_BYTES + q * 1024)));
        }
    };
    // ---- input transform task of this wave (when it is its turn): (tile block tt = (wv & 3) >> 1, row half = wv & 1); lane = (kq = lane >> 4,
    // tile = lane & 15 of the block: 2 tile rows x 8).  Rows 3 half .. 3 half + 2 of B^T d B, all 6 columns.
    const int tt = (wv & 3) >> 1, half = wv & 1;
    const unsigned t_rd = (unsigned)((lane >> 4) * W5_PLANE * 4 + ((2 * (2 * tt + ((lane & 15) >> 3)) + half) * W5_IW + 2 + 2 * (lane & 7)) * 4);
    const unsigned t_wr = (unsigned)(W5_V_OFF + tt * 1024 + lane * 16);
    auto transform = [&](unsigned rslot, unsigned vslot) {
        if (no_tf) return;
        const unsigned char *src = lds + rslot * W5_RAW_SLOT + t_rd;
        f32x2 d[5][3];   // rows half .. half + 4 of the patch, columns in pairs
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) d[r][jj] = *reinterpret_cast<const f32x2 *>(src + (r * W5_IW + 2 * jj) * 4);
        f32x2 t[3][3];   // rows 3 half .. 3 half + 2 of B^T d
        if (half == 0) {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w5_bt_rows012(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], t[0][jj], t[1][jj], t[2][jj]);
        } else {   // d[r] = patch row r + 1
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w5_bt_rows345(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], t[0][jj], t[1][jj], t[2][jj]);
        }
        float v[18];   // components 18 half .. 18 half + 17
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float c[6] = {t[i][0].x, t[i][0].y, t[i][1].x, t[i][1].y, t[i][2].x, t[i][2].y};
            w5_bt6(c, v + i * 6);
        }
        // V[group of 4 components][tile block][lane][4]: a lane's operands of four matrix instructions are ONE 16-byte read; this wave's 18
        // components are four whole groups and half of group 4 (components 16, 17 from the rows 0-2 wave, 18, 19 from the rows 3-5 wave)
        unsigned char *dst = lds + t_wr + vslot * W5_V_BYTES;
        if (half == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(dst + g * 2048) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *reinterpret_cast<f32x2 *>(dst + 4 * 2048) = (f32x2){v[16], v[17]};
        } else {
            *reinterpret_cast<f32x2 *>(dst + 4 * 2048 + 8) = (f32x2){v[0], v[1]};
#pragma unroll
            for (int g = 5; g < 9; ++g) *reinterpret_cast<f32x4 *>(dst + g * 2048) = (f32x4){v[4 * g - 18], v[4 * g - 17], v[4 * g - 16], v[4 * g - 15]};
        }
    };

    // ---- matrix phase: wave wv = (channel block cb = wv & 3, tile block tb = wv >> 2): A = U[component][cb][kq][16 channels],
    // B = V[component][tb][kq][16 tiles], both lane-linear; acc[component][r] = channel cb * 16 + 4 (lane >> 4) + r of tile tb * 16 + (lane & 15)
    const int cb = wv & 3, tb = wv >> 2;
    const unsigned a_lane = (unsigned)(W5_U_OFF + cb * 1024 + lane * 16), b_lane = (unsigned)(W5_V_OFF + tb * 1024 + lane * 16);
    f32x4 acc[36];
    auto phase = [&](auto FIRST_, unsigned s3, unsigned s2) {
        constexpr bool FIRST = decltype(FIRST_)::value;
        const unsigned char *ua = lds + a_lane + s3 * W5_U_BYTES, *vb = lds + b_lane + s2 * W5_V_BYTES;
        constexpr int NG = 9, PD = 2;   // groups of 4 components; groups requested ahead of their matrix instructions
        f32x4 au[NG], bv[NG];
        auto rd = [&](int g) {
            au[g] = *reinterpret_cast<const f32x4 *>(ua + g * 4096);
            bv[g] = *reinterpret_cast<const f32x4 *>(vb + g * 2048);
        };
#pragma unroll
        for (int g = 0; g < PD; ++g) rd(g);
        if (!(p.ablate & 32)) __builtin_amdgcn_s_setprio(2);   // the multiplying wave first at its SIMD's issue port (287 -> 279 us; PWS_OPT_EXPERIMENT 1332: without)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + PD < NG) rd(g + PD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * g + e;
                if constexpr (FIRST) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[g][e], bv[g][e], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[g][e], bv[g][e], acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(p.ablate & 32)) __builtin_amdgcn_s_setprio(0);
    };

    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    issue_raw(0);
    issue_raw(1);
    issue_u(0, 0);
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
    w5_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // P: raw_0 is there
    if ((wv >> 2) == 1) transform(0, 0);
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    issue_u(1, 1);
    issue_raw(2);
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");

    // ---- the unit loop.  The two waves of a SIMD run the SAME k-step in OPPOSITE order (round 5, second cut): waves 0-3 multiply first and do
    // everything else -- epilogue, DMA issue, their turn at the transform -- behind it; waves 4-7 do all that first and multiply last.  One of
    // the two is in its matrix phase at any time.  (First cut, both in the same order: the timing-only ablations ADDED UP -- 123 us of matrix
    // instructions + 55 DMA issue + 31 transform + 33 epilogue + 52 barriers and waits = the 288 us measured.)
    // k is a compile-time constant and the two orders are two separate loops: a unit is straight-line code -- the first step's matrix
    // instructions (C = 0) define the accumulators, the other seven update them, nothing merges two forms of them at a control-flow join (merged,
    // hipcc spilled ~250 registers or moved all 144 through copies at every loop edge).
    auto run = [&](auto MF_) {
        constexpr bool MF = decltype(MF_)::value;   // matrix phase first
        unsigned s = 0, s3 = 0;                     // k-step, s % 3  (s % 2 = k % 2, s % 8 = k: 8 k-steps per unit)
        for (unsigned cu = u_begin; cu < u_end; cu += u_step) {
            auto epilogue = [&]() {
                if (p.ablate & 8) return;
                const W5Unit CU = w5_unit(p, cu);
                // ---- Y = A^T M A, A^T = [[1,1,1,1,1,0],[0,1,-1,2,-1/2,1]]: rows first (over j), then columns (over i)
                f32x4 z[6][2];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const f32x4 *m = acc + i * 6;
                    z[i][0] = ((m[0] + m[1]) + (m[2] + m[3])) + m[4];
                    z[i][1] = ((m[1] - m[2]) + (2.f * m[3] - 0.5f * m[4])) + m[5];
                }
                f32x4 y[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    y[0][b] = ((z[0][b] + z[1][b]) + (z[2][b] + z[3][b])) + z[4][b];
                    y[1][b] = ((z[1][b] - z[2][b]) + (2.f * z[3][b] - 0.5f * z[4][b])) + z[5][b];
                }
                const int co = cb * 16 + 4 * (lane >> 4);
                f32x4 bs = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bs = *reinterpret_cast<const f32x4 *>(p.bias + co);
                const int t16 = lane & 15;
                const int oy = CU.y0 + 2 * (2 * tb + (t16 >> 3)), ox = CU.x0 + 2 * (t16 & 7);
                float *o = p.out + ((size_t)(CU.n * p.H + oy) * p.W + ox) * p.out_ld + co;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 v = y[a][b] + bs;
                        v.x = act_apply(v.x, p.act), v.y = act_apply(v.y, p.act), v.z = act_apply(v.z, p.act), v.w = act_apply(v.w, p.act);
                        *reinterpret_cast<f32x4 *>(o + ((size_t)a * p.W + b) * p.out_ld) = v;
                    }
            };
            auto step = [&](auto KC) {
                constexpr int k = decltype(KC)::value;
                const unsigned sp1_3 = s3 == 2 ? 0u : s3 + 1, sp2_3 = sp1_3 == 2 ? 0u : sp1_3 + 1;
                if constexpr (MF) {
                    if (!(p.ablate & 1)) phase(std::integral_constant<bool, k == 0>{}, s3, (unsigned)(k & 1));
                    if constexpr (k == W5_NK - 1) epilogue();
                }
                asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
                if (s + 2 < total) issue_u(sp2_3, (k + 2) & 7);
                if (s + 3 < total) issue_raw(s3);
                asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
                if (s + 1 < total && (wv >> 2) == (k & 1)) transform(sp1_3, (unsigned)((k & 1) ^ 1));
                if constexpr (!MF)
                    if (!(p.ablate & 1)) phase(std::integral_constant<bool, k == 0>{}, s3, (unsigned)(k & 1));
                if (s + 1 < total) {
                    // own pieces of U_(s+1) / raw_(s+2) have landed: at most this interval's pieces still fly (loads return in order; the
                    // epilogue's stores, if they still fly, only make this wait longer)
                    if (s + 3 < total && !no_dma) {
                        if (wv == 7) w5_wait_vmcnt<6>();
                        else w5_wait_vmcnt<5>();
                    } else {
                        w5_wait_vmcnt<0>();
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // B_(s+1)
                    asm volatile("" ::: "memory");
                }
                ++s, s3 = sp1_3;
            };
            step(std::integral_constant<int, 0>{}), step(std::integral_constant<int, 1>{}), step(std::integral_constant<int, 2>{}), step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}), step(std::integral_constant<int, 5>{}), step(std::integral_constant<int, 6>{}), step(std::integral_constant<int, 7>{});
            if constexpr (!MF) epilogue();   // (behind the barrier: beside the other waves' first matrix phase of the next unit)
        }
    };
    if (wv < 4 && !(p.ablate & 16)) run(std::true_type{});
    else run(std::false_type{});
}

// Runs the first-layer launch described by kp on the Winograd kernel when it is covered: fp32, NCHW source of 17 .. 32 channels with
// 16-byte aligned rows, exactly 64 output channels, whole 8 x 16 units, transformed weights given (pws_conv_args.w_wring), enough units for
// the chip (PWS_OPT_EXPERIMENT 26: never; 30: also for the tests' small launches).  Returns 1 when not covered (conv_first_kernel is next).
int wino5_first_try(const ConvKParams &kp, const float *u, hipStream_t st, const ProfInfo &pi) {
    if (!u || kp.io_bf16 || kp.ndst != 0 || kp.nsrc != 1 || g_experiment == 26 || g_experiment == 25) return 1;
    if (kp.src_c[0] > 32 || kp.cin_pad != 32 || kp.cout != 64 || kp.W % (2 * W5_TC) != 0 || kp.H % (2 * W5_TR) != 0 || kp.out_ld % 4 != 0) return 1;
    const size_t sstride = kp.src_ld[0] ? (size_t)kp.src_ld[0] : (size_t)kp.src_c[0] * kp.H * kp.W;
    if ((reinterpret_cast<size_t>(kp.src_ptr[0]) & 15) || sstride % 4 != 0 || (reinterpret_cast<size_t>(kp.out) & 15) ||
        (kp.bias && (reinterpret_cast<size_t>(kp.bias) & 15)) || (reinterpret_cast<size_t>(u) & 15))
        return 1;
    if ((size_t)kp.src_c[0] * kp.H * kp.W * 4 >= (1u << 30)) return 1;   // kW5Oob (+ a unit's scalar offset) must stay beyond the descriptor's records and below 2^32
    Wino5Params p{};
    p.src = static_cast<const float *>(kp.src_ptr[0]), p.C = kp.src_c[0], p.N = kp.N, p.H = kp.H, p.W = kp.W, p.sstride = sstride;
    p.u = u, p.bias = kp.bias, p.act = kp.act, p.out = static_cast<float *>(kp.out), p.out_ld = kp.out_ld;
    p.tiles_x = kp.W / (2 * W5_TC), p.tiles_y = kp.H / (2 * W5_TR);
    p.nunits = (unsigned)(p.tiles_x * p.tiles_y) * (unsigned)kp.N;
    p.ablate = g_experiment >= 1300 && g_experiment < 1364 ? g_experiment - 1300 : 0;
    static PerDeviceInt ncu_dev;
    int &ncu = ncu_dev.cur();
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (p.nunits < (unsigned)(2 * ncu) && g_experiment != 30) return 1;   // a unit is short (8 k-steps): the pipeline wants a few per workgroup
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino5_first_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W5_LDS);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wino5_first_kernel, %d B LDS): %s", W5_LDS, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    const unsigned grid = p.nunits < (unsigned)ncu ? p.nunits : (unsigned)ncu;   // one persistent workgroup per CU
    ProfScope prof(KID_CONV_FIRST_WINO, pi.flops, pi.bytes, st);
    hipLaunchKernelGGL(wino5_first_kernel, dim3(grid), dim3(W5_THREADS), W5_LDS, st, p);
    return check_launch("wino5_first_kernel");
}

}  // namespace pws

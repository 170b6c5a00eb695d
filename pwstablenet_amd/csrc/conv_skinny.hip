// One-shot fp32 convolution for the deep, small maps of the generator (at most 128 output pixels per launch and parity class:
// the 4x4 .. 1x1 levels of a batch of 8 -- BASELINE configs[1]).  These launches are a skinny GEMM  out[M][cout] = A[M][K] W[K][cout]
// with M <= 128 and K = taps x cin up to 16 384: 0.04 - 1 GFLOP against 9 - 34 MB of weights, i.e. bound by the latency of the
// weight stream, not by arithmetic.  conv_mfma_kernel's small tiles walk their K split chunk by chunk (global -> VGPR -> LDS ->
// barrier -> matrix instructions, one exposed memory latency per 16-channel chunk: 20 - 27 us per layer, 31 such launches per
// forward).  Here a workgroup owns 64 output channels x one K split of at most 144 KB of weights and fetches ALL of it with one
// burst of LDS-DMA instructions (buffer_load_dwordx4 ... lds) issued before anything else: one memory latency per launch.
//   * B operand (weights) from LDS in the lane-linear image of v_mfma_f32_16x16x4_f32: per (tap, 16-channel chunk) a 4 KB block
//     [k-step st][n-tile nt][kq][16 cout], so that a ds_read_b32 of lane (kq, l15) is conflict-free; a DMA instruction fetches four
//     whole 256-byte rows of the packed layout [tap][cin][cout] (the permutation is in the per-lane source offset);
//   * A operand straight from global memory / L2 (the activations of these levels are < 2 MB): lane (pixel m = l15 of its
//     16-pixel tile, kq) loads the 4 channels kq * 4 .. + 3 of its pixel's tap as one 16-byte load = the four k-steps;
//     the next chunk's loads are in flight during the matrix instructions of the current one;
//   * the 4 waves split the M tiles and / or the N tiles by the launch's M (template), so that no wave multiplies padding:
//     M <= 16: 1 M tile, wave = N tile;  M <= 32: 2 x 2;  M <= 128: wave = M tiles w and w + 4, all 4 N tiles;
//   * K is split over workgroups until the grid fills the chip; partial sums go to the caller's workspace and
//     splitk_reduce_kernel (conv_mfma.hip) adds them in split order (deterministic) and applies bias + activation.
// Arithmetic: exact fp32 products and sums (v_mfma_f32_16x16x4_f32); the summation order differs from conv_mfma_kernel's.
#include "conv_common.h"

namespace pws {

enum SkinnyMode { SK_K3S1 = 0, SK_K3S2 = 1, SK_CT4 = 2 };

struct SkParams {
    const float *src_ptr[4];
    int src_c[4], src_ld[4];
    int nsrc;
    int N, H, W;      // input
    int LH, LW;       // logical output extent of one class
    int OH, OW;       // output tensor extent
    int mode;
    int cin_pad, cout;
    const float *w;   // packed [plane][cin_pad][cout]
    unsigned w_bytes;
    const float *bias;
    int act;
    float *out;       // ksplit == 1: the final output (bias + activation applied here); else the partial buffers
    int out_ld;
    size_t split_stride;
    int ksplit, cps, nchunks;
    int M;            // N * LH * LW
    unsigned ncob;
};

constexpr unsigned kSkOob = 0x7ffffff0u;
__device__ __forceinline__ void sk_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
template <class T>
__device__ __forceinline__ T sksel4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

template <int MT_W, int NT_W, int WM>
__global__ void __launch_bounds__(256) conv_skinny_kernel(const SkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WN = 4 / WM;
    static_assert(NT_W * WN == 4, "4 N tiles of 16 output channels per workgroup");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wm = wv % WM, wn = wv / WM;

    unsigned b = blockIdx.x;
    const int ks = (int)(b % (unsigned)p.ksplit);
    b /= (unsigned)p.ksplit;
    const int cob = (int)(b % p.ncob), cls = (int)(b / p.ncob);
    const int py = cls >> 1, px = cls & 1;
    const int co0 = cob * 64;
    const int c_begin = ks * p.cps;
    const int nck = min(p.nchunks - c_begin, p.cps);
    const int ntaps = p.mode == SK_CT4 ? 4 : 9;

    // ---- all the weights of this workgroup: nck x ntaps blocks of 4 KB = 4 DMA instructions each, dealt to the 4 waves
    {
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.w_bytes, 0x00020000);
        const int nt = lane >> 4 >> 0 & 3, kqd = (lane >> 2) & 3, c4 = lane & 3;   // LDS slot `lane` of a piece = (nt, kq, 4 couts)
        const int col = co0 + nt * 16 + c4 * 4;
        const bool col_ok = col < p.cout;
        const size_t plane = (size_t)p.cin_pad * p.cout;
        const int npieces = nck * ntaps * 4;
        for (int q = wv; q < npieces; q += 4) {
            const int blk = q >> 2, st = q & 3;
            const int ck = blk / ntaps, tap = blk - ck * ntaps;
            const int wplane = p.mode == SK_CT4 ? cls * 4 + tap : tap;
            const unsigned row = (unsigned)((c_begin + ck) * 16 + kqd * 4 + st);
            const unsigned voff = col_ok ? (row * (unsigned)p.cout + (unsigned)col) * 4u : kSkOob;
            sk_dma16((unsigned)(blk * 4096 + st * 1024), voff, rsrc_w, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((size_t)wplane * plane * 4)));
        }
    }

    // ---- this lane's pixels: M tile mt = wm + i * WM, pixel m = mt * 16 + l15 -> (sample, first input row / column of its taps)
    int pn[MT_W], piy[MT_W], pix_[MT_W];
    bool pok[MT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
        const int m = (wm + i * WM) * 16 + l15;
        pok[i] = m < p.M;
        const int mm = pok[i] ? m : 0;
        const int ox = mm % p.LW, t2 = mm / p.LW;
        const int oy = t2 % p.LH;
        pn[i] = t2 / p.LH;
        if (p.mode == SK_K3S2) piy[i] = 2 * oy - 1, pix_[i] = 2 * ox - 1;
        else if (p.mode == SK_CT4) piy[i] = oy + py - 1, pix_[i] = ox + px - 1;
        else piy[i] = oy - 1, pix_[i] = ox - 1;
    }
    // source cursor of chunk c_begin
    int s = 0, c0 = c_begin * 16;
    while (s < p.nsrc - 1 && c0 >= sksel4(p.src_c, s)) c0 -= sksel4(p.src_c, s), ++s;

    constexpr int MAXT = 9;
    f32x4 a_cur[MAXT][MT_W], a_nxt[MAXT][MT_W];
    auto load_a = [&](f32x4 (&a)[MAXT][MT_W]) {
        const float *sp = sksel4(p.src_ptr, s) + c0 + kq * 4;
        const int ld = sksel4(p.src_ld, s);
        const int kw = p.mode == SK_CT4 ? 2 : 3;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntaps) {
                const int dy = t / kw, dx = t - dy * kw;
#pragma unroll
                for (int i = 0; i < MT_W; ++i) {
                    const int iy = piy[i] + dy, ix = pix_[i] + dx;
                    const bool ok = pok[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    a[t][i] = ok ? *reinterpret_cast<const f32x4 *>(sp + ((size_t)(pn[i] * p.H + iy) * p.W + ix) * ld) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        c0 += 16;
        if (c0 >= sksel4(p.src_c, s) && s < p.nsrc - 1) ++s, c0 = 0;
    };

    f32x4 acc[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_a(a_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every DMA piece of this wave has landed (they are older than the loads above)
    __syncthreads();
    const int b_lane = (kq * 16 + l15) * 4;   // + ((blk * 4 + st) * 4 + nt) * 256
    for (int ck = 0; ck < nck; ++ck) {
        if (ck + 1 < nck) load_a(a_nxt);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntaps) {
                const unsigned char *bb = lds + (unsigned)((ck * ntaps + t) * 4096) + b_lane;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    float bv[NT_W];
#pragma unroll
                    for (int j = 0; j < NT_W; ++j) bv[j] = *reinterpret_cast<const float *>(bb + (st * 4 + wn * NT_W + j) * 256);
#pragma unroll
                    for (int i = 0; i < MT_W; ++i)
#pragma unroll
                        for (int j = 0; j < NT_W; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t][i][st], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (ck + 1 < nck) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int i = 0; i < MT_W; ++i) a_cur[t][i] = a_nxt[t][i];
        }
    }

    // ---- results: lane (l15, kq) holds D[pixel 4 kq + r of the tile][cout nt * 16 + l15]
    float *out = p.out + (size_t)ks * p.split_stride;
    const bool fin = p.ksplit == 1;
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (wm + i * WM) * 16 + 4 * kq + r;
            if (m >= p.M) continue;
            const int ox = m % p.LW, t2 = m / p.LW;
            const int oy = t2 % p.LH, n = t2 / p.LH;
            size_t opix;
            if (p.mode == SK_CT4) opix = ((size_t)n * p.OH + 2 * oy + py) * p.OW + 2 * ox + px;
            else opix = ((size_t)n * p.OH + oy) * p.OW + ox;
#pragma unroll
            for (int j = 0; j < NT_W; ++j) {
                const int co = co0 + (wn * NT_W + j) * 16 + l15;
                if (co < p.cout) {
                    if (fin) p.out[opix * p.out_ld + co] = act_apply(acc[i][j][r] + (p.bias ? p.bias[co] : 0.f), p.act);
                    else out[opix * p.cout + co] = acc[i][j][r];
                }
            }
        }
    }
}

template <int MT_W, int NT_W, int WM>
static int skinny_launch(const SkParams &p, unsigned grid, int lds_bytes, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_skinny_kernel<MT_W, NT_W, WM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_skinny_kernel): %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_skinny_kernel<MT_W, NT_W, WM>), dim3(grid), dim3(256), lds_bytes, st, p);
    return check_launch("conv_skinny_kernel");
}

// Runs the fp32 forward launch described by kp (prepared by conv2d_fwd_impl, conv_mfma.hip) on the one-shot kernel when it is
// covered: NHWC fp32 sources in multiples of 16 channels, at most 128 output pixels per parity class, cout a multiple of 4, and a
// workspace for the K split.  Returns 1 when not covered (the caller runs conv_mfma_kernel's small tiles).
int conv_skinny_try(int kind, ConvKParams &kp, float *final_out, float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi) {
    if (kp.io_bf16 || kp.ndst != 0 || g_experiment == 70) return 1;
    int mode;
    if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) mode = SK_K3S1;
    else if (kind == PWS_CONV_K3S2) mode = SK_K3S2;
    else if (kind == PWS_CONVT_K4S2) mode = SK_CT4;
    else return 1;
    const long M = (long)kp.N * kp.LH * kp.LW;
    if (M < 1 || M > 128 || kp.cout % 4 != 0) return 1;
    int cin = 0;
    for (int s = 0; s < kp.nsrc; ++s) {
        if (kp.src_ld[s] == 0 || kp.src_c[s] % 16 != 0 || kp.src_ld[s] % 4 != 0 || (reinterpret_cast<size_t>(kp.src_ptr[s]) & 15)) return 1;
        cin += kp.src_c[s];
    }
    if (cin != kp.cin_pad) return 1;
    const int ntaps = mode == SK_CT4 ? 4 : 9, ncls = mode == SK_CT4 ? 4 : 1;
    const size_t w_bytes = (size_t)(mode == SK_CT4 ? 16 : 9) * kp.cin_pad * kp.cout * 4;
    if (w_bytes >= (1u << 31)) return 1;
    SkParams p{};
    for (int s = 0; s < 4; ++s) p.src_ptr[s] = kp.src_ptr[s], p.src_c[s] = kp.src_c[s], p.src_ld[s] = kp.src_ld[s];
    p.nsrc = kp.nsrc, p.N = kp.N, p.H = kp.H, p.W = kp.W, p.LH = kp.LH, p.LW = kp.LW, p.OH = kp.OH, p.OW = kp.OW, p.mode = mode;
    p.cin_pad = kp.cin_pad, p.cout = kp.cout, p.w = kp.w, p.w_bytes = (unsigned)w_bytes, p.bias = kp.bias, p.act = kp.act, p.out_ld = kp.out_ld;
    p.M = (int)M, p.ncob = (unsigned)((kp.cout + 63) / 64);
    p.nchunks = cin / 16;
    // K split: at most 144 KB of weights per workgroup, and enough workgroups for ~2 per CU (the weight stream of a launch then
    // arrives in two waves of bursts instead of one long one per CU), limited by the workspace
    const int maxc = 36 / ntaps;                    // chunks whose weights fit the LDS: 4 (3x3) / 9 (2x2)
    const long blocks1 = (long)p.ncob * ncls;
    int ksplit = (int)cdiv(512, blocks1);
    if (ksplit > p.nchunks) ksplit = p.nchunks;
    const int kmin = (int)cdiv(p.nchunks, maxc);
    if (ksplit < kmin) ksplit = kmin;
    const size_t out_floats = (size_t)kp.N * kp.OH * kp.OW * kp.cout;
    if (ksplit > 1 && (!ws || (size_t)ksplit * out_floats > ws_floats)) {
        if (!ws || (size_t)kmin * out_floats > ws_floats) return 1;
        ksplit = (int)(ws_floats / out_floats);
    }
    p.cps = (int)cdiv(p.nchunks, ksplit);
    p.ksplit = (int)cdiv(p.nchunks, p.cps);   // no empty splits
    if (p.cps > maxc) return 1;
    p.split_stride = out_floats;
    p.out = p.ksplit > 1 ? ws : final_out;
    const unsigned grid = (unsigned)(blocks1 * p.ksplit);
    const int lds_bytes = p.cps * ntaps * 4096;
    if (g_experiment == 71) return PWS_OK;   // TIMING ONLY: the launch is skipped (what the deep levels cost end to end)
    ProfScope prof(KID_CONV_SKINNY, pi.flops, pi.bytes, st);   // covers the split-K reduce as well
    int rc;
    if (M <= 16) rc = skinny_launch<1, 1, 1>(p, grid, lds_bytes, st);
    else if (M <= 32) rc = skinny_launch<1, 2, 2>(p, grid, lds_bytes, st);
    else rc = skinny_launch<2, 4, 4>(p, grid, lds_bytes, st);
    if (rc != PWS_OK || p.ksplit == 1) return rc;
    kp.ksplit = p.ksplit, kp.split_stride = out_floats, kp.out = final_out;
    return launch_splitk_reduce(kp, ws, out_floats / 4, st);
}

}  // namespace pws

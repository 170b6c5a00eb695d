// One-shot bf16 convolution for the deep, small maps (8 x 8 .. 1 x 1) with bf16 activation storage: forward AND data-gradient kinds
// of BASELINE configs[2] / [3] (batch 64: 64 .. 4096 output pixels per launch and parity class) and of bf16 inference.
//
// Why (tools/layer_profile.py --batch 64 --math bf16, profiles/r02_kernel_stats_configs2_bf16.csv): these launches are skinny GEMMs
// out[M][cout] = A[M][K] W[K][cout] with K = taps x cin up to 16 384 and 0.3 - 19 GFLOP each, which conv_bf16_kernel's small tiles ran
// at 14 - 200 TFLOP/s: 20 - 60 us per launch whatever its size, because a workgroup walks its K split chunk by chunk (global -> VGPR
// -> LDS -> barrier -> matrix instructions: one exposed memory latency per 16 / 32-channel chunk).  72 such launches and 42 K-split
// reductions cost 3.9 ms of the 35 ms configs[2] step.  This is csrc/conv_skinny.hip (fp32, batch 8) for bf16:
//   * a workgroup owns (M block of <= 128 pixels) x 64 output channels x one K split of <= 144 KB of bf16 weights, fetched with ONE
//     burst of LDS-DMA instructions issued before anything else; the packed bf16 layout [plane][cout][k] is already the B operand's:
//     a 1 KB piece = 16 output channels x 32 input channels of one tap (rows of 64 bytes, their four 16-byte slots XOR-swizzled by
//     bits 2..3 of the row against bank conflicts, as in conv_ring.hip);
//   * A operands straight from global memory / L2: lane (pixel l15 of its 16-pixel tile, kq) loads the 8 channels 8 kq .. of its pixel's
//     tap as one 16-byte load = its share of a v_mfma_f32_16x16x32_bf16; the next chunk's loads fly during the current chunk's matrix
//     instructions;
//   * every kind of the bf16 path: taps (ky, kx) of a KS x KS window at input pixel (S y - pad_y + ky, S x - pad_x + kx) -- 3x3 s1,
//     3x3 s2, 4x4 s2 (data gradient of the transposed layers), and the two sub-pixel kinds with 4 parity classes of 2x2 taps
//     (transposed forward: pad = 1 - parity; data gradient of 3x3 s2: pad 0 and only the taps (ty <= py, tx <= px));
//   * K is split over workgroups until the grid fills the chip; raw partial sums go to the caller's workspace and
//     splitk_reduce_kernel (conv_mfma.hip) runs the ordinary epilogue (bias + activation, or the data gradient's scatter /
//     accumulate / act') in bf16 storage.
// Arithmetic is conv_bf16_kernel's: bf16 x bf16 products exact in fp32, fp32 accumulation; the summation order differs.
#include "conv_common.h"

namespace pws {

struct Sk16Params {
    const void *src_ptr[4];   // bf16 NHWC sources of the virtual concat (multiples of 32 channels)
    int src_c[4], src_ld[4];
    int nsrc;
    int N, H, W;      // input
    int LH, LW;       // logical output extent of one class
    int OH, OW;       // output tensor extent
    int ks, stride, pad, subpix;   // window, stride, padding (subpix 0); subpix 1 / 2: the parity-class kinds
    int kpad, npad, cout;
    const void *w;    // [plane][npad][kpad] bf16
    unsigned w_bytes;
    float *out;       // partial buffers [split][output pixel][cout]
    size_t split_stride;
    int ksplit, cps, nchunks;   // chunks of 32 input channels
    int M;            // N * LH * LW
    unsigned ncob, nmb;
};

__device__ __forceinline__ void sk16_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
template <class T>
__device__ __forceinline__ T sk16sel4(const T (&a)[4], int i) {
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// MT_W M tiles (16 pixels) and NT_W N tiles (16 output channels) per wave, WM waves along M; MAXT >= taps of the kind
template <int MT_W, int NT_W, int WM, int MAXT>
__global__ void __launch_bounds__(256) conv_skinny16_kernel(const Sk16Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WN = 4 / WM;
    static_assert(NT_W * WN == 4, "4 N tiles of 16 output channels per workgroup");
    constexpr int MB = MT_W * WM * 16;   // pixels per workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wm = wv % WM, wn = wv / WM;

    unsigned b = blockIdx.x;
    const int ks = (int)(b % (unsigned)p.ksplit);
    b /= (unsigned)p.ksplit;
    const int mb = (int)(b % p.nmb);
    b /= p.nmb;
    const int cob = (int)(b % p.ncob), cls = (int)(b / p.ncob);
    const int py = cls >> 1, px = cls & 1;
    const int co0 = cob * 64;
    const int c_begin = ks * p.cps;
    const int nck = min(p.nchunks - c_begin, p.cps);
    const int ntaps = p.ks * p.ks;
    const int pad_y = p.subpix == 1 ? 1 - py : (p.subpix == 2 ? 0 : p.pad), pad_x = p.subpix == 1 ? 1 - px : (p.subpix == 2 ? 0 : p.pad);
    // taps this class multiplies (bit t): all, or (ty <= py, tx <= px) for the data gradient of the stride-2 layers
    unsigned tapmask = (1u << ntaps) - 1u;
    if (p.subpix == 2) tapmask = py ? (px ? 0xfu : 0x5u) : (px ? 0x3u : 0x1u);

    // ---- all the weights of this workgroup: nck x taps x 4 pieces of 1 KB (16 output channels x 32 input channels), dealt to the 4 waves
    {
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, (int)p.w_bytes, 0x00020000);
        const int r = lane >> 2, s = lane & 3;   // LDS slot `lane` of a piece = row r (output channel), 16-byte slot s
        const unsigned voff = (unsigned)(r * p.kpad * 2 + ((s ^ ((r >> 2) & 3)) << 4));
        const int npieces = nck * ntaps * 4;
        for (int q = wv; q < npieces; q += 4) {
            const int blk = q >> 2, nt = q & 3;
            const int ck = blk / ntaps, tap = blk - ck * ntaps;
            if (!((tapmask >> tap) & 1u)) continue;   // wave-uniform
            const int plane = (p.subpix ? cls * ntaps : 0) + tap;
            const size_t soff = (((size_t)plane * p.npad + co0 + nt * 16) * p.kpad + (size_t)(c_begin + ck) * 32) * 2;
            sk16_dma16((unsigned)(q * 1024), voff, rsrc_w, (unsigned)__builtin_amdgcn_readfirstlane((unsigned)soff));
        }
    }

    // ---- this lane's pixels: M tile mt = wm + i * WM of the block, pixel m = mb * MB + mt * 16 + l15 -> (sample, first input row / column)
    int pn[MT_W], piy[MT_W], pix_[MT_W];
    bool pok[MT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
        const int m = mb * MB + (wm + i * WM) * 16 + l15;
        pok[i] = m < p.M;
        const int mm = pok[i] ? m : 0;
        const int ox = mm % p.LW, t2 = mm / p.LW;
        const int oy = t2 % p.LH;
        pn[i] = t2 / p.LH;
        piy[i] = oy * p.stride - pad_y, pix_[i] = ox * p.stride - pad_x;
    }
    // source cursor of chunk c_begin
    int s = 0, c0 = c_begin * 32;
    while (s < p.nsrc - 1 && c0 >= sk16sel4(p.src_c, s)) c0 -= sk16sel4(p.src_c, s), ++s;

    bf16x8 a_cur[MAXT][MT_W], a_nxt[MAXT][MT_W];
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
    auto load_a = [&](bf16x8 (&a)[MAXT][MT_W]) {
        const __bf16 *sp = static_cast<const __bf16 *>(sk16sel4(p.src_ptr, s)) + c0 + kq * 8;
        const int ld = sk16sel4(p.src_ld, s);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntaps && ((tapmask >> t) & 1u)) {
                const int dy = t / p.ks, dx = t - dy * p.ks;
#pragma unroll
                for (int i = 0; i < MT_W; ++i) {
                    const int iy = piy[i] + dy, ix = pix_[i] + dx;
                    const bool ok = pok[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    a[t][i] = ok ? *reinterpret_cast<const bf16x8 *>(sp + ((size_t)(pn[i] * p.H + iy) * p.W + ix) * ld) : zero8;
                }
            }
        }
        c0 += 32;
        if (c0 >= sk16sel4(p.src_c, s) && s < p.nsrc - 1) ++s, c0 = 0;
    };

    f32x4 acc[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_a(a_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every DMA piece of this wave has landed (they are older than the loads above)
    __syncthreads();
    const int b_lane = l15 * 64 + ((kq ^ ((l15 >> 2) & 3)) << 4);   // + ((ck * ntaps + tap) * 4 + nt) * 1024
    for (int ck = 0; ck < nck; ++ck) {
        if (ck + 1 < nck) load_a(a_nxt);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntaps && ((tapmask >> t) & 1u)) {
                const unsigned char *bb = lds + (unsigned)((ck * ntaps + t) * 4096) + b_lane;
                bf16x8 bv[NT_W];
#pragma unroll
                for (int j = 0; j < NT_W; ++j) bv[j] = *reinterpret_cast<const bf16x8 *>(bb + (wn * NT_W + j) * 1024);
#pragma unroll
                for (int i = 0; i < MT_W; ++i)
#pragma unroll
                    for (int j = 0; j < NT_W; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[t][i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if (ck + 1 < nck) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int i = 0; i < MT_W; ++i) a_cur[t][i] = a_nxt[t][i];
        }
    }

    // ---- raw partial sums: lane (l15, kq) holds D[pixel 4 kq + r of the tile][cout nt * 16 + l15]
    float *out = p.out + (size_t)ks * p.split_stride;
    const int so = p.subpix ? 2 : 1;
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mb * MB + (wm + i * WM) * 16 + 4 * kq + r;
            if (m >= p.M) continue;
            const int ox = m % p.LW, t2 = m / p.LW;
            const int oy = t2 % p.LH, n = t2 / p.LH;
            const int ty = so * oy + (p.subpix ? py : 0), tx = so * ox + (p.subpix ? px : 0);
            if (ty >= p.OH || tx >= p.OW) continue;
            const size_t opix = ((size_t)n * p.OH + ty) * p.OW + tx;
#pragma unroll
            for (int j = 0; j < NT_W; ++j) {
                const int co = co0 + (wn * NT_W + j) * 16 + l15;
                if (co < p.cout) out[opix * p.cout + co] = acc[i][j][r];
            }
        }
    }
}

template <int MT_W, int NT_W, int WM, int MAXT>
static int skinny16_launch(const Sk16Params &p, unsigned grid, int lds_bytes, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_skinny16_kernel<MT_W, NT_W, WM, MAXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_skinny16_kernel): %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_skinny16_kernel<MT_W, NT_W, WM, MAXT>), dim3(grid), dim3(256), lds_bytes, st, p);
    return check_launch("conv_skinny16_kernel");
}

// Runs the bf16 launch described by kp (prepared by conv2d_fwd_impl / conv2d_bwd_data_impl, conv_mfma.hip) on the one-shot kernel
// when it is covered: bf16 storage, a map of at most 8 x 8 pixels per class and at most 4096 pixels in all, sources in multiples of
// 32 channels, a workspace for the partial sums.  kchan: channels of the contraction (forward: cin; data gradient: the forward
// layer's cout).  Returns 1 when not covered (the caller runs conv_bf16_kernel's small tiles).  PWS_OPT_EXPERIMENT 71 switches it off.
int conv_skinny16_try(int kind, bool dgrad, ConvKParams &kp, int kchan, float *final_out, float *ws, size_t ws_floats, hipStream_t st,
                      const ProfInfo &pi) {
    if (!kp.io_bf16 || g_experiment == 71 || !ws) return 1;
    Sk16Params p{};
    if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) p.ks = 3, p.stride = 1, p.pad = 1, p.subpix = 0;
    else if (kind == PWS_CONV_K3S2 && !dgrad) p.ks = 3, p.stride = 2, p.pad = 1, p.subpix = 0;
    else if (kind == PWS_CONV_K3S2) p.ks = 2, p.stride = 1, p.pad = 0, p.subpix = 2;
    else if (kind == PWS_CONVT_K4S2 && !dgrad) p.ks = 2, p.stride = 1, p.pad = 0, p.subpix = 1;
    else if (kind == PWS_CONVT_K4S2) p.ks = 4, p.stride = 2, p.pad = 1, p.subpix = 0;
    else return 1;
    if (kp.LH > 8 || kp.LW > 8) return 1;
    const long M = (long)kp.N * kp.LH * kp.LW;
    // Measured at batch 64 (tools/layer_profile.py --batch 64 --math bf16 --train): 64 .. 256 pixels per class 16 - 30 us against 21 - 35 us
    // of conv_bf16_kernel, 1024 pixels +-10 % either way, 4096 pixels 2 - 3.5x SLOWER (an A operand gathered per tap and lane re-reads
    // the input nine times, and every 128-pixel block fetches the layer's weights again): small M only (PWS_OPT_EXPERIMENT 72: up to 4096)
    const long mmax = g_experiment == 72 ? 4096 : (g_experiment >= 730 && g_experiment <= 739 ? (64L << (g_experiment - 730)) : 256);
    if (M < 1 || M > mmax || kp.cout % 4 != 0 || kchan % 32 != 0) return 1;
    int cin = 0;
    for (int s = 0; s < kp.nsrc; ++s) {
        if (kp.src_ld[s] == 0 || kp.src_c[s] % 32 != 0 || kp.src_ld[s] % 8 != 0 || (reinterpret_cast<size_t>(kp.src_ptr[s]) & 15)) return 1;
        cin += kp.src_c[s];
    }
    if (cin != kchan || kchan > kp.kpad_bf) return 1;
    const int ntaps = p.ks * p.ks, ncls = p.subpix ? 4 : 1;
    const size_t w_bytes = (size_t)ncls * ntaps * kp.npad_bf * kp.kpad_bf * 2;
    if (w_bytes >= (1u << 31) || (size_t)64 * kp.kpad_bf * 2 >= (1u << 24)) return 1;
    for (int s = 0; s < 4; ++s) p.src_ptr[s] = kp.src_ptr[s], p.src_c[s] = kp.src_c[s], p.src_ld[s] = kp.src_ld[s];
    p.nsrc = kp.nsrc, p.N = kp.N, p.H = kp.H, p.W = kp.W, p.LH = kp.LH, p.LW = kp.LW, p.OH = kp.OH, p.OW = kp.OW;
    p.kpad = kp.kpad_bf, p.npad = kp.npad_bf, p.cout = kp.cout, p.w = kp.w_bf, p.w_bytes = (unsigned)w_bytes;
    p.M = (int)M, p.ncob = (unsigned)((kp.cout + 63) / 64);
    p.nchunks = kchan / 32;
    // 16 taps: 64-pixel blocks (the A operands of a chunk and of the next one are held in registers: 16 taps x 2 M tiles would not fit)
    const int mblock = ntaps == 16 ? 64 : (M <= 16 ? 16 : (M <= 32 ? 32 : 128));
    p.nmb = (unsigned)cdiv(M, mblock);
    // K split: at most 144 KB of weights per workgroup (4 KB per tap and chunk), and enough workgroups for ~2 per CU, limited by the workspace
    const int maxc = 36 / ntaps;   // 4 (3x3) / 9 (2x2) / 2 (4x4) chunks
    const long blocks1 = (long)p.ncob * ncls * p.nmb;
    int ksplit = (int)cdiv(512, blocks1);
    if (ksplit > p.nchunks) ksplit = p.nchunks;
    const int kmin = (int)cdiv(p.nchunks, maxc);
    if (ksplit < kmin) ksplit = kmin;
    const size_t out_floats = (size_t)kp.N * kp.OH * kp.OW * kp.cout;
    if ((size_t)ksplit * out_floats > ws_floats) {
        if ((size_t)kmin * out_floats > ws_floats) return 1;
        ksplit = (int)(ws_floats / out_floats);
    }
    p.cps = (int)cdiv(p.nchunks, ksplit);
    p.ksplit = (int)cdiv(p.nchunks, p.cps);   // no empty splits
    if (p.cps > maxc) return 1;
    p.split_stride = out_floats;
    p.out = ws;
    const unsigned grid = (unsigned)(blocks1 * p.ksplit);
    const int lds_bytes = p.cps * ntaps * 4096;
    ProfScope prof(KID_CONV_SKINNY16, pi.flops, pi.bytes, st);   // covers the split-K reduce as well
    int rc;
    if (ntaps == 16) rc = skinny16_launch<1, 4, 4, 16>(p, grid, lds_bytes, st);
    else if (M <= 16) rc = skinny16_launch<1, 1, 1, 9>(p, grid, lds_bytes, st);
    else if (M <= 32) rc = skinny16_launch<1, 2, 2, 9>(p, grid, lds_bytes, st);
    else rc = skinny16_launch<2, 4, 4, 9>(p, grid, lds_bytes, st);
    if (rc != PWS_OK) return rc;
    // the ordinary epilogue (bias + activation / the data gradient's scatter) runs in the reduce, also for a single split
    kp.ksplit = p.ksplit, kp.chunks_per_split = p.cps, kp.split_stride = out_floats, kp.out = final_out;
    return launch_splitk_reduce(kp, ws, out_floats / 4, st);
}

}  // namespace pws

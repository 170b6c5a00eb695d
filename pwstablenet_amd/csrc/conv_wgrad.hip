// Weight gradient of the conv / transposed-conv layers on the fp32 matrix cores (gfx950, v_mfma_f32_32x32x2_f32).
//
// Computed directly in the FORWARD kernel's packed layout, where every layer is a correlation
//     y[m][co] = sum_{tap,ci} x[m*S + tap - pad][ci] * P[class][tap][ci][co]           (conv_mfma.hip)
// so   dP[class][tap][ci][co] = sum_m x[m*S + tap - pad][ci] * dy[m][co]   -- a GEMM with K = pixels.
//
// GEMM view per tap:  M = 32 input channels, N = 32 output channels, K = pixels of a spatial tile.
//   * A workgroup owns one (ci block, co block, parity class, tap group) and a strided subset of the spatial tiles
//     (the K dimension is split over workgroups: "pixel split").  Per tile it stages the halo'd x tile [pix][32+1]
//     and the dy tile [pix][32+1] in LDS; both MFMA operands are then conflict-free row reads (lane = channel).
//   * The 4 waves split the tile's pixels; each keeps one 32x32 accumulator per tap (9 taps = 144 VGPRs), the dy
//     fragment is shared by all taps.  After the last tile the 4 partial sums are added through LDS and the result
//     is added to global memory with fp32 atomics (pixel splits, and stages 2/3 sharing weights, accumulate there).
//   * k5 (first layer, 25 taps) runs as 5 tap groups of one kernel row each (blockIdx.z).
// Numerics: fp32, atomics => the sum order over workgroups varies run to run (last-ulp differences in dW).
#include "common.h"

namespace pws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
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
    int tiles_x, tiles_y, tiles_n;
    int ntiles;
    int ci_blocks, co_blocks;
};

template <int KS_, int STRIDE_, int PAD_, int SUBPIX_, int TH_, int TW_, int TN_, int TG_, bool NCHW_ = false>
struct WgCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, PAD = PAD_, SUBPIX = SUBPIX_, TH = TH_, TW = TW_, TN = TN_;
    static constexpr int TG = TG_;              // taps per workgroup (tap group); TAPS % TG == 0
    static constexpr bool NCHW = NCHW_;
    static constexpr int TAPS = KS * KS, NGROUPS = TAPS / TG;
    static constexpr int BM = TH * TW * TN, QP = BM / 4;  // pixels per wave
    static_assert(QP % TW == 0 && QP % 2 == 0, "a wave's share must be whole rows");
    static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    static constexpr int PIX = TN * IH * IW;
    static constexpr int CP = 33;
    static constexpr int LDS_X = PIX * CP, LDS_G = BM * CP;
    static constexpr int LDS_RED = 4 * 32 * CP;  // cross-wave reduction of one tap
    static constexpr int LDS_FLOATS = (LDS_X + LDS_G) > LDS_RED ? (LDS_X + LDS_G) : LDS_RED;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// offset (in pixels of the LDS x tile) of tile-local output pixel p
template <class C>
__host__ __device__ constexpr int xoff(int p) {
    return ((p / (C::TH * C::TW)) * C::IH + ((p % (C::TH * C::TW)) / C::TW) * C::STRIDE) * C::IW + (p % C::TW) * C::STRIDE;
}

template <class C>
__global__ void __launch_bounds__(256, 2) wgrad_mfma_kernel(const WgradParams p) {
    extern __shared__ float lds[];
    float *xs = lds;
    float *gs = lds + C::LDS_X;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;

    const int cb = blockIdx.y;
    const int ci0 = (cb / p.co_blocks) * 32, co0 = (cb % p.co_blocks) * 32;
    const int cls = C::SUBPIX ? (int)(blockIdx.z & 3) : 0;
    const int tg = C::SUBPIX ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    const int py = cls >> 1, px = cls & 1;
    const int pad_y = C::SUBPIX ? 1 - py : C::PAD, pad_x = C::SUBPIX ? 1 - px : C::PAD;

    f32x16 acc[C::TG];
#pragma unroll
    for (int t = 0; t < C::TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int a_base = (xoff<C>(wv * C::QP) + hi * C::STRIDE) * C::CP + l31;
    const int b_base = (wv * C::QP + hi) * C::CP + l31;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int tx_i = tile % p.tiles_x, ty_i = (tile / p.tiles_x) % p.tiles_y, tn_i = tile / (p.tiles_x * p.tiles_y);
        const int n0 = tn_i * C::TN, y0 = ty_i * C::TH, x0 = tx_i * C::TW;
        const int iy0 = y0 * C::STRIDE - pad_y, ix0 = x0 * C::STRIDE - pad_x;
        __syncthreads();  // previous tile fully consumed
        // ---- x halo tile: channels ci0..ci0+31 of (virtually concatenated) sources.
        // All loads of a batch are issued unconditionally (masked items read a valid dummy address) before the first
        // LDS store, so the batch costs one memory latency instead of one per item.
        if constexpr (C::NCHW) {
            constexpr int ITS = (C::PIX * 32 + 255) / 256;
            constexpr int BATCH = 16;
            for (int it0 = 0; it0 < ITS; it0 += BATCH) {
                float r[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int item = tid + (it0 + k) * 256;
                    const int c = item / C::PIX, pix = item % C::PIX;
                    const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                    const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx, ch = ci0 + c;
                    const bool ok = item < C::PIX * 32 && ch < p.cin && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    const size_t off = ok ? ((size_t)(n * p.cin + ch) * p.H + iy) * p.W + ix : 0;
                    const float v = p.src_ptr[0][off];
                    r[k] = ok ? v : 0.f;
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int item = tid + (it0 + k) * 256;
                    if (item < C::PIX * 32) xs[(item % C::PIX) * C::CP + item / C::PIX] = r[k];
                }
            }
        } else {
            constexpr int ITS = (C::PIX * 8 + 255) / 256;
            float4 r[ITS];
#pragma unroll
            for (int k = 0; k < ITS; ++k) {
                const int item = tid + k * 256;
                const int pix = item >> 3, c4 = (item & 7) * 4;
                const int lx = pix % C::IW, ly = (pix / C::IW) % C::IH, tn = pix / (C::IW * C::IH);
                const int n = n0 + tn, iy = iy0 + ly, ix = ix0 + lx;
                int ch = ci0 + c4;
                const bool ok = item < C::PIX * 8 && ch < p.cin && n < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                int s = 0;
                while (s < p.nsrc - 1 && ch >= p.src_c[s]) ch -= p.src_c[s], ++s;
                const size_t off = ok ? ((size_t)(n * p.H + iy) * p.W + ix) * p.src_ld[s] + ch : 0;
                const float4 v = *reinterpret_cast<const float4 *>(p.src_ptr[ok ? s : 0] + off);
                r[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < ITS; ++k) {
                const int item = tid + k * 256;
                if (item < C::PIX * 8) {
                    float *d = xs + (item >> 3) * C::CP + (item & 7) * 4;
                    d[0] = r[k].x, d[1] = r[k].y, d[2] = r[k].z, d[3] = r[k].w;
                }
            }
        }
        // ---- dy tile: channels co0..co0+31 at the tile's output pixels
        {
            constexpr int ITG = (C::BM * 8 + 255) / 256;
            float4 r[ITG];
#pragma unroll
            for (int k = 0; k < ITG; ++k) {
                const int item = tid + k * 256;
                const int m = item >> 3, c4 = (item & 7) * 4;
                const int tx = m % C::TW, ty = (m / C::TW) % C::TH, tn = m / (C::TW * C::TH);
                const int n = n0 + tn, y = y0 + ty, x = x0 + tx;
                const int oy = C::SUBPIX ? 2 * y + py : y, ox = C::SUBPIX ? 2 * x + px : x;
                const bool ok = item < C::BM * 8 && co0 + c4 < p.cout && n < p.N && y < p.LH && x < p.LW && oy < p.OH && ox < p.OW;
                const size_t off = ok ? ((size_t)(n * p.OH + oy) * p.OW + ox) * p.gout_ld + co0 + c4 : 0;
                const float4 v = *reinterpret_cast<const float4 *>(p.gout + off);
                r[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < ITG; ++k) {
                const int item = tid + k * 256;
                if (item < C::BM * 8) {
                    float *d = gs + (item >> 3) * C::CP + (item & 7) * 4;
                    d[0] = r[k].x, d[1] = r[k].y, d[2] = r[k].z, d[3] = r[k].w;
                }
            }
        }
        __syncthreads();
        // ---- this wave's QP pixels: K steps of 2 pixels
#pragma unroll
        for (int j = 0; j < C::QP / 2; ++j) {
            const float b = gs[b_base + 2 * j * C::CP];
#pragma unroll
            for (int t = 0; t < C::TG; ++t) {
                // tap inside the kernel window; tg is block-uniform and only the k5 kernel has NGROUPS > 1
                const int tap = (C::NGROUPS == 1 ? 0 : tg * C::TG) + t;
                const int toff = ((tap / C::KS) * C::IW + (tap % C::KS)) * C::CP;
                const float a = xs[a_base + xoff<C>(2 * j) * C::CP + toff];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }

    // ---- cross-wave reduction through LDS, then one atomic per element
    float *red = lds;
#pragma unroll
    for (int t = 0; t < C::TG; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;  // row = input channel
            red[(wv * 32 + i) * C::CP + l31] = acc[t][r];
        }
        __syncthreads();
        const int tap = tg * C::TG + t;
        for (int e = tid; e < 32 * 32; e += 256) {
            const int i = e >> 5, jn = e & 31;
            const float v = red[i * C::CP + jn] + red[(32 + i) * C::CP + jn] + red[(64 + i) * C::CP + jn] +
                            red[(96 + i) * C::CP + jn];
            if (ci0 + i < p.cin_pad && co0 + jn < p.cout)
                atomicAdd(p.dw + ((size_t)(cls * C::TAPS + tap) * p.cin_pad + ci0 + i) * p.cout + co0 + jn, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct WgChoice {
    int th, tw, tn;
    int (*launch)(WgradParams &, int nclasses, hipStream_t);
};

template <class C>
static int launch_wg(WgradParams &p, int nclasses, hipStream_t st) {
    static PerDeviceFlag attr_set_dev;
    bool &attr_set = attr_set_dev.cur();   // hipFuncSetAttribute acts on the CURRENT device's function object
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mfma_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(wgrad_mfma_kernel, %d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return PWS_EHIP;
        }
        attr_set = true;
    }
    p.tiles_x = (p.LW + C::TW - 1) / C::TW, p.tiles_y = (p.LH + C::TH - 1) / C::TH, p.tiles_n = (p.N + C::TN - 1) / C::TN;
    p.ntiles = p.tiles_x * p.tiles_y * p.tiles_n;
    p.ci_blocks = (p.cin_pad + 31) / 32, p.co_blocks = (p.cout + 31) / 32;
    const long other = (long)p.ci_blocks * p.co_blocks * nclasses * C::NGROUPS;
    long ps = (1024 + other - 1) / other;  // ~4 workgroups per CU overall
    if (ps > p.ntiles) ps = p.ntiles;
    if (ps < 1 || t_deterministic) ps = 1;
    dim3 grid((unsigned)ps, (unsigned)(p.ci_blocks * p.co_blocks), (unsigned)(nclasses * C::NGROUPS));
    hipLaunchKernelGGL(wgrad_mfma_kernel<C>, grid, dim3(256), C::LDS_BYTES, st, p);
    return check_launch("wgrad_mfma_kernel");
}

template <class C>
static constexpr WgChoice wchoice() {
    return WgChoice{C::TH, C::TW, C::TN, &launch_wg<C>};
}

//                      KS S  P  subpix TH  TW  TN  TG
using WG_K3S1_T256 = WgCfg<3, 1, 1, 0, 16, 16, 1, 9>;
using WG_K3S1_T64 = WgCfg<3, 1, 1, 0, 8, 8, 1, 9>;
using WG_K3S1_T64N4 = WgCfg<3, 1, 1, 0, 4, 4, 4, 9>;
using WG_K3S1_T64N16 = WgCfg<3, 1, 1, 0, 2, 2, 16, 9>;
using WG_K3S2_T64 = WgCfg<3, 2, 1, 0, 8, 8, 1, 9>;
using WG_K3S2_T64N4 = WgCfg<3, 2, 1, 0, 4, 4, 4, 9>;
using WG_K3S2_T64N16 = WgCfg<3, 2, 1, 0, 2, 2, 16, 9>;
using WG_K5S1_T128 = WgCfg<5, 1, 2, 0, 8, 16, 1, 5>;
using WG_K5S1_T128_NCHW = WgCfg<5, 1, 2, 0, 8, 16, 1, 5, true>;
using WG_CT4_T256 = WgCfg<2, 1, 0, 1, 16, 16, 1, 4>;
using WG_CT4_T64 = WgCfg<2, 1, 0, 1, 8, 8, 1, 4>;
using WG_CT4_T64N4 = WgCfg<2, 1, 0, 1, 4, 4, 4, 4>;
using WG_CT4_T64N16 = WgCfg<2, 1, 0, 1, 2, 2, 16, 4>;

static const WgChoice kWgK3S1[] = {wchoice<WG_K3S1_T256>(), wchoice<WG_K3S1_T64>(), wchoice<WG_K3S1_T64N4>(),
                                   wchoice<WG_K3S1_T64N16>()};
static const WgChoice kWgK3S2[] = {wchoice<WG_K3S2_T64>(), wchoice<WG_K3S2_T64N4>(), wchoice<WG_K3S2_T64N16>()};
static const WgChoice kWgCT4[] = {wchoice<WG_CT4_T256>(), wchoice<WG_CT4_T64>(), wchoice<WG_CT4_T64N4>(),
                                  wchoice<WG_CT4_T64N16>()};
static const WgChoice kWgK5[] = {wchoice<WG_K5S1_T128>()};
static const WgChoice kWgK5N[] = {wchoice<WG_K5S1_T128_NCHW>()};

// largest tile that is not mostly padding for this map
static const WgChoice &pick(const WgChoice *c, int n, int LH, int LW, int N) {
    for (int i = 0; i < n; ++i) {
        const long tiles = (long)((LW + c[i].tw - 1) / c[i].tw) * ((LH + c[i].th - 1) / c[i].th) * ((N + c[i].tn - 1) / c[i].tn);
        const double useful = (double)N * LH * LW / ((double)tiles * c[i].th * c[i].tw * c[i].tn);
        if (useful >= 0.45 || i + 1 == n) return c[i];
    }
    return c[n - 1];
}

int wgrad_bf16_launch(const pws_conv_bwd_weight_args *a, hipStream_t st);  // wgrad_bf16.hip

int wgrad_ring_try_pair(const pws_conv_bwd_weight_args *a, hipStream_t st);   // wgrad_ring.hip: both operand pairs in one launch; 1 = not covered
int wgrad_bf16_launch_pair(const pws_conv_bwd_weight_args *a, hipStream_t st);   // wgrad_bf16.hip: the same on wgrad_bf16_kernel; 1 = not covered

int conv2d_bwd_weight_impl(const pws_conv_bwd_weight_args *a, hipStream_t st) {
    PWS_REQUIRE(a != nullptr, "pws_conv2d_bwd_weight: args is NULL");
    PWS_REQUIRE(a->n >= 0 && a->h > 0 && a->w > 0 && a->cout > 0 && a->cout % 4 == 0, "pws_conv2d_bwd_weight: bad shape");
    PWS_REQUIRE(a->nsrc >= 1 && a->nsrc <= 4 && a->gout && a->dw_packed && a->gout_ld >= a->cout && a->gout_ld % 4 == 0 &&
                    (reinterpret_cast<size_t>(a->gout) & 15) == 0,
                "pws_conv2d_bwd_weight: bad sources / gout / dw");
    if (a->n == 0) return PWS_OK;
    if (a->gout2) {
        // two operand pairs (a layer shared by stages 2 and 3): one launch where the ring kernel covers the layer, else one after the other
        for (int s = 0; s < a->nsrc; ++s) PWS_REQUIRE(a->src2_ptr[s] != nullptr, "pws_conv2d_bwd_weight: gout2 without src2_ptr[%d]", s);
        PWS_REQUIRE((reinterpret_cast<size_t>(a->gout2) & 15) == 0, "pws_conv2d_bwd_weight: gout2 alignment");
        if (a->math == PWS_MATH_BF16 && a->store == PWS_STORE_BF16 && !a->src_nchw && !(t_deterministic && a->dbias)) {
            int rc = wgrad_ring_try_pair(a, st);   // 1: not covered
            if (rc != 1) return rc;
            rc = wgrad_bf16_launch_pair(a, st);          // the deep levels (maps below 16 x 16, short tile streams)
            if (rc != 1) return rc;
        }
        pws_conv_bwd_weight_args b = *a;
        b.gout2 = nullptr;
        for (int s = 0; s < 4; ++s) b.src2_ptr[s] = nullptr;
        int rc = conv2d_bwd_weight_impl(&b, st);
        if (rc != PWS_OK) return rc;
        b.gout = a->gout2;
        for (int s = 0; s < a->nsrc; ++s) b.src[s].ptr = static_cast<const float *>(a->src2_ptr[s]);
        return conv2d_bwd_weight_impl(&b, st);
    }
    PWS_REQUIRE(!(t_deterministic && a->dbias),
                "pws_conv2d_bwd_weight: deterministic with dbias: the parity classes / the scratch-less bias pass add in arrival order -- pass "
                "dbias = NULL and sum the bias with pws_act_bwd_bias_s(act = PWS_ACT_NONE) and a workspace (ordered slab sums)");
    if (a->math == PWS_MATH_BF16) {
        const int rc = wgrad_bf16_launch(a, st);  // 1: not covered by the bf16 kernel (first layer, odd channel counts)
        if (rc != 1) return rc;
    }
    PWS_REQUIRE(a->store != PWS_STORE_BF16, "pws_conv2d_bwd_weight: bf16 storage needs bf16 math and a layer the bf16 kernel covers");
    if (a->dbias) {   // the fp32 kernels do not take the bias sum along: one elementwise pass (y is not read for PWS_ACT_NONE)
        const int oh = a->kind == PWS_CONV_K3S2 ? (a->h - 1) / 2 + 1 : (a->kind == PWS_CONVT_K4S2 ? 2 * a->h : a->h);
        const int ow = a->kind == PWS_CONV_K3S2 ? (a->w - 1) / 2 + 1 : (a->kind == PWS_CONVT_K4S2 ? 2 * a->w : a->w);
        PWS_REQUIRE(a->gout_ld == a->cout, "pws_conv2d_bwd_weight: dbias with the fp32 kernels needs a dense gout");
        const int rc = pws_act_bwd_bias_s(const_cast<float *>(a->gout), a->gout, (size_t)a->n * oh * ow, a->cout, PWS_ACT_NONE, a->dbias,
                                          PWS_STORE_FP32, nullptr, 0, st);
        if (rc != PWS_OK) return rc;
    }
    WgradParams p{};
    p.nsrc = a->nsrc;
    int cin = 0;
    const bool nchw = a->src_nchw != 0;
    if (nchw) {
        PWS_REQUIRE(a->kind == PWS_CONV_K5S1 && a->nsrc == 1 && a->src[0].ptr, "pws_conv2d_bwd_weight: NCHW source only for k5");
        p.src_ptr[0] = a->src[0].ptr, p.src_c[0] = a->src[0].channels, cin = a->src[0].channels;
    } else {
        for (int s = 0; s < a->nsrc; ++s) {
            const pws_src &sr = a->src[s];
            PWS_REQUIRE(sr.ptr && sr.channels > 0 && sr.channels % 16 == 0 && sr.ld >= sr.channels && sr.ld % 4 == 0 &&
                            (reinterpret_cast<size_t>(sr.ptr) & 15) == 0,
                        "pws_conv2d_bwd_weight: bad source %d", s);
            p.src_ptr[s] = sr.ptr, p.src_c[s] = sr.channels, p.src_ld[s] = sr.ld;
            cin += sr.channels;
        }
    }
    p.cin = cin, p.cin_pad = (cin + 15) / 16 * 16, p.cout = a->cout;
    p.N = a->n, p.H = a->h, p.W = a->w;
    p.gout = a->gout, p.gout_ld = a->gout_ld, p.dw = a->dw_packed;
    PWS_REQUIRE((size_t)a->n * a->h * a->w < (1u << 30), "pws_conv2d_bwd_weight: too large");
    double k2 = 9;
    int nclasses = 1;
    const WgChoice *c = nullptr;
    switch (a->kind) {
    case PWS_CONV_K3S1:
    case PWS_CONVT_K3S1:
        p.OH = p.LH = a->h, p.OW = p.LW = a->w;
        c = &pick(kWgK3S1, 4, p.LH, p.LW, p.N);
        break;
    case PWS_CONV_K3S2:
        p.OH = p.LH = (a->h - 1) / 2 + 1, p.OW = p.LW = (a->w - 1) / 2 + 1;
        c = &pick(kWgK3S2, 3, p.LH, p.LW, p.N);
        break;
    case PWS_CONV_K5S1:
        p.OH = p.LH = a->h, p.OW = p.LW = a->w, k2 = 25;
        c = nchw ? &kWgK5N[0] : &kWgK5[0];
        break;
    case PWS_CONVT_K4S2:
        p.LH = a->h, p.LW = a->w, p.OH = 2 * a->h, p.OW = 2 * a->w, nclasses = 4, k2 = 4;
        c = &pick(kWgCT4, 4, p.LH, p.LW, p.N);
        break;
    default:
        set_error("pws_conv2d_bwd_weight: kind %d has no weight gradient here", a->kind);
        return PWS_EINVAL;
    }
    const double out_pix = (double)a->n * p.OH * p.OW;
    ProfScope prof(KID_WGRAD, 2.0 * out_pix * a->cout * cin * k2,
                   4.0 * ((double)a->n * a->h * a->w * cin + out_pix * a->cout + k2 * cin * a->cout * (nclasses == 4 ? 4 : 1)), st);
    return c->launch(p, nclasses, st);
}

// ------------------------------------------------------------------------------------------------ act' and bias grad
// dy <- dy * act'(y) in place; dbias[c] += sum over pixels.  HBM-bound (12 B per element).
// Lane t owns channel quad (t % c4n) of every (256 / c4n)-th pixel of the workgroup's (grid-strided) pixels, so the
// per-channel sums stay in registers; one LDS pass folds the pixel groups, then one global atomic per channel and
// workgroup (at most 512 workgroups: all of them hit the same c words).
constexpr int ABB_PIX = 256;
constexpr int ABB_MAX_BLOCKS = 1024;  // slab reduction (two launches); measured per bf16 training step: 256 -> 3.26 ms, 1024 -> 2.87 ms, 2048 -> 3.21 ms (batch 32)

// One lane owns VEC = 4 (fp32) or 8 (bf16 storage) consecutive channels = one 16-byte load per tensor and pixel.
// Bias-gradient reduction across workgroups: with a scratch buffer `ws` the workgroups store their partial sums as slabs
// [workgroup][c] and a second tiny launch adds the slabs into dbias.  Without ws: one fp32 atomic per channel and workgroup
// -- every workgroup hits the same c words, and cross-XCD atomics on one line serialise (measured per training step, batch 8:
// 1024 workgroups 3.4 ms, 128 workgroups 2.2 ms).  (An in-kernel ticket + last-arriver reduction was tried: its release fence
// has to write back the L2 lines the dy stores just dirtied -- 4.2 ms.)
// NOACT: act == PWS_ACT_NONE known at compile time (bias sum of an already pre-activation gradient: y is not read at all)
template <bool IO16, bool NOACT>
__global__ void __launch_bounds__(256) act_bwd_bias_kernel(float *__restrict__ dy, const float *__restrict__ y, size_t pixels,
                                                           int c, int act, float *__restrict__ dbias, float *__restrict__ ws) {
    constexpr int VEC = IO16 ? 8 : 4, NV = VEC / 4;
    float *slab = ws ? ws + (size_t)blockIdx.x * c : nullptr;
    extern __shared__ float sred[];  // 256 x VEC floats
    const int tid = threadIdx.x;
    const int cvn = c / VEC;
    const int span = cvn < 256 ? cvn : 256;  // channel groups handled per pass
    const int groups = 256 / span;           // pixel groups (lanes beyond groups*span idle when span does not divide 256)
    // grid-stride over pixel chunks: few workgroups => few (contended) global atomics on the c bias-gradient words
    for (int q0 = 0; q0 < cvn; q0 += 256) {  // one pass unless c > 256 * VEC
        const int q = q0 + tid % span;
        const int grp = tid / span;
        float s[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] = 0.f;
        if (q < cvn && grp < groups) {
            // 4 independent pixel strides in flight per lane (a single dependent load pair per lane is latency-bound)
            const size_t step = (size_t)gridDim.x * groups;
            for (size_t pb = (size_t)blockIdx.x * groups + grp; pb < pixels; pb += 4 * step) {
                float4 g[4][NV], v[4][NV];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t p = pb + u * step;
                    ok[u] = p < pixels;
                    const size_t e = ((ok[u] ? p : pb) * cvn + q) * VEC;
#pragma unroll
                    for (int h = 0; h < NV; ++h) {
                        g[u][h] = ld4<IO16>(dy, e + 4 * h);
                        if constexpr (!NOACT) v[u][h] = ld4<IO16>(y, e + 4 * h);
                        else v[u][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    const size_t e = ((pb + u * step) * cvn + q) * VEC;
#pragma unroll
                    for (int h = 0; h < NV; ++h) {
                        float4 gg = g[u][h];
                        const float4 vv = v[u][h];
                        if (act == PWS_ACT_LRELU) {
                            gg.x *= vv.x > 0.f ? 1.f : 0.2f, gg.y *= vv.y > 0.f ? 1.f : 0.2f, gg.z *= vv.z > 0.f ? 1.f : 0.2f,
                                gg.w *= vv.w > 0.f ? 1.f : 0.2f;
                        } else if (act == PWS_ACT_RELU) {
                            gg.x = vv.x > 0.f ? gg.x : 0.f, gg.y = vv.y > 0.f ? gg.y : 0.f, gg.z = vv.z > 0.f ? gg.z : 0.f,
                            gg.w = vv.w > 0.f ? gg.w : 0.f;
                        }
                        if (act != PWS_ACT_NONE) st4<IO16>(dy, e + 4 * h, gg);
                        s[4 * h] += gg.x, s[4 * h + 1] += gg.y, s[4 * h + 2] += gg.z, s[4 * h + 3] += gg.w;
                    }
                }
            }
        }
        if (dbias) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VEC; ++k) sred[tid * VEC + k] = s[k];
            __syncthreads();
            if (tid < span && q0 + tid < cvn) {
                float t[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) t[k] = 0.f;
                for (int g_ = 0; g_ < groups; ++g_)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) t[k] += sred[(g_ * span + tid) * VEC + k];
                if (slab) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) slab[(size_t)(q0 + tid) * VEC + k] = t[k];
                } else {
                    float *d = dbias + (size_t)(q0 + tid) * VEC;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) atomicAdd(d + k, t[k]);
                }
            }
        }
    }
}

// second launch of the slab path: dbias[ch] += sum over the workgroups' slabs (the kernel boundary orders the slab stores;
// a device-scope release fence inside the first kernel would have to write back the L2 lines dirtied by the dy stores)
__global__ void __launch_bounds__(256) bias_slab_reduce_kernel(const float *__restrict__ slabs, int nslabs, int c,
                                                               float *__restrict__ dbias) {
    // workgroup = 16 channels x 16 slab groups: every lane sums nslabs / 16 slabs (one serial chain per channel took 8 us)
    __shared__ float part[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (ch < c)
        for (int b = g; b < nslabs; b += 16) s += slabs[(size_t)b * c + ch];
    part[g][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && blockIdx.x * 16 + threadIdx.x < c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
        dbias[blockIdx.x * 16 + threadIdx.x] += t;  // stream-ordered read-modify-write: the only writer of dbias in this launch
    }
}

}  // namespace pws

extern "C" int pws_conv2d_bwd_weight(const pws_conv_bwd_weight_args *args, pws_stream_t stream) {
    pws::DeterministicScope det(args && args->deterministic != 0);
    return pws::conv2d_bwd_weight_impl(args, pws::as_stream(stream));
}

extern "C" int pws_act_bwd_bias(float *dy, const float *y, size_t pixels, int c, int act, float *dbias, pws_stream_t stream) {
    return pws_act_bwd_bias_s(dy, y, pixels, c, act, dbias, PWS_STORE_FP32, nullptr, 0, stream);
}

extern "C" size_t pws_act_bwd_bias_ws_bytes(int c) { return c > 0 ? sizeof(float) * (size_t)pws::ABB_MAX_BLOCKS * c : 0; }

extern "C" int pws_act_bwd_bias_s(float *dy, const float *y, size_t pixels, int c, int act, float *dbias, int store, void *ws,
                                  size_t ws_bytes, pws_stream_t stream) {
    PWS_REQUIRE(c > 0 && c % (store == PWS_STORE_BF16 ? 8 : 4) == 0, "pws_act_bwd_bias: c=%d must be a positive multiple of 4 (8 for bf16 storage)", c);
    PWS_REQUIRE(act >= PWS_ACT_NONE && act <= PWS_ACT_RELU, "pws_act_bwd_bias: bad act %d", act);
    if (pixels == 0) return PWS_OK;
    PWS_REQUIRE(dy && y && ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(y)) & 15) == 0,
                "pws_act_bwd_bias: NULL or unaligned pointer");
    if (act == PWS_ACT_NONE && !dbias) return PWS_OK;
    const bool io16 = store == PWS_STORE_BF16;
    pws::ProfScope prof(pws::KID_ACT_BWD, 2.0 * pixels * c, (act == PWS_ACT_NONE ? 1.0 : 3.0) * (io16 ? 2.0 : 4.0) * pixels * c, pws::as_stream(stream));
    size_t blocks = (pixels + pws::ABB_PIX - 1) / pws::ABB_PIX;
    float *slabs = nullptr;
    if (dbias && ws && ws_bytes >= pws_act_bwd_bias_ws_bytes(c) && (reinterpret_cast<size_t>(ws) & 15) == 0) {
        slabs = static_cast<float *>(ws);
        if (blocks > (size_t)pws::ABB_MAX_BLOCKS) blocks = pws::ABB_MAX_BLOCKS;
    } else if (!dbias) {
        if (blocks > 4096) blocks = 4096;   // no bias gradient (frozen layers): a plain elementwise pass, fill the chip
    } else if (blocks > 128) {
        blocks = 128;  // atomic tail: fewer, longer workgroups win (see the kernel comment)
    }
    const bool noact = act == PWS_ACT_NONE;
    if (io16 && noact)
        hipLaunchKernelGGL((pws::act_bwd_bias_kernel<true, true>), dim3((unsigned)blocks), dim3(256), sizeof(float) * 256 * 8,
                           pws::as_stream(stream), dy, y, pixels, c, act, dbias, slabs);
    else if (io16)
        hipLaunchKernelGGL((pws::act_bwd_bias_kernel<true, false>), dim3((unsigned)blocks), dim3(256), sizeof(float) * 256 * 8,
                           pws::as_stream(stream), dy, y, pixels, c, act, dbias, slabs);
    else if (noact)
        hipLaunchKernelGGL((pws::act_bwd_bias_kernel<false, true>), dim3((unsigned)blocks), dim3(256), sizeof(float) * 256 * 4,
                           pws::as_stream(stream), dy, y, pixels, c, act, dbias, slabs);
    else
        hipLaunchKernelGGL((pws::act_bwd_bias_kernel<false, false>), dim3((unsigned)blocks), dim3(256), sizeof(float) * 256 * 4,
                           pws::as_stream(stream), dy, y, pixels, c, act, dbias, slabs);
    if (slabs)
        hipLaunchKernelGGL(pws::bias_slab_reduce_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, pws::as_stream(stream), slabs,
                           (int)blocks, c, dbias);
    return pws::check_launch("act_bwd_bias_kernel");
}

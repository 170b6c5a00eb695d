/*
 * pws_oracle.c -- CPU restatement of PWStableNet's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in pwstablenet_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * What it restates (reference = mindazhao/PWStableNet, paths relative to the reference checkout):
 *   - lib/networks_cascading.py:108-237  UnetGenerator (3-stage cascade, weight sharing s2/s3)
 *   - lib/networks_cascading.py:245-350  blocks down / down_bottom / up / up_bottom
 *   - torch ops the reference dispatches (the arithmetic lives in PyTorch, not in the reference tree;
 *     the reference pins "pytorch 0.4.0+", README.md:27; the oracle follows the semantics of the torch
 *     2.10 build the goldens were generated with): conv2d, conv_transpose2d, leaky_relu(0.2), relu, tanh,
 *     affine_grid / grid_sample (bilinear, zeros padding, align_corners False by default),
 *     UpsamplingBilinear2d (align_corners=True), Adam.
 *
 * Parity pinning: the reference ships no tests or golden vectors, so this
 * restatement is pinned against vectors produced by importing the reference's own Python in the build
 * container (tests/golden/make_golden.py -> tests/golden/ *.npz), see tests/test_oracle_golden.py.
 *
 * All tensors are dense fp32.  Activations are NCHW (as in the reference); warp fields are N,H,W,2.
 * Parallelised with OpenMP over independent outputs only (no reduction is split across threads), so
 * results do not depend on the thread count.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ACT_NONE 0
#define ORC_ACT_LRELU 1 /* nn.LeakyReLU(0.2, True)  lib/networks_cascading.py:250,271 */
#define ORC_ACT_RELU 2  /* nn.ReLU(True)            lib/networks_cascading.py:305,328 */
#define ORC_ACT_TANH 3  /* nn.Tanh()                lib/networks_cascading.py:254,256 */

#ifdef __cplusplus
extern "C" {
#endif

int orc_version(void) { return 1; }

static inline float orc_act(float v, int act) {
    switch (act) {
    case ORC_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case ORC_ACT_RELU: return v > 0.f ? v : 0.f;
    case ORC_ACT_TANH: return tanhf(v);
    default: return v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * conv2d: out[n,co,oy,ox] = act(b[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * in[n,ci,oy*s+ky-p,ox*s+kx-p])
 * (nn.Conv2d call sites lib/networks_cascading.py:248,269,274,285; zero padding.)
 * ---------------------------------------------------------------------------------------------- */
void orc_conv2d(const float *in, const float *w, const float *b, float *out, int N, int Cin, int H,
                int W, int Cout, int k, int s, int p, int act) {
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *o = out + ((size_t)n * Cout + co) * Ho * Wo;
            const float bias = b ? b[co] : 0.f;
            for (int i = 0; i < Ho * Wo; ++i) o[i] = bias;
            for (int ci = 0; ci < Cin; ++ci) {
                const float *ip = in + ((size_t)n * Cin + ci) * H * W;
                const float *wp = w + ((size_t)co * Cin + ci) * k * k;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wp[ky * k + kx];
                        /* valid output range for this tap */
                        int oy0 = 0, oy1 = Ho, ox0 = 0, ox1 = Wo;
                        while (oy0 < Ho && oy0 * s + ky - p < 0) ++oy0;
                        while (oy1 > oy0 && (oy1 - 1) * s + ky - p >= H) --oy1;
                        while (ox0 < Wo && ox0 * s + kx - p < 0) ++ox0;
                        while (ox1 > ox0 && (ox1 - 1) * s + kx - p >= W) --ox1;
                        for (int oy = oy0; oy < oy1; ++oy) {
                            const float *irow = ip + (size_t)(oy * s + ky - p) * W + (kx - p);
                            float *orow = o + (size_t)oy * Wo;
                            if (s == 1) {
                                for (int ox = ox0; ox < ox1; ++ox) orow[ox] += wv * irow[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox) orow[ox] += wv * irow[ox * s];
                            }
                        }
                    }
            }
            if (act != ORC_ACT_NONE)
                for (int i = 0; i < Ho * Wo; ++i) o[i] = orc_act(o[i], act);
        }
}

/* ------------------------------------------------------------------------------------------------
 * conv_transpose2d (weight layout Cin,Cout,k,k as torch):
 *   out[n,co,iy*s-p+ky, ix*s-p+kx] += in[n,ci,iy,ix] * w[ci,co,ky,kx];  Ho=(H-1)*s-2p+k.
 * (nn.ConvTranspose2d call sites lib/networks_cascading.py:306,330,339.)
 * Written as a gather over outputs so that every output element is owned by one thread.
 * ---------------------------------------------------------------------------------------------- */
void orc_conv_transpose2d(const float *in, const float *w, const float *b, float *out, int N, int Cin,
                          int H, int W, int Cout, int k, int s, int p, int act) {
    const int Ho = (H - 1) * s - 2 * p + k, Wo = (W - 1) * s - 2 * p + k;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *o = out + ((size_t)n * Cout + co) * Ho * Wo;
            const float bias = b ? b[co] : 0.f;
            for (int i = 0; i < Ho * Wo; ++i) o[i] = bias;
            for (int ci = 0; ci < Cin; ++ci) {
                const float *ip = in + ((size_t)n * Cin + ci) * H * W;
                const float *wp = w + ((size_t)ci * Cout + co) * k * k;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wp[ky * k + kx];
                        for (int iy = 0; iy < H; ++iy) {
                            const int oy = iy * s - p + ky;
                            if (oy < 0 || oy >= Ho) continue;
                            float *orow = o + (size_t)oy * Wo;
                            const float *irow = ip + (size_t)iy * W;
                            for (int ix = 0; ix < W; ++ix) {
                                const int ox = ix * s - p + kx;
                                if (ox < 0 || ox >= Wo) continue;
                                orow[ox] += wv * irow[ix];
                            }
                        }
                    }
            }
            if (act != ORC_ACT_NONE)
                for (int i = 0; i < Ho * Wo; ++i) o[i] = orc_act(o[i], act);
        }
}

/* ------------------------------------------------------------------------------------------------
 * affine_grid(theta[N,2,3], size=(N,C,H,W)) -> grid[N,H,W,2]  (lib/networks_cascading.py:164,188,210)
 * base coords: align_corners=False: x_j=(2j+1)/W-1 ; True: x_j = 2j/(W-1)-1 (W>1).
 * grid[n,h,w,0] = t00*x + t01*y + t02 ; grid[n,h,w,1] = t10*x + t11*y + t12.
 * ---------------------------------------------------------------------------------------------- */
static inline float orc_base_coord(int j, int size, int align_corners) {
    if (align_corners) return size > 1 ? (2.f * j) / (float)(size - 1) - 1.f : 0.f;
    return (2.f * j + 1.f) / (float)size - 1.f;
}

void orc_affine_grid(const float *theta, float *grid, int N, int H, int W, int align_corners) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) {
            const float *t = theta + (size_t)n * 6;
            const float y = orc_base_coord(h, H, align_corners);
            for (int x_ = 0; x_ < W; ++x_) {
                const float x = orc_base_coord(x_, W, align_corners);
                float *g = grid + (((size_t)n * H + h) * W + x_) * 2;
                g[0] = t[0] * x + t[1] * y + t[2];
                g[1] = t[3] * x + t[4] * y + t[5];
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * grid_sample, bilinear, padding_mode='zeros'  (F.grid_sample call sites main_new.py:106,109,116,118,
 * 197,716).  Unnormalise: align_corners=False: ix=((gx+1)*W-1)/2 ; True: ix=(gx+1)/2*(W-1).
 * Taps outside the image contribute 0.
 * ---------------------------------------------------------------------------------------------- */
static inline float orc_unnorm(float g, int size, int align_corners) {
    /* same operation order as ATen's CPU kernel (the build that produced the goldens):
     * (g+1)*scaling - 0.5 with scaling = size/2 ; ATen's GPU kernel evaluates ((g+1)*size-1)/2, which
     * differs by an ulp of the coordinate (~1.5e-5 px at size 256) */
    if (align_corners) return (g + 1.f) * ((float)(size - 1) * 0.5f);
    return (g + 1.f) * ((float)size * 0.5f) - 0.5f;
}

void orc_grid_sample_fwd(const float *input, const float *grid, float *out, int N, int C, int H, int W,
                         int Ho, int Wo, int align_corners) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < Ho; ++h)
            for (int x_ = 0; x_ < Wo; ++x_) {
                const float *g = grid + (((size_t)n * Ho + h) * Wo + x_) * 2;
                const float ix = orc_unnorm(g[0], W, align_corners);
                const float iy = orc_unnorm(g[1], H, align_corners);
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
                const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
                const int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
                const int vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
                for (int c = 0; c < C; ++c) {
                    const float *ip = input + ((size_t)n * C + c) * H * W;
                    float acc = 0.f;
                    if (vy0 && vx0) acc += ip[(size_t)y0 * W + x0] * nw;
                    if (vy0 && vx1) acc += ip[(size_t)y0 * W + x1] * ne;
                    if (vy1 && vx0) acc += ip[(size_t)y1 * W + x0] * sw;
                    if (vy1 && vx1) acc += ip[(size_t)y1 * W + x1] * se;
                    out[(((size_t)n * C + c) * Ho + h) * Wo + x_] = acc;
                }
            }
}

/* Backward of the above (autograd through loss_g.backward(), main_new.py:214).  ginput and/or ggrid
 * may be NULL.  ginput is accumulated per image serially (scatter-add), so it is deterministic. */
void orc_grid_sample_bwd(const float *gout, const float *input, const float *grid, float *ginput,
                         float *ggrid, int N, int C, int H, int W, int Ho, int Wo, int align_corners) {
    const float sx = align_corners ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
    const float sy = align_corners ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
    if (ginput) memset(ginput, 0, sizeof(float) * (size_t)N * C * H * W);
#pragma omp parallel for
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < Ho; ++h)
            for (int x_ = 0; x_ < Wo; ++x_) {
                const size_t gi = (((size_t)n * Ho + h) * Wo + x_) * 2;
                const float ix = orc_unnorm(grid[gi], W, align_corners);
                const float iy = orc_unnorm(grid[gi + 1], H, align_corners);
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
                const int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
                const int vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
                float gix = 0.f, giy = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float go = gout[(((size_t)n * C + c) * Ho + h) * Wo + x_];
                    const float *ip = input + ((size_t)n * C + c) * H * W;
                    const float v_nw = (vy0 && vx0) ? ip[(size_t)y0 * W + x0] : 0.f;
                    const float v_ne = (vy0 && vx1) ? ip[(size_t)y0 * W + x1] : 0.f;
                    const float v_sw = (vy1 && vx0) ? ip[(size_t)y1 * W + x0] : 0.f;
                    const float v_se = (vy1 && vx1) ? ip[(size_t)y1 * W + x1] : 0.f;
                    gix += go * ((v_ne - v_nw) * wy0 + (v_se - v_sw) * wy1);
                    giy += go * ((v_sw - v_nw) * wx0 + (v_se - v_ne) * wx1);
                    if (ginput) {
                        float *gp = ginput + ((size_t)n * C + c) * H * W;
                        if (vy0 && vx0) gp[(size_t)y0 * W + x0] += go * wx0 * wy0;
                        if (vy0 && vx1) gp[(size_t)y0 * W + x1] += go * wx1 * wy0;
                        if (vy1 && vx0) gp[(size_t)y1 * W + x0] += go * wx0 * wy1;
                        if (vy1 && vx1) gp[(size_t)y1 * W + x1] += go * wx1 * wy1;
                    }
                }
                if (ggrid) {
                    ggrid[gi] = gix * sx;
                    ggrid[gi + 1] = giy * sy;
                }
            }
}

/* ------------------------------------------------------------------------------------------------
 * UpsamplingBilinear2d(size=(Ho,Wo)) == bilinear, align_corners=True  (main_new.py:708-709):
 *   src = dst * (in-1)/(out-1)
 * ---------------------------------------------------------------------------------------------- */
void orc_upsample_bilinear_ac(const float *in, float *out, int N, int C, int H, int W, int Ho, int Wo) {
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
#pragma omp parallel for collapse(2)
    for (int nc = 0; nc < N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy) {
            const float *ip = in + (size_t)nc * H * W;
            const float sy = ry * oy;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = sy - y0, hy = 1.f - ly;
            for (int ox = 0; ox < Wo; ++ox) {
                const float sx = rx * ox;
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = sx - x0, hx = 1.f - lx;
                out[((size_t)nc * Ho + oy) * Wo + ox] =
                    hy * (hx * ip[(size_t)y0 * W + x0] + lx * ip[(size_t)y0 * W + x1]) +
                    ly * (hx * ip[(size_t)y1 * W + x0] + lx * ip[(size_t)y1 * W + x1]);
            }
        }
}

/* Adam, torch.optim.Adam defaults as used at main_new.py:63 (no weight decay, no amsgrad):
 *   m=b1*m+(1-b1)*g ; v=b2*v+(1-b2)*g*g ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t)+eps) */
void orc_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2,
                   float eps, int step) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        m[i] = b1 * m[i] + (1.f - b1) * g[i];
        v[i] = b2 * v[i] + (1.f - b2) * g[i] * g[i];
        const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] -= step_size * (m[i] / denom);
    }
}

/* ================================================================================================
 * Whole-network forward.  A tiny NCHW tensor type plus helpers that mirror the reference blocks.
 * ============================================================================================== */
typedef struct {
    float *d;
    int c, h, w;
} T;

static int g_N; /* batch size of the current orc_netg_forward call */

static T t_new(int c, int h, int w) {
    T t;
    t.c = c, t.h = h, t.w = w;
    t.d = (float *)malloc(sizeof(float) * (size_t)g_N * c * h * w);
    return t;
}
static void t_free(T *t) {
    free(t->d);
    t->d = NULL;
}
/* torch.cat([a,b], dim=1)  lib/networks_cascading.py:296,321,346,348,350 */
static T t_cat(T a, T b) {
    T o = t_new(a.c + b.c, a.h, a.w);
    const size_t hw = (size_t)a.h * a.w;
    for (int n = 0; n < g_N; ++n) {
        memcpy(o.d + (size_t)n * o.c * hw, a.d + (size_t)n * a.c * hw, sizeof(float) * a.c * hw);
        memcpy(o.d + ((size_t)n * o.c + a.c) * hw, b.d + (size_t)n * b.c * hw, sizeof(float) * b.c * hw);
    }
    return o;
}
static T t_conv(T x, const float *w, const float *b, int cout, int k, int s, int p, int act) {
    T o = t_new(cout, (x.h + 2 * p - k) / s + 1, (x.w + 2 * p - k) / s + 1);
    orc_conv2d(x.d, w, b, o.d, g_N, x.c, x.h, x.w, cout, k, s, p, act);
    return o;
}
static T t_convT(T x, const float *w, const float *b, int cout, int k, int s, int p, int act) {
    T o = t_new(cout, (x.h - 1) * s - 2 * p + k, (x.w - 1) * s - 2 * p + k);
    orc_conv_transpose2d(x.d, w, b, o.d, g_N, x.c, x.h, x.w, cout, k, s, p, act);
    return o;
}

/* Parameter indices in state-dict order (registration order of lib/networks_cascading.py:112-149;
 * down_bottom registers conv_same before mpconv (:274-287), up_bottom registers mpconv before
 * conv_same (:334-341)).  Each module contributes weight then bias. */
enum {
    P_TRANSFER = 0, /* transfer.mpconv.0 */
    P_DOWN1 = 2,    /* down1..down7: 2,4,...,14 */
    P_UP7 = 16,     /* up7,up6,...,up1: 16,18,...,28 */
    P_OUT = 30,
    P_DB1 = 32,  /* down_bottom k (1..7): conv_same at 32+4(k-1), mpconv at 34+4(k-1) */
    P_UB7 = 60,  /* up_bottom7..1 (j=0..6): mpconv at 60+4j, conv_same at 62+4j */
    P_FLATTEN = 88,
    P_LINEAR = 90,
    P_COUNT = 92
};
#define PW(i) (params[(i)])
#define PB(i) (params[(i) + 1])

/* down.forward  lib/networks_cascading.py:262-264 (conv + LeakyReLU, k3 s2 p1 by default) */
static T blk_down(const float *const *params, int pi, T x, int cout) {
    return t_conv(x, PW(pi), PB(pi), cout, 3, 2, 1, ORC_ACT_LRELU);
}
/* down_bottom.forward  lib/networks_cascading.py:291-298 */
static T blk_down_bottom(const float *const *params, int k, const T *x_left, T x_up, int cout) {
    const int cs = P_DB1 + 4 * (k - 1), mp = cs + 2;
    T c = t_conv(x_up, PW(cs), PB(cs), x_up.c, 3, 1, 1, ORC_ACT_LRELU);
    T o;
    if (x_left) {
        T cc = t_cat(*x_left, c);
        o = t_conv(cc, PW(mp), PB(mp), cout, 3, 2, 1, ORC_ACT_LRELU);
        t_free(&cc);
    } else {
        o = t_conv(c, PW(mp), PB(mp), cout, 3, 2, 1, ORC_ACT_LRELU);
    }
    t_free(&c);
    return o;
}
/* up.forward  lib/networks_cascading.py:316-321 ; level = 7..1 */
static T blk_up(const float *const *params, int level, T x1, const T *x2, int cout) {
    const int pi = P_UP7 + 2 * (7 - level);
    T u = t_convT(x1, PW(pi), PB(pi), cout, 4, 2, 1, ORC_ACT_RELU);
    if (!x2) return u;
    T o = t_cat(u, *x2);
    t_free(&u);
    return o;
}
/* up_bottom.forward  lib/networks_cascading.py:344-350 ; level = 7..1 */
static T blk_up_bottom(const float *const *params, int level, T x_up, T x_left, const T *x_before,
                       int cout) {
    const int mp = P_UB7 + 4 * (7 - level), cs = mp + 2;
    T e = t_convT(x_up, PW(cs), PB(cs), x_up.c, 3, 1, 1, ORC_ACT_RELU);
    T nima = t_cat(e, x_left);
    t_free(&e);
    T v = t_convT(nima, PW(mp), PB(mp), cout, 4, 2, 1, ORC_ACT_RELU);
    t_free(&nima);
    if (!x_before) return v;
    T o = t_cat(v, *x_before);
    t_free(&v);
    return o;
}
/* theta = linear(flatten(x)).view(-1,2,3)  lib/networks_cascading.py:148-149,162-163: both are `down`
 * blocks, i.e. conv + LeakyReLU(0.2) -- theta itself passes through LeakyReLU. */
static void head_theta(const float *const *params, T x_s8, int ngf, float *theta) {
    T f = t_conv(x_s8, PW(P_FLATTEN), PB(P_FLATTEN), ngf * 8, 2, 1, 0, ORC_ACT_LRELU);
    T l = t_conv(f, PW(P_LINEAR), PB(P_LINEAR), 6, 1, 1, 0, ORC_ACT_LRELU);
    memcpy(theta, l.d, sizeof(float) * (size_t)g_N * 6);
    t_free(&f);
    t_free(&l);
}
/* residual = tanh(out(x)) where `out` already ends in tanh (lib/networks_cascading.py:128,174,198,219),
 * returned permuted to N,H,W,2 (:235-237). */
static void head_residual(const float *const *params, T x, float *res_nhwc2) {
    T o = t_conv(x, PW(P_OUT), PB(P_OUT), 2, 3, 1, 1, ORC_ACT_TANH);
    const size_t hw = (size_t)o.h * o.w;
    for (int n = 0; n < g_N; ++n)
        for (size_t i = 0; i < hw; ++i) {
            res_nhwc2[((size_t)n * hw + i) * 2 + 0] = tanhf(o.d[((size_t)n * 2 + 0) * hw + i]);
            res_nhwc2[((size_t)n * hw + i) * 2 + 1] = tanhf(o.d[((size_t)n * 2 + 1) * hw + i]);
        }
    t_free(&o);
}

/* probes: optional taps of intermediate activations for the golden tests.  probe_ids selects tensors
 * by the reference's variable name order below; each probe writes (sum, abs-sum) as doubles. */
typedef struct {
    double sum, abssum;
} orc_probe;
static void probe(orc_probe *pr, int idx, T t) {
    if (!pr) return;
    double s = 0, a = 0;
    const size_t n = (size_t)g_N * t.c * t.h * t.w;
    for (size_t i = 0; i < n; ++i) s += t.d[i], a += fabs((double)t.d[i]);
    pr[idx].sum = s, pr[idx].abssum = a;
}
#define ORC_NUM_PROBES 12
/* probe order: x11,x14,x18,x177,x122,x22,x25,x28,x277,x222,x38,x322 */

/*
 * UnetGenerator.forward(input1, is_training)   lib/networks_cascading.py:152-237
 *   params : 92 pointers, state-dict order, torch layouts (conv OIHW, convT IOHW)
 *   x      : N x input_nc x S x S  (S must be 256: 7 stride-2 levels + the 2x2 flatten conv)
 *   grids  : is_training ? [3][N,S,S,2] (residual+affine per stage) : [1][N,S,S,2] (stage 3)
 *   resid  : is_training ? [3][N,S,S,2] : may be NULL
 *   thetas : [3][N,6] (always written)
 *   probes : NULL or ORC_NUM_PROBES entries
 */
int orc_netg_forward(const float *const *params, const float *x, int N, int input_nc, int ngf, int S,
                     int is_training, int align_corners, float *grids, float *resid, float *thetas,
                     orc_probe *probes) {
    if (S != 256 || N <= 0) return -1;
    g_N = N;
    const size_t gsz = (size_t)N * S * S * 2;
    const int g = ngf;
    T in;
    in.d = (float *)x, in.c = input_nc, in.h = S, in.w = S;

    /* stage 1  :153-160 */
    T x11 = t_conv(in, PW(P_TRANSFER), PB(P_TRANSFER), g, 5, 1, 2, ORC_ACT_LRELU);
    T x12 = blk_down(params, P_DOWN1 + 0, x11, g);
    T x13 = blk_down(params, P_DOWN1 + 2, x12, 2 * g);
    T x14 = blk_down(params, P_DOWN1 + 4, x13, 4 * g);
    T x15 = blk_down(params, P_DOWN1 + 6, x14, 4 * g);
    T x16 = blk_down(params, P_DOWN1 + 8, x15, 4 * g);
    T x17 = blk_down(params, P_DOWN1 + 10, x16, 4 * g);
    T x18 = blk_down(params, P_DOWN1 + 12, x17, 4 * g);
    probe(probes, 0, x11), probe(probes, 1, x14), probe(probes, 2, x18);
    head_theta(params, x18, g, thetas + 0 * (size_t)N * 6); /* :162-164 */

    T x177 = blk_up(params, 7, x18, &x17, 4 * g); /* :166-171 */
    T x166 = blk_up(params, 6, x177, &x16, 4 * g);
    T x155 = blk_up(params, 5, x166, &x15, 4 * g);
    T x144 = blk_up(params, 4, x155, &x14, 4 * g);
    T x133 = blk_up(params, 3, x144, &x13, 2 * g);
    T x122 = blk_up(params, 2, x133, &x12, g);
    probe(probes, 3, x177), probe(probes, 4, x122);
    float *res[3] = {NULL, NULL, NULL};
    float *tmp_res = NULL;
    if (is_training) {
        res[0] = resid, res[1] = resid + gsz, res[2] = resid + 2 * gsz;
        T x111 = blk_up(params, 1, x122, NULL, g); /* :173 */
        head_residual(params, x111, res[0]);        /* :174 */
        t_free(&x111);
    } else {
        tmp_res = (float *)malloc(sizeof(float) * gsz);
        res[2] = tmp_res;
    }

    /* stage 2  :178-198 */
    T x22 = blk_down_bottom(params, 1, NULL, x11, g);
    T x23 = blk_down_bottom(params, 2, &x22, x12, 2 * g);
    T x24 = blk_down_bottom(params, 3, &x23, x13, 4 * g);
    T x25 = blk_down_bottom(params, 4, &x24, x14, 4 * g);
    T x26 = blk_down_bottom(params, 5, &x25, x15, 4 * g);
    T x27 = blk_down_bottom(params, 6, &x26, x16, 4 * g);
    T x28 = blk_down_bottom(params, 7, &x27, x17, 4 * g);
    probe(probes, 5, x22), probe(probes, 6, x25), probe(probes, 7, x28);
    head_theta(params, x28, g, thetas + 1 * (size_t)N * 6);

    T x277 = blk_up_bottom(params, 7, x18, x28, &x27, 4 * g);
    T x266 = blk_up_bottom(params, 6, x177, x277, &x26, 4 * g);
    T x255 = blk_up_bottom(params, 5, x166, x266, &x25, 4 * g);
    T x244 = blk_up_bottom(params, 4, x155, x255, &x24, 4 * g);
    T x233 = blk_up_bottom(params, 3, x144, x244, &x23, 2 * g);
    T x222 = blk_up_bottom(params, 2, x133, x233, &x22, g);
    probe(probes, 8, x277), probe(probes, 9, x222);
    if (is_training) {
        T x211 = blk_up_bottom(params, 1, x122, x222, NULL, g); /* :197 (x_before unused, out=True) */
        head_residual(params, x211, res[1]);
        t_free(&x211);
    }

    /* stage 3  :200-219 -- same weights as stage 2 */
    T x32 = blk_down_bottom(params, 1, NULL, x11, g);
    T x33 = blk_down_bottom(params, 2, &x32, x22, 2 * g);
    T x34 = blk_down_bottom(params, 3, &x33, x23, 4 * g);
    T x35 = blk_down_bottom(params, 4, &x34, x24, 4 * g);
    T x36 = blk_down_bottom(params, 5, &x35, x25, 4 * g);
    T x37 = blk_down_bottom(params, 6, &x36, x26, 4 * g);
    T x38 = blk_down_bottom(params, 7, &x37, x27, 4 * g);
    probe(probes, 10, x38);
    head_theta(params, x38, g, thetas + 2 * (size_t)N * 6);

    T x377 = blk_up_bottom(params, 7, x28, x38, &x37, 4 * g);
    T x366 = blk_up_bottom(params, 6, x277, x377, &x36, 4 * g);
    T x355 = blk_up_bottom(params, 5, x266, x366, &x35, 4 * g);
    T x344 = blk_up_bottom(params, 4, x255, x355, &x34, 4 * g);
    T x333 = blk_up_bottom(params, 3, x244, x344, &x33, 2 * g);
    T x322 = blk_up_bottom(params, 2, x233, x333, &x32, g);
    probe(probes, 11, x322);
    T x311 = blk_up_bottom(params, 1, x222, x322, NULL, g); /* :218 */
    head_residual(params, x311, res[2]);                    /* :219 */

    /* outputs :235-237: residual (N,H,W,2) + affine_grid(theta) */
    float *aff = (float *)malloc(sizeof(float) * gsz);
    for (int st = is_training ? 0 : 2; st < 3; ++st) {
        orc_affine_grid(thetas + (size_t)st * N * 6, aff, N, S, S, align_corners);
        float *dst = is_training ? grids + (size_t)st * gsz : grids;
        for (size_t i = 0; i < gsz; ++i) dst[i] = res[st][i] + aff[i];
    }
    free(aff);
    free(tmp_res);

    T *all[] = {&x11, &x12, &x13, &x14, &x15, &x16, &x17, &x18, &x177, &x166, &x155, &x144, &x133, &x122,
                &x22, &x23, &x24, &x25, &x26, &x27, &x28, &x277, &x266, &x255, &x244, &x233, &x222,
                &x32, &x33, &x34, &x35, &x36, &x37, &x38, &x377, &x366, &x355, &x344, &x333, &x322, &x311};
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) t_free(all[i]);
    return 0;
}

#ifdef __cplusplus
}
#endif

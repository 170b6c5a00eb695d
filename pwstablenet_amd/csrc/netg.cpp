// Host-side executor of the cascading generator: sequences the HIP kernels of this library on one stream,
// with all activations in a caller-provided arena.  Mirrors UnetGenerator.forward of the reference
// (lib/networks_cascading.py:152-237) -- variable names below are the reference's.
//
// Differences in HOW (not in WHAT):
//   * activations are NHWC; a torch.cat([a, b], 1) is a list of (pointer, channels, stride) sources consumed
//     directly by the next convolution -- the reference's 43-44 concat copies per forward disappear;
//   * stage 3's first block (`x32 = down_bottom1(None, x11)`, reference :200) has the same weights and the same
//     input as stage 2's (`x22`, :178) and is bit-identical, so it is computed once;
//   * `out` conv + tanh + tanh + permute + affine_grid + add run as one kernel (field head).
// No allocation, no synchronisation: every launch goes to `stream`, so the whole forward can be captured
// into a hipGraph by the caller.
#include <vector>

#include "common.h"

namespace pws {

struct Layer {
    int kind, cin, cout;
    size_t w_off, b_off;  // float offsets into the packed buffer
};

enum {
    L_TRANSFER = 0,
    L_DOWN1 = 1,   // down1..down7 = 1..7
    L_UP7 = 8,     // up7..up1 = 8..14   (level l -> 8 + (7 - l))
    L_OUT = 15,
    L_DB1_CS = 16,  // down_bottom k: conv_same 16+2(k-1), mpconv 17+2(k-1)
    L_UB7_MP = 30,  // up_bottom level l: mpconv 30+2(7-l), conv_same 31+2(7-l)
    L_FLATTEN = 44,
    L_LINEAR = 45,
    L_COUNT = 46
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Same registration order as the reference's __init__ (lib/networks_cascading.py:112-149) / spec.py.
static std::vector<Layer> build_layers(int input_nc, int g, size_t *total_floats) {
    std::vector<Layer> L;
    size_t off = 0;
    auto add = [&](int kind, int cin, int cout) {
        Layer l{kind, cin, cout, 0, 0};
        l.w_off = off;
        off = align_up(off + pws_packed_weight_floats(kind, cin, cout), 64);
        l.b_off = off;
        off = align_up(off + (size_t)cout, 64);
        L.push_back(l);
    };
    const int enc[7][2] = {{g, g}, {g, 2 * g}, {2 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}};
    add(PWS_CONV_K5S1, input_nc, g);
    for (int i = 0; i < 7; ++i) add(PWS_CONV_K3S2, enc[i][0], enc[i][1]);
    const int dec[7][2] = {{4 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 2 * g}, {4 * g, g}, {2 * g, g}};
    for (int j = 0; j < 7; ++j) add(PWS_CONVT_K4S2, dec[j][0], dec[j][1]);  // up7..up1
    add(PWS_CONV_K3S1_OUT, g, 2);
    for (int i = 0; i < 7; ++i) {
        add(PWS_CONV_K3S1, enc[i][0], enc[i][0]);
        add(PWS_CONV_K3S2, i == 0 ? enc[i][0] : 2 * enc[i][0], enc[i][1]);
    }
    const int ub[7][3] = {{4 * g, 4 * g, 8 * g}, {8 * g, 4 * g, 16 * g}, {8 * g, 4 * g, 16 * g}, {8 * g, 4 * g, 16 * g},
                          {8 * g, 2 * g, 16 * g}, {4 * g, g, 8 * g}, {2 * g, g, 4 * g}};  // (input_nc, output_nc, inner_nc)
    for (int j = 0; j < 7; ++j) {
        add(PWS_CONVT_K4S2, ub[j][2], ub[j][1]);
        add(PWS_CONVT_K3S1, ub[j][0], ub[j][0]);
    }
    add(PWS_CONV_K2S1P0, 4 * g, 8 * g);
    add(PWS_CONV_K1, 8 * g, 6);
    if (total_floats) *total_floats = off;
    return L;
}

struct Seg {
    float *ptr;
    int c, ld;
};
struct Tn {  // a (virtually concatenated) NHWC tensor
    Seg seg[4];
    int nseg, h, w;
    int channels() const {
        int c = 0;
        for (int i = 0; i < nseg; ++i) c += seg[i].c;
        return c;
    }
};

class Exec {
  public:
    Exec(const float *packed, const std::vector<Layer> &layers, int n, char *ws, size_t ws_bytes, hipStream_t st, bool dry)
        : packed_(packed), L_(layers), n_(n), ws_(ws), cap_(ws_bytes), st_(st), dry_(dry) {}

    size_t used() const { return off_; }
    int rc() const { return rc_; }

    float *alloc(size_t floats) {
        const size_t bytes = align_up(floats * sizeof(float), 256);
        const size_t at = off_;
        off_ += bytes;
        if (dry_) return nullptr;
        if (off_ > cap_) {
            if (rc_ == PWS_OK) {
                set_error("pws_netg_forward: workspace too small (%zu B needed so far, %zu B given)", off_, cap_);
                rc_ = PWS_ENOMEM;
            }
            return nullptr;
        }
        return reinterpret_cast<float *>(ws_ + at);
    }

    static Tn cat(const Tn &a, const Tn &b) {  // torch.cat([a, b], dim=1)
        Tn o = a;
        for (int i = 0; i < b.nseg && o.nseg < 4; ++i) o.seg[o.nseg++] = b.seg[i];
        return o;
    }

    Tn conv(int layer, const Tn &x, int act, const float *nchw_src = nullptr, int nchw_c = 0) {
        const Layer &l = L_[layer];
        int oh = x.h, ow = x.w;
        if (l.kind == PWS_CONV_K3S2) oh = (x.h - 1) / 2 + 1, ow = (x.w - 1) / 2 + 1;
        if (l.kind == PWS_CONVT_K4S2) oh = 2 * x.h, ow = 2 * x.w;
        Tn o{};
        o.nseg = 1, o.h = oh, o.w = ow;
        o.seg[0] = Seg{alloc((size_t)n_ * oh * ow * l.cout), l.cout, l.cout};
        if (dry_ || rc_ != PWS_OK) return o;
        pws_conv_args a{};
        a.kind = l.kind, a.n = n_, a.h = x.h, a.w = x.w;
        if (nchw_src) {
            a.nsrc = 1, a.src_nchw = 1, a.src[0] = pws_src{nchw_src, nchw_c, 0};
        } else {
            a.nsrc = x.nseg;
            for (int i = 0; i < x.nseg; ++i) a.src[i] = pws_src{x.seg[i].ptr, x.seg[i].c, x.seg[i].ld};
        }
        a.cout = l.cout, a.w_packed = packed_ + l.w_off, a.bias = packed_ + l.b_off, a.act = act;
        a.out = o.seg[0].ptr, a.out_ld = l.cout;
        a.ws = splitk_ws_, a.ws_bytes = splitk_bytes_;
        g_prof_tag = layer;
        rc_ = pws_conv2d_fwd(&a, st_);
        g_prof_tag = -1;
        return o;
    }

    // down.forward (reference :262-264)
    Tn down(int i, const Tn &x) { return conv(L_DOWN1 + (i - 1), x, PWS_ACT_LRELU); }
    // down_bottom.forward (reference :291-298)
    Tn down_bottom(int k, const Tn *x_left, const Tn &x_up) {
        Tn c = conv(L_DB1_CS + 2 * (k - 1), x_up, PWS_ACT_LRELU);
        return conv(L_DB1_CS + 2 * (k - 1) + 1, x_left ? cat(*x_left, c) : c, PWS_ACT_LRELU);
    }
    // up.forward (reference :316-321)
    Tn up(int level, const Tn &x1, const Tn *x2) {
        Tn u = conv(L_UP7 + (7 - level), x1, PWS_ACT_RELU);
        return x2 ? cat(u, *x2) : u;
    }
    // up_bottom.forward (reference :344-350)
    Tn up_bottom(int level, const Tn &x_up, const Tn &x_left, const Tn *x_before) {
        Tn e = conv(L_UB7_MP + 2 * (7 - level) + 1, x_up, PWS_ACT_RELU);
        Tn v = conv(L_UB7_MP + 2 * (7 - level), cat(e, x_left), PWS_ACT_RELU);
        return x_before ? cat(v, *x_before) : v;
    }
    // theta = linear(flatten(x)) (reference :162-163)
    void theta(const Tn &x_s8, float *theta_out) {
        if (dry_ || rc_ != PWS_OK) return;
        const Layer &f = L_[L_FLATTEN], &l = L_[L_LINEAR];
        rc_ = pws_theta_head_fwd(x_s8.seg[0].ptr, n_, x_s8.seg[0].c, f.cout, packed_ + f.w_off, packed_ + f.b_off,
                                 packed_ + l.w_off, packed_ + l.b_off, theta_ws_, theta_out, st_);
    }
    // tanh(out(x)).permute(0,2,3,1) [+ affine_grid(theta)] (reference :174,235-237)
    void field(const Tn &x, const float *theta_k, int ac, float *resid, float *grid) {
        if (dry_ || rc_ != PWS_OK) return;
        const Layer &o = L_[L_OUT];
        rc_ = pws_field_head_fwd(x.seg[0].ptr, x.seg[0].ld, n_, x.h, x.w, x.seg[0].c, packed_ + o.w_off, packed_ + o.b_off,
                                 theta_k, ac, resid, grid, st_);
    }

    // scratch shared by all layers (launches are stream-ordered): split-K partial tiles and the theta head's partials
    void reserve_scratch(int ngf) {
        splitk_bytes_ = (size_t)(n_ > 8 ? n_ : 8) * (2u << 20);
        splitk_ws_ = alloc(splitk_bytes_ / sizeof(float));
        theta_ws_ = alloc(pws_theta_head_ws_floats(n_, 4 * ngf, 8 * ngf));
        if (!splitk_ws_) splitk_bytes_ = 0;
    }

  private:
    float *splitk_ws_ = nullptr, *theta_ws_ = nullptr;
    size_t splitk_bytes_ = 0;
    const float *packed_;
    const std::vector<Layer> &L_;
    int n_;
    char *ws_;
    size_t cap_, off_ = 0;
    hipStream_t st_;
    bool dry_;
    int rc_ = PWS_OK;
};

static int run_forward(const float *packed, const float *x, int n, int input_nc, int g, int is_training, int ac, char *ws,
                       size_t ws_bytes, float *grids, float *resid, float *thetas, hipStream_t st, bool dry, size_t *used) {
    const int S = 256;
    size_t total = 0;
    const std::vector<Layer> layers = build_layers(input_nc, g, &total);
    Exec E(packed, layers, n, ws, ws_bytes, st, dry);
    E.reserve_scratch(g);
    const size_t gsz = (size_t)n * S * S * 2;
    float *th = thetas ? thetas : E.alloc((size_t)3 * n * 6);
    if (thetas == nullptr && !dry && E.rc() != PWS_OK) return E.rc();
    float *th1 = th, *th2 = th ? th + (size_t)n * 6 : nullptr, *th3 = th ? th + (size_t)2 * n * 6 : nullptr;

    Tn in{};
    in.nseg = 1, in.h = S, in.w = S, in.seg[0] = Seg{nullptr, input_nc, 0};
    // ---- stage 1 (reference :153-174)
    Tn x11 = E.conv(L_TRANSFER, in, PWS_ACT_LRELU, x, input_nc);
    Tn x12 = E.down(1, x11), x13 = E.down(2, x12), x14 = E.down(3, x13), x15 = E.down(4, x14);
    Tn x16 = E.down(5, x15), x17 = E.down(6, x16), x18 = E.down(7, x17);
    E.theta(x18, th1);
    Tn x177 = E.up(7, x18, &x17), x166 = E.up(6, x177, &x16), x155 = E.up(5, x166, &x15);
    Tn x144 = E.up(4, x155, &x14), x133 = E.up(3, x144, &x13), x122 = E.up(2, x133, &x12);
    if (is_training) {
        Tn x111 = E.up(1, x122, nullptr);
        E.field(x111, th1, ac, resid, grids);
    }
    // ---- stage 2 (reference :178-198)
    Tn x22 = E.down_bottom(1, nullptr, x11);
    Tn x23 = E.down_bottom(2, &x22, x12), x24 = E.down_bottom(3, &x23, x13), x25 = E.down_bottom(4, &x24, x14);
    Tn x26 = E.down_bottom(5, &x25, x15), x27 = E.down_bottom(6, &x26, x16), x28 = E.down_bottom(7, &x27, x17);
    E.theta(x28, th2);
    Tn x277 = E.up_bottom(7, x18, x28, &x27), x266 = E.up_bottom(6, x177, x277, &x26);
    Tn x255 = E.up_bottom(5, x166, x266, &x25), x244 = E.up_bottom(4, x155, x255, &x24);
    Tn x233 = E.up_bottom(3, x144, x244, &x23), x222 = E.up_bottom(2, x133, x233, &x22);
    if (is_training) {
        Tn x211 = E.up_bottom(1, x122, x222, nullptr);
        E.field(x211, th2, ac, resid ? resid + gsz : nullptr, grids ? grids + gsz : nullptr);
    }
    // ---- stage 3 (reference :200-219); x32 == x22 (same weights, same input)
    const Tn &x32 = x22;
    Tn x33 = E.down_bottom(2, &x32, x22), x34 = E.down_bottom(3, &x33, x23), x35 = E.down_bottom(4, &x34, x24);
    Tn x36 = E.down_bottom(5, &x35, x25), x37 = E.down_bottom(6, &x36, x26), x38 = E.down_bottom(7, &x37, x27);
    E.theta(x38, th3);
    Tn x377 = E.up_bottom(7, x28, x38, &x37), x366 = E.up_bottom(6, x277, x377, &x36);
    Tn x355 = E.up_bottom(5, x266, x366, &x35), x344 = E.up_bottom(4, x255, x355, &x34);
    Tn x333 = E.up_bottom(3, x244, x344, &x33), x322 = E.up_bottom(2, x233, x333, &x32);
    Tn x311 = E.up_bottom(1, x222, x322, nullptr);
    if (is_training)
        E.field(x311, th3, ac, resid ? resid + 2 * gsz : nullptr, grids ? grids + 2 * gsz : nullptr);
    else
        E.field(x311, th3, ac, nullptr, grids);
    if (used) *used = E.used();
    return E.rc();
}

}  // namespace pws

using namespace pws;

extern "C" size_t pws_netg_packed_floats(int input_nc, int ngf) {
    if (input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0;
    build_layers(input_nc, ngf, &total);
    return total;
}

extern "C" int pws_netg_pack_weights(const float *const *params, float *packed, int input_nc, int ngf, pws_stream_t stream) {
    PWS_REQUIRE(params && packed, "pws_netg_pack_weights: NULL pointer");
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_pack_weights: ngf must be a positive multiple of 16 (got %d)",
                ngf);
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    hipError_t e = hipMemsetAsync(packed, 0, total * sizeof(float), as_stream(stream));
    if (e != hipSuccess) {
        set_error("pws_netg_pack_weights: hipMemsetAsync: %s", hipGetErrorString(e));
        return PWS_EHIP;
    }
    for (int i = 0; i < L_COUNT; ++i) {
        PWS_REQUIRE(params[2 * i] && params[2 * i + 1], "pws_netg_pack_weights: params[%d] is NULL", 2 * i);
        int rc = pws_pack_conv_weight(params[2 * i], packed + L[i].w_off, L[i].kind, L[i].cin, L[i].cout, stream);
        if (rc != PWS_OK) return rc;
        e = hipMemcpyAsync(packed + L[i].b_off, params[2 * i + 1], sizeof(float) * L[i].cout, hipMemcpyDeviceToDevice,
                           as_stream(stream));
        if (e != hipSuccess) {
            set_error("pws_netg_pack_weights: hipMemcpyAsync(bias %d): %s", i, hipGetErrorString(e));
            return PWS_EHIP;
        }
    }
    return PWS_OK;
}

extern "C" size_t pws_netg_workspace_bytes(int n, int input_nc, int ngf, int is_training) {
    if (n <= 0 || input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t used = 0;
    run_forward(nullptr, nullptr, n, input_nc, ngf, is_training, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, true, &used);
    return used;
}

extern "C" int pws_netg_forward(const float *packed, const float *x, int n, int input_nc, int ngf, int is_training,
                                int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid, float *thetas,
                                pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_forward: bad n/input_nc/ngf %d/%d/%d", n, input_nc,
                ngf);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(packed && x && ws && grids, "pws_netg_forward: NULL pointer");
    PWS_REQUIRE(!is_training || resid, "pws_netg_forward: resid must be given when is_training");
    PWS_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "pws_netg_forward: workspace must be 256-byte aligned");
    return run_forward(packed, x, n, input_nc, ngf, is_training, align_corners, static_cast<char *>(ws), ws_bytes, grids, resid,
                       thetas, as_stream(stream), false, nullptr);
}

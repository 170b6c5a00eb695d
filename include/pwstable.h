/*
 * pwstable.h -- C ABI of libpwstable_hip.so: the MI355X (gfx950) implementation of PWStableNet's hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only (no torch types); every pointer is a
 * DEVICE pointer unless stated otherwise; every launch goes to the caller's HIP stream (`stream` is a
 * hipStream_t passed as void*, NULL = the default stream); nothing here allocates, frees or
 * synchronises -- scratch comes from the caller (`ws`), sized by the matching *_workspace_bytes().
 * Return value: 0 on success, negative errno-style code otherwise (pws_last_error() has the text).
 * Re-entrant per stream and per thread.  Global mutable state: the thread-local error string, the process-wide DEFAULTS of
 * pws_set_option() (read only by the entry points that take no pws_netg_opts; the *_opts entry points carry math / store / queue
 * mode in their arguments and read no global), the per-device side queue of the generator forward / backward (one stream per device
 * and process, shared by the EAGER calls of all host threads; every thread owns the events it orders it with -- one pool per CALLER
 * STREAM, so that calls a thread makes for different streams, as autograd's worker thread does, never re-record each other's events --
 * and a call made inside a hipGraph capture forks into a private stream of the calling thread instead, so the shared queue is never
 * part of anybody's capture) and the measurement hooks (pws_prof_*, off by default).
 * What is exercised (tests/test_hip_threads.py, results bit-equal to serial execution): concurrent eager forwards of two generators
 * and of ONE generator from two threads / streams, concurrent replays of graphs, two whole fp32 training steps beside graph replays.
 * NOT bit-reproducible: bf16-math work of one stream beside other GPU work.  While a kernel that leaves room on its CUs executes
 * v_mfma_f32_32x32x16_bf16 (conv_bf16_kernel above all), kernels of other streams -- of other PROCESSES too -- that keep many registers
 * live over long gather sequences compute other values in a few lanes (measured down to that one instruction:
 * profiles/r04_cross_stream_interference.txt; below this library's level, not host-side).  The effect is local to a compute unit: tenants
 * whose streams own DISJOINT CUs (hipExtStreamCreateWithCUMask; the generator on one queue, pws_netg_opts.two_queues = 0, since the side
 * queue is not masked) are bit-reproducible again -- 0 of 150 three-thread rounds against 1 in 10.  One training process per GPU, the
 * deployment this library is built for, never overlaps the two kinds of kernel in the first place.
 * Graph CAPTURE is the caller's: capture after one eager call on the same thread (the library makes its private stream then,
 * not while the capture is open) and while no other host thread issues GPU work -- on ROCm 7.0 captures that overlapped another
 * thread's capture, device-wide synchronisation or training step ended invalidated, crashed inside the runtime, or (1 run in 12)
 * perturbed the other thread's gradient sums (gpurun_out/r4c, tools/probes/thread_race_probe.py).
 *
 * Reference interfaces replaced (paths relative to the mindazhao/PWStableNet checkout; the reference has
 * no native code -- each entry point replaces the PyTorch/ATen op the reference dispatches at that line):
 *   pws_netg_forward              UnetGenerator.forward            lib/networks_cascading.py:152-237
 *   pws_conv2d_fwd                down / down_bottom / up / up_bottom blocks        :245-350
 *                                 (nn.Conv2d :248,269,274,285 ; nn.ConvTranspose2d :306,330,339 ;
 *                                  LeakyReLU :250,271 ; ReLU :305,328 ; torch.cat :296,321,346-350)
 *   pws_theta_head_fwd            flatten + linear -> theta        :148-149,162-163,186-187,208-209
 *   pws_field_head_fwd            out conv + tanh + tanh, permute, + F.affine_grid  :128,164,174,235-237
 *   pws_affine_grid               F.affine_grid                    :164,188,210 ; main_new.py:195
 *   pws_grid_sample_fwd / _bwd    F.grid_sample (+ autograd)       main_new.py:106,109,116,118,197,214,716
 *   pws_upsample_bilinear_ac      UpsamplingBilinear2d(size=...)   main_new.py:706-710
 *   pws_upsample_grid_sample_fwd  the two above fused (720p path)  main_new.py:706-716
 *   pws_adam_step                 optim.Adam(...).step()           main_new.py:63,216
 *   pws_pack_* / pws_netg_pack_weights   state_dict (OIHW / IOHW) -> kernel layout; main_new.py:60,471
 *
 * Layouts.  Frames / images: NCHW fp32 as in the reference.  Warp fields: N,H,W,2 fp32 (x then y,
 * normalised to [-1,1]) as in the reference.  Activations inside the generator: NHWC fp32 with an
 * explicit pixel stride `ld` (floats), so a torch.cat of feature maps is a list of sources, never a copy.
 */
#ifndef PWSTABLE_H
#define PWSTABLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWS_VERSION 5   /* 2: pws_netg_opts carries flags + x_sample_stride (round 3); 4: dpacked is the compact gradient slab
                          (pws_netg_grad_floats / pws_netg_grad_layout), pws_netg_backward_lists (round 4); 5: pws_netg_backward_plan,
                          pws_netg_pack_weights_train (round 5; additions only) */

#define PWS_OK 0
#define PWS_EINVAL (-22) /* bad argument / unsupported shape */
#define PWS_ENOMEM (-12) /* caller's workspace too small */
#define PWS_EHIP (-5)    /* HIP runtime reported an error at launch */

typedef void *pws_stream_t; /* hipStream_t */

int pws_version(void);
const char *pws_last_error(void); /* thread-local, never NULL */
/* Process-wide options.  PWS_OPT_TWO_QUEUES (default 1): the generator forward forks an internal second queue so that
 * stage k+1's encoder runs beside stage k's decoder; 0 = issue everything on the caller's stream (clean per-kernel timing). */
#define PWS_OPT_TWO_QUEUES 1
/* PWS_OPT_MATH (default PWS_MATH_FP32): arithmetic of the generator's conv / transposed-conv contractions inside
 * pws_netg_forward / pws_netg_backward.  PWS_MATH_BF16: operands rounded to bf16 (round-to-nearest-even) as they are
 * staged into LDS, products accumulated in fp32 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16); activations, biases,
 * gradients and master weights stay fp32 in memory.  (BASELINE configs 3/4: "bf16 with MFMA convs"; tolerance vs fp32
 * stated in tests/test_hip_bf16.py.)  Layers the bf16 kernels do not cover (first layer: 31-channel NCHW window; heads)
 * keep fp32 arithmetic. */
#define PWS_OPT_MATH 2
#define PWS_MATH_FP32 0
#define PWS_MATH_BF16 1
/* Element type of activations / gradients in HBM for the bf16 path (per call: the `store` fields; whole generator:
 * PWS_OPT_STORE, default PWS_STORE_FP32).  PWS_STORE_BF16 needs PWS_MATH_BF16. */
/* Whole generator (pws_netg_forward / _backward): with PWS_OPT_MATH = BF16 and PWS_OPT_STORE = BF16 (and ngf % 32 == 0) every
 * activation and activation gradient inside the arena is bf16; inputs, fields, thetas, weights and weight gradients stay fp32.
 * The arena size does not depend on the option (the bf16 tensors use half of their slots). */
#define PWS_OPT_STORE 3
#define PWS_STORE_FP32 0
#define PWS_STORE_BF16 1
/* Measurement only: selects alternative kernel variants that the scripts under tools/ compare (0 = the product default).
 * 1: non-temporal loads / stores in grid_sample at any size; 2: the general grid_sample backward kernel also for the field-only
 * gradient; 9: sub-pixel conv classes as grid.z planes (not on consecutive block slots of one XCD); 11: no fused act' / bias
 * gradient in the bf16 generator backward (separate pws_act_bwd_bias passes); 20: first-generation conv_bf16_kernel instead of
 * the persistent LDS-ring kernel (conv_ring.hip); 21: the ring kernel also for launches too small to fill the chip; 22: the
 * first-generation conv_mfma_kernel instead of the fp32 ring kernel (conv_ring_f32.hip); 23 / 24: the fp32 ring kernel also for
 * small launches, with 8 x 32 / 16 x 32 tiles; 41..47:
 * timing-only ablations of the ring kernel (bit 0: the DMA pieces fetch nothing, bit 1: no matrix phase, bit 2: no epilogue --
 * results are meaningless). */
#define PWS_OPT_EXPERIMENT 100
int pws_set_option(int key, int value);
int pws_get_option(int key); /* current value, or PWS_EINVAL */
/* Number of compute units / XCDs the library sees on the current device (diagnostics). */
int pws_device_info(int *compute_units, int *arch_is_gfx950);

/* ---------------------------------------------------------------- activations fused into conv epilogues */
#define PWS_ACT_NONE 0
#define PWS_ACT_LRELU 1 /* LeakyReLU(0.2) */
#define PWS_ACT_RELU 2

/* ---------------------------------------------------------------- conv kinds (what the generator uses) */
#define PWS_CONV_K3S1 0  /* Conv2d k3 s1 p1            (down_bottom*.conv_same)                 */
#define PWS_CONV_K3S2 1  /* Conv2d k3 s2 p1            (down*, down_bottom*.mpconv)             */
#define PWS_CONV_K5S1 2  /* Conv2d k5 s1 p2            (transfer)                               */
#define PWS_CONVT_K3S1 3 /* ConvTranspose2d k3 s1 p1   (up_bottom*.conv_same) == flipped conv   */
#define PWS_CONVT_K4S2 4 /* ConvTranspose2d k4 s2 p1   (up*, up_bottom*.mpconv), 4 sub-pixel 2x2 convs */
#define PWS_CONV_K2S1P0 5 /* Conv2d k2 s1 p0 (flatten) -- packing only; run by pws_theta_head_fwd */
#define PWS_CONV_K1 6     /* Conv2d k1 (linear)        -- packing only                           */
#define PWS_CONV_K3S1_OUT 7 /* Conv2d k3 s1 p1, Cout=2 (out) -- packing only; run by pws_field_head_fwd */

/* Packed size (floats) of one layer's weight for the kernels: [class][tap][cin padded to 16][cout]. */
size_t pws_packed_weight_floats(int kind, int cin, int cout);
/* w_torch: OIHW (conv kinds) or IOHW (convT kinds), dense.  w_packed: pws_packed_weight_floats() floats. */
int pws_pack_conv_weight(const float *w_torch, float *w_packed, int kind, int cin, int cout,
                         pws_stream_t stream);

/* Winograd F(2x2,3x3) weights U = G g G^T as [16][cin padded to 16][cout], computed from the PACKED weights of a
 * K3S1 / CONVT_K3S1 layer (so the transposed conv's tap flip is already applied). */
size_t pws_packed_wino_floats(int cin, int cout);
int pws_pack_conv_weight_wino(const float *w_packed, float *w_wino, int cin, int cout, pws_stream_t stream);
/* The same for ConvTranspose2d k4 s2 p1 (CONVT_K4S2): each of the 4 output parity classes is a 2x2 correlation, run as
 * Winograd F(3x3,2x2) on 4x4 input patches; U as [4 classes][16][cin padded to 16][cout] from the PACKED weights. */
size_t pws_packed_wino_ct4_floats(int cin, int cout);
int pws_pack_conv_weight_wino_ct4(const float *w_packed, float *w_wino, int cin, int cout, pws_stream_t stream);

/* Winograd weights in the layout of the persistent LDS-ring kernel (second generation, taken when the map consists of whole
 * 16 x 32-pixel units and fills the chip): F(2x2,3x3) for K3S1 / CONVT_K3S1 (16 components), F(2x2,2x2) per output parity class
 * for CONVT_K4S2 (2 x 18 components), from the PACKED weights.  Needs cout % 32 == 0 (pws_packed_wring_floats returns 0
 * otherwise: leave pws_conv_args.w_wring NULL).
 * Round 5: also K5S1 with 17..32 input channels and exactly 64 output channels -- the generator's first layer (reference
 * lib/networks_cascading.py:21-23) as Winograd F(2x2,5x5), 36 components on the points {0, 1, -1, 2, -1/2, inf}, for fp32 NCHW
 * sources (pws_conv_args.src_nchw) whose maps are whole 8 x 16-pixel units, at least two per compute unit. */
size_t pws_packed_wring_floats(int kind, int cin, int cout);
int pws_pack_conv_weight_wring(const float *w_packed, float *w_wring, int kind, int cin, int cout, pws_stream_t stream);

/* bf16 copy of a packed weight for the bf16 matrix-core kernels: w_packed is [planes][krows][ncols] fp32 (the forward
 * layout: planes = taps (x4 classes for CONVT_K4S2), krows = cin padded to 16, ncols = cout; or the data-gradient layout:
 * planes = gradient taps, krows = cout, ncols = cin).  Result: [planes][ncols padded to 64][krows padded to 32] bf16
 * (k contiguous: one MFMA B fragment = 8 consecutive k = one 16-byte read), zero padded.  Size in FLOATS (2 bf16 each). */
size_t pws_packed_bf16_floats(int planes, int krows, int ncols);
int pws_pack_weight_bf16(const float *w_packed, void *w_bf16, int planes, int krows, int ncols, pws_stream_t stream);

/* NCHW fp32 [n,c,h,w] -> NHWC fp32 [n,h,w,cpad], channels c..cpad-1 zero (cpad a multiple of 4, c <= cpad <= 32): the bf16
 * first layer reads the 31-channel window (main_new.py:650) as a 32-channel NHWC source. */
int pws_nchw_to_nhwc_pad(const float *x, float *out, int n, int c, int h, int w, int cpad, pws_stream_t stream);
/* ..._s variants: `store` = PWS_STORE_FP32 | PWS_STORE_BF16 selects the element type of the NHWC activation / gradient
 * tensors (`out` here; x / dx of the heads; dy / y of pws_act_bwd_bias_s), everything else is unchanged and fp32. */
int pws_nchw_to_nhwc_pad_s(const float *x, float *out, int n, int c, int h, int w, int cpad, int store, pws_stream_t stream);
/* dst[i] = float(src[i]) for `count` bf16 values; dst (bf16, `count` even) = or += bf16(src[i]). */
int pws_cvt_bf16_to_f32(const void *src, float *dst, size_t count, pws_stream_t stream);
int pws_cvt_f32_to_bf16(const float *src, void *dst, size_t count, int accumulate, pws_stream_t stream);

/* One NHWC source of a (virtually concatenated) conv input: `channels` channels starting at `ptr`,
 * consecutive pixels `ld` floats apart.  ptr 16-byte aligned, ld % 4 == 0, channels % 16 == 0. */
typedef struct pws_src {
    const float *ptr;
    int channels;
    int ld;
} pws_src;

typedef struct pws_conv_args {
    int kind;         /* PWS_CONV_K3S1 | K3S2 | K5S1 | CONVT_K3S1 | CONVT_K4S2 */
    int n, h, w;      /* batch and INPUT height/width */
    int nsrc;         /* 1..4 sources, concatenated along channels in this order */
    pws_src src[4];
    int src_nchw;     /* 1: src[0] is an NCHW tensor with `channels` channels (any count), planes dense; nsrc==1.  src[0].ld: floats
                         between consecutive SAMPLES (0 = dense = channels * h * w; h * w for overlapping sliding windows) */
    int cout;         /* multiple of 4 */
    const float *w_packed;
    const float *bias; /* cout floats or NULL */
    int act;          /* PWS_ACT_* */
    float *out;       /* NHWC, output height/width implied by kind */
    int out_ld;       /* pixel stride of out in floats (>= cout) */
    const float *w_wino; /* optional: Winograd-domain weights from pws_pack_conv_weight_wino (K3S1 / CONVT_K3S1) or
                            pws_pack_conv_weight_wino_ct4 (CONVT_K4S2); when given and the map is large enough the
                            F(2x2,3x3) / F(3x3,2x2) kernel runs instead of the direct one */
    void *ws;         /* optional scratch for split-K partial tiles (16-B aligned) or NULL: small-spatial layers */
    size_t ws_bytes;  /* then run un-split.  Any size works; 64 x n*oh*ow*cout*4 bytes never limits the split. */
    int math;         /* PWS_MATH_FP32 (0) | PWS_MATH_BF16: needs w_bf16 and every source's channels % 32 == 0,
                         otherwise the launch runs in fp32 */
    const void *w_bf16; /* bf16 weights from pws_pack_weight_bf16(w_packed, planes, cin padded to 16, cout) or NULL */
    int store;        /* PWS_STORE_FP32 (0) | PWS_STORE_BF16: every src[].ptr and `out` then point to bf16 elements (channels /
                         ld / out_ld still count ELEMENTS; ld % 8 == 0, out_ld even).  Needs math == PWS_MATH_BF16 and a kind
                         the bf16 kernels cover.  Halves the activation traffic of the bf16 path. */
    const float *w_wring; /* optional: ring-layout Winograd weights from pws_pack_conv_weight_wring (K3S1 / CONVT_K3S1 / CONVT_K4S2,
                         fp32 NHWC sources; K5S1 of the first layer, fp32 NCHW source): tried before w_wino and the direct kernels */
    void *out_sign;   /* optional, store == PWS_STORE_BF16 and cout % 8 == 0 only: the SIGN BITS of `out`, written beside it -- bit (c & 7)
                         of byte out_sign[pixel * out_sign_ld + c / 8] = (out[pixel][c] > 0), for the rounded bf16 value.  What the
                         backward of a LeakyReLU / ReLU block needs of its forward tensor (pws_dst.act_sign): 1/16 of its bytes. */
    int out_sign_ld;  /* bytes per pixel of out_sign (>= cout / 8) */
} pws_conv_args;

int pws_conv2d_fwd(const pws_conv_args *args, pws_stream_t stream);

/* ---------------------------------------------------------------- backward of the conv blocks
 * (autograd of the reference's loss_g.backward(), main_new.py:214, through nn.Conv2d / nn.ConvTranspose2d /
 *  LeakyReLU / ReLU / torch.cat of lib/networks_cascading.py:245-350)
 *
 * Order per layer:  1. pws_act_bwd_bias   dy <- dy * act'(y) in place, db += sum_pixels dy
 *                   2. pws_conv2d_bwd_weight   dW_packed += x (*) dy
 *                   3. pws_conv2d_bwd_data     dx (scattered over the layer's concat sources) = dy (*) W
 * `kind` is always the FORWARD kind of the layer; n,h,w the forward INPUT extent; cout the forward cout. */

/* dy[i] *= act'(y[i]) (LeakyReLU(0.2): y>0 ? 1 : 0.2 ; ReLU: y>0 ? 1 : 0 ; NONE: 1); dbias[c] += sum over pixels.
 * dy, y: dense NHWC [pixels][c]; dbias (nullable): c floats, ACCUMULATED (fp32 atomics, one per workgroup and channel). */
int pws_act_bwd_bias(float *dy, const float *y, size_t pixels, int c, int act, float *dbias, pws_stream_t stream);
/* ws (optional, 16-byte aligned, pws_act_bwd_bias_ws_bytes(c) bytes): the workgroups' partial bias sums go through slabs
 * and a second small launch instead of contended atomics. */
size_t pws_act_bwd_bias_ws_bytes(int c);
int pws_act_bwd_bias_s(float *dy, const float *y, size_t pixels, int c, int act, float *dbias, int store, void *ws,
                       size_t ws_bytes, pws_stream_t stream);

/* Weights re-laid-out for the data-gradient convolution of a layer.  kind in {K3S1, K3S2, CONVT_K3S1, CONVT_K4S2}. */
size_t pws_packed_dgrad_floats(int kind, int cin, int cout);
int pws_pack_conv_weight_dgrad(const float *w_torch, float *w_packed, int kind, int cin, int cout, pws_stream_t stream);

typedef struct pws_dst {
    float *ptr;      /* NHWC gradient buffer of one forward source */
    int channels;
    int ld;
    int accumulate;  /* 0: overwrite, 1: add to what is there (a tensor consumed by several layers) */
    /* Optional, bf16 storage only (store == PWS_STORE_BF16): this call completes the gradient of the tensor `act_y` (the forward
     * tensor this destination is the gradient of, bf16 NHWC, pixel stride act_y_ld, same channels) -- the epilogue then
     * multiplies the (accumulated) sum by act'(act_y) for act = PWS_ACT_LRELU / PWS_ACT_RELU before the single rounding, which
     * saves the separate pws_act_bwd_bias pass over that tensor (a bias gradient, if one is needed, is then taken with
     * pws_act_bwd_bias_s(..., PWS_ACT_NONE, ...)).  act_y == NULL or act == PWS_ACT_NONE: plain gradient. */
    const void *act_y;
    int act_y_ld;
    int act;
    /* Optional beside act_y: its sign bits as written by pws_conv2d_fwd (pws_conv_args.out_sign, same layout, act_sign_ld bytes per
     * pixel).  A kernel that knows them reads one byte instead of 16 per 8 channels (the persistent ring kernel does when every
     * destination with an act has them); the others read act_y, which must be given as well.  Same result bit for bit. */
    const void *act_sign;
    int act_sign_ld;
} pws_dst;

typedef struct pws_conv_bwd_data_args {
    int kind;
    int n, h, w;       /* forward input extent */
    int cout;          /* forward output channels (multiple of 16) */
    const float *gout; /* dy, NHWC [n, oh, ow, cout], pixel stride gout_ld */
    int gout_ld;
    const float *w_dgrad; /* from pws_pack_conv_weight_dgrad */
    int ndst;          /* 1..4 destinations = the forward sources, in concat order */
    pws_dst dst[4];
    void *ws;          /* optional split-K scratch, as in pws_conv_args */
    size_t ws_bytes;
    int math;          /* as in pws_conv_args (bf16 needs w_dgrad_bf16 and cout % 32 == 0) */
    const void *w_dgrad_bf16; /* pws_pack_weight_bf16(w_dgrad, dgrad taps, cout, cin) or NULL */
    int store;         /* PWS_STORE_BF16: gout and every dst[].ptr hold bf16 elements (accumulation rounds to bf16) */
} pws_conv_bwd_data_args;
int pws_conv2d_bwd_data(const pws_conv_bwd_data_args *args, pws_stream_t stream);

typedef struct pws_conv_bwd_weight_args {
    int kind;          /* K3S1 | K3S2 | K5S1 | CONVT_K3S1 | CONVT_K4S2 */
    int n, h, w;       /* forward input extent */
    int nsrc;          /* forward sources (x), as in pws_conv_args */
    pws_src src[4];
    int src_nchw;      /* first layer: x is the dense NCHW window */
    int cout;
    const float *gout; /* dy (already multiplied by act'), NHWC dense, pixel stride gout_ld */
    int gout_ld;
    float *dw_packed;  /* gradient in the FORWARD packed layout (pws_packed_weight_floats), ACCUMULATED into
                          (fp32 atomics: several pixel ranges, and stages 2/3 share weights) */
    int math;          /* PWS_MATH_BF16: x and dy rounded to bf16 at LDS staging, fp32 accumulation; dW stays fp32 */
    int store;         /* PWS_STORE_BF16: src[].ptr and gout hold bf16 elements (dw_packed stays fp32) */
    float *dbias;      /* optional: dbias[co] += sum over the pixels of gout[., co] (fp32 atomics), taken from the dy tiles the
                          weight-gradient kernel stages anyway -- for a gout that already is the gradient wrt the pre-activation
                          (pws_dst.act_y) this replaces the pws_act_bwd_bias pass */
    int deterministic; /* 1: no pixel split -- every element of dw_packed / dbias receives exactly one fp32 atomic add from this
                          launch, so repeated runs are bit-identical (slower: one workgroup per channel block) */
    /* Optional SECOND operand pair of the same geometry (n, h, w, channels, strides as src[] / gout; ABI version 4): stages 2 and 3 of the
     * generator run the same modules (lib/networks_cascading.py:178-214), so a shared layer's weight gradient is the sum of two
     * contributions -- given both, the ring kernels walk the two tensors' tiles in ONE launch (one prologue, one set of epilogue
     * atomics: 5-11 % less than two launches on the decoder shapes); kinds the ring does not cover run as two launches.
     * gout2 == NULL: none. */
    const void *src2_ptr[4];
    const float *gout2;
} pws_conv_bwd_weight_args;
int pws_conv2d_bwd_weight(const pws_conv_bwd_weight_args *args, pws_stream_t stream);

/* Inverse of pws_pack_conv_weight: packed gradient -> torch layout (OIHW / IOHW), overwriting dw_torch. */
int pws_unpack_conv_weight(const float *w_packed, float *w_torch, int kind, int cin, int cout, pws_stream_t stream);

/* theta = LReLU(W2 . LReLU(W1 . vec(x) + b1) + b2)   x: NHWC [n,2,2,c] (ld == c), theta: [n,6].
 * w_flat packed as PWS_CONV_K2S1P0 (c -> hidden), w_lin packed as PWS_CONV_K1 (hidden -> 6).
 * ws: pws_theta_head_ws_floats(n, c, hidden) floats of scratch (K-split partial sums). */
size_t pws_theta_head_ws_floats(int n, int c, int hidden);
int pws_theta_head_fwd(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat,
                       const float *w_lin, const float *b_lin, float *ws, float *theta, pws_stream_t stream);
/* Same, additionally saving the hidden activations h[n,hidden] (post-LeakyReLU) for pws_theta_head_bwd. */
int pws_theta_head_fwd_save(const float *x, int n, int c, int hidden, const float *w_flat, const float *b_flat,
                            const float *w_lin, const float *b_lin, float *ws, float *theta, float *h_saved,
                            pws_stream_t stream);
/* Backward of the theta head.  dtheta[n,6] in; dw_flat (packed K2S1P0 layout [4c][hidden]), db_flat, dw_lin
 * ([hidden][6]), db_lin ACCUMULATED; dx (nullable) [n,2,2,c] overwritten or accumulated; ws: n*hidden floats. */
int pws_theta_head_bwd(const float *x, int n, int c, int hidden, const float *w_flat, const float *w_lin,
                       const float *h_saved, const float *theta, const float *dtheta, float *dw_flat, float *db_flat,
                       float *dw_lin, float *db_lin, float *dx, int dx_accumulate, float *ws, pws_stream_t stream);

/* field = tanh(tanh(conv3x3(x; 2 outputs) + b)) as N,H,W,2, plus affine_grid(theta) when theta != NULL.
 * x: NHWC [n,h,w,c] pixel stride ld.  w_out packed as PWS_CONV_K3S1_OUT.  resid (nullable) receives the
 * field without the affine part; grid (nullable) receives resid + affine. */
int pws_field_head_fwd(const float *x, int ld, int n, int h, int w, int c, const float *w_out,
                       const float *b_out, const float *theta, int align_corners, float *resid, float *grid,
                       pws_stream_t stream);

int pws_field_head_fwd_s(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *b_out,
                         const float *theta, int align_corners, float *resid, float *grid, int store, pws_stream_t stream);

/* Backward of pws_field_head_fwd.  resid: the forward's residual output; g_grid / g_resid (either nullable): gradients
 * wrt the two outputs.  dx (nullable) [n,h,w,c] overwritten or accumulated; dw_out (packed [9][c][2]) and db_out[2]
 * ACCUMULATED (atomics); dtheta (nullable) [n,6] overwritten; ws: n*h*w*2 floats. */
int pws_field_head_bwd(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                       const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld,
                       int dx_accumulate, float *dw_out, float *db_out, float *dtheta, float *ws, pws_stream_t stream);

int pws_field_head_bwd_s(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                         const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld, int dx_accumulate,
                         float *dw_out, float *db_out, float *dtheta, float *ws, int store, pws_stream_t stream);

/* The same; dx_act = PWS_ACT_LRELU / PWS_ACT_RELU (bf16 storage only): this call completes the gradient of x, which is the
 * output of that activation -- dx (after the accumulation) is multiplied by act'(x), as pws_dst.act_y does for the
 * data-gradient calls.  PWS_ACT_NONE: pws_field_head_bwd_s. */
int pws_field_head_bwd_act(const float *x, int ld, int n, int h, int w, int c, const float *w_out, const float *resid,
                           const float *g_grid, const float *g_resid, int align_corners, float *dx, int dx_ld, int dx_accumulate,
                           float *dw_out, float *db_out, float *dtheta, float *ws, int store, int dx_act, pws_stream_t stream);

/* F.affine_grid(theta[n,2,3], (n,*,h,w)) -> grid[n,h,w,2] */
int pws_affine_grid(const float *theta, float *grid, int n, int h, int w, int align_corners,
                    pws_stream_t stream);

/* F.grid_sample(input[n,c,h,w], grid[n,ho,wo,2]) -> out[n,c,ho,wo]; bilinear, zeros padding. */
int pws_grid_sample_fwd(const float *input, const float *grid, float *out, int n, int c, int h, int w, int ho,
                        int wo, int align_corners, pws_stream_t stream);
/* Backward.  ginput (nullable): [n,c,h,w], OVERWRITTEN (zeroed then scatter-added with atomics);
 * ggrid (nullable): [n,ho,wo,2], overwritten. */
int pws_grid_sample_bwd(const float *gout, const float *input, const float *grid, float *ginput, float *ggrid,
                        int n, int c, int h, int w, int ho, int wo, int align_corners, pws_stream_t stream);

/* Bilinear resize with align_corners=True of an NCHW tensor (reference resizes the field as NCHW). */
int pws_upsample_bilinear_ac(const float *in, float *out, int n, int c, int h, int w, int ho, int wo,
                             pws_stream_t stream);
/* Adjoint of pws_upsample_bilinear_ac: gin[n,c,h,w] (overwritten) from gout[n,c,ho,wo]; deterministic gather.  autograd of
 * torch.nn.UpsamplingBilinear2d when the reference's video loop runs with gradients enabled (main_new.py:697-710). */
int pws_upsample_bilinear_ac_bwd(const float *gout, float *gin, int n, int c, int h, int w, int ho, int wo,
                                 pws_stream_t stream);
/* Adjoint of pws_affine_grid: gtheta[n,6] (overwritten) = sum over h,w of ggrid[n,h,w,:] (x) [x_w, y_h, 1]; autograd of
 * F.affine_grid (lib/networks_cascading.py:164,188,210; main_new.py:195) for callers outside the fused field head. */
int pws_affine_grid_bwd(const float *ggrid, float *gtheta, int n, int h, int w, int align_corners, pws_stream_t stream);

/* Fused 720p path: field[n,fh,fw,2] is resized on the fly (align_corners=True) to (h,w) and applied to
 * input[n,c,h,w] -> out[n,c,h,w]; the resized field is never materialised. */
int pws_upsample_grid_sample_fwd(const float *input, const float *field, float *out, int n, int c, int h,
                                 int w, int fh, int fw, int align_corners, pws_stream_t stream);

/* The same for uint8 frames in the layout OpenCV hands over (main_new.py:679-684,717-721): frame_hwc / out_hwc are
 * [n,h,w,3] uint8; the blend runs in fp32 exactly as above and is truncated like numpy's astype(uint8).  swap_rb: output
 * channel c samples input channel 2-c (the reference's COLOR_BGR2RGB before the warp).  w % 4 == 0, out 4-byte aligned.
 * 6 bytes per pixel of frame traffic instead of 24. */
int pws_upsample_grid_sample_u8(const unsigned char *frame_hwc, const float *field, unsigned char *out_hwc, int n, int h, int w,
                                int fh, int fw, int swap_rb, int align_corners, pws_stream_t stream);

/* The image processing either side of the path in the reference's video loop, on the device (csrc/frameio.hip; follows
 * OpenCV 4.x's algorithms -- cv2 is not available to pin against):
 * out[n,oh,ow] = cv2.resize(cv2.cvtColor(frame, COLOR_BGR2GRAY), (ow, oh), INTER_AREA) [/255*2-1 when normalize]
 * (main_new.py:639-643,653-667); frames [n,h,w,3] uint8 as cv2 delivers them (swap_rb: they are RGB instead). */
int pws_gray_area_u8(const unsigned char *frames_hwc, float *out, int n, int h, int w, int oh, int ow, int normalize, int swap_rb,
                     pws_stream_t stream);
/* out[n,h/2,w/2,3] = cv2.resize(in, (w/2, h/2), INTER_AREA) [+ R<->B swap] (main_new.py:723-725) */
int pws_area_half_u8(const unsigned char *in_hwc, unsigned char *out_hwc, int n, int h, int w, int swap_rb, pws_stream_t stream);
/* out[n,oh,ow,3] = cv2.resize(in[n,h,w,3], (ow, oh), INTER_AREA) [+ R<->B swap] for ANY down-scaling ratio: the reference writes
 * every output frame at (640, 360) whatever the source size (main_new.py:723).  2 x 2 -> pws_area_half_u8; both ratios integer
 * (1080p, 2160p) -> OpenCV's integer-area path (int sum * float(1/area), round half to even); otherwise its area tables. */
int pws_area_resize_u8(const unsigned char *in_hwc, unsigned char *out_hwc, int n, int h, int w, int oh, int ow, int swap_rb,
                       pws_stream_t stream);

/* Adam (no weight decay / amsgrad) on a flat fp32 buffer, in place; step counts from 1. */
int pws_adam_step(float *p, const float *g, float *m, float *v, size_t count, float lr, float beta1,
                  float beta2, float eps, int step, pws_stream_t stream);

/* The same update for `ntensors` tensors in one launch per 48 tensors (p, g, m, v, counts: HOST arrays of DEVICE pointers /
 * element counts): a generator has 92 tensors, most of them small. */
int pws_adam_step_multi(float *const *p, const float *const *g, float *const *m, float *const *v, const size_t *counts,
                        int ntensors, float lr, float beta1, float beta2, float eps, int step, pws_stream_t stream);

/* ---------------------------------------------------------------- training objective around the path
 * The per-step losses the reference's train() wraps round netG + grid_sample (main_new.py:101-118,184-212;
 * lib/utils.py:246-255,339-362,405-447), generator part, no GAN; the VGG perceptual term stays with the caller (it gets
 * the warped frames and hands back a gradient through `gextra`).  All tensors carry m samples (the reference's two
 * forwards per item pair batched: branch 1 = samples [0,n), branch 2 = [n,2n)), fields [m,h,w,2], frames [m,3,h,w].
 * Sums are accumulated into caller-zeroed DOUBLE slot arrays of PWS_OBJ_SLOTS entries per quantity (one f64 atomic per
 * workgroup, spread over the slots); pws_objective_finalize reduces them. */
#define PWS_OBJ_SLOTS 64
/* images.float()*(1/255)*2-1 (lib/utils.py:247) for `per_sample` contiguous bytes of each of m samples; strides in elements */
int pws_u8_normalize(const unsigned char *src, size_t src_nstride, float *dst, size_t dst_nstride, int m, size_t per_sample,
                     pws_stream_t stream);
/* fake = grid_sample((src+1)*127.5, grid)/127.5 - 1 (main_new.py:106-107); src: 3 planes per sample, sample i at
 * src + i*src_nstride (a view of image_unstable[:, period+1 : period+4]).  target/l1_slots (both or neither):
 * l1_slots += sum|target - fake| (lib/utils.py:349), target sample i at target + i*tgt_nstride. */
int pws_warp_norm_fwd(const float *src, size_t src_nstride, const float *grid, float *fake, const float *target,
                      size_t tgt_nstride, double *l1_slots, int m, int h, int w, pws_stream_t stream);
/* `scale` in the backward entry points: nullable DEVICE pointer to one float that multiplies the host coefficient (the
 * upstream gradient of the loss, which autograd holds on the device: reading it there avoids a host synchronisation).
 * ggrid (=, or += when accumulate) (d fake/d grid)^T (c_l1*scale*sign(fake - target) + gextra); target, gextra nullable */
int pws_warp_norm_bwd(const float *src, size_t src_nstride, const float *grid, const float *target, size_t tgt_nstride,
                      float c_l1, const float *scale, const float *gextra, float *ggrid, int accumulate, int m, int h, int w,
                      pws_stream_t stream);
/* slots += sum|grid_sample(fake2, affine_grid(theta[n,6])) - fake1| (main_new.py:195-198); fake1, fake2: [n,3,h,w] */
int pws_temporal_l1_fwd(const float *fake1, const float *fake2, const float *theta, double *slots, int n, int h, int w,
                        pws_stream_t stream);
/* gfake1 -= c*sign(d) (plain read-modify-write), gfake2 += c*sign(d)*bilinear weights (atomics): both must be initialised */
int pws_temporal_l1_bwd(const float *fake1, const float *fake2, const float *theta, float c, const float *scale, float *gfake1,
                        float *gfake2, int n, int h, int w, pws_stream_t stream);
/* The same gradients without atomics: the scatter into gfake2 runs as an ordered gather (one lane per source pixel walks the output
 * pixels whose taps can hit it), so repeated runs are bit-identical.  scratch: n * 3 * h * w floats.  Slower (deterministic mode). */
int pws_temporal_l1_bwd_det(const float *fake1, const float *fake2, const float *theta, float c, const float *scale, float *gfake1,
                            float *gfake2, float *scratch, int n, int h, int w, pws_stream_t stream);
/* features [m,nf,6] = [stable x,y,1, unstable x,y,1] (lib/utils.py:225); slots += sum_k |unstable_k - grid[stable_k]|^2
 * with the reference's index int((coord+1)*size/2) (lib/utils.py:341-345) */
int pws_feature_loss_fwd(const float *grid, const float *features, double *slots, int m, int nf, int h, int w,
                         pws_stream_t stream);
/* ggrid += c * d(sum)/d grid (atomics; ggrid must be initialised) */
int pws_feature_loss_bwd(const float *grid, const float *features, float c, const float *scale, float *ggrid, int m, int nf,
                         int h, int w, pws_stream_t stream);
/* The same without atomics: one lane per sample adds its points in order (bit-identical runs; deterministic mode). */
int pws_feature_loss_bwd_det(const float *grid, const float *features, float c, const float *scale, float *ggrid, int m, int nf,
                             int h, int w, pws_stream_t stream);
/* slots_dx += sum|grid[:,:,1:]-grid[:,:,:-1]|, slots_dy likewise along h (lib/utils.py:351-357; reported, not optimised) */
int pws_field_smoothness(const float *grid, double *slots_dx, double *slots_dy, int m, int h, int w, pws_stream_t stream);
/* loss_pixel1 (lib/utils.py:405-425): fp64 L1 residual of the least-squares fit of generate_affine_matrix's bilinear
 * corner basis to every (size/block)^2-pixel block of resid [m,size,size,2]; requires size == block*block as the
 * reference's basis does.  bwd: gresid = c * d(sum)/d resid, overwritten. */
int pws_shape_loss_fwd(const float *resid, double *slots, int m, int size, int block, pws_stream_t stream);
int pws_shape_loss_bwd(const float *resid, double c, const float *scale, float *gresid, int m, int size, int block,
                       pws_stream_t stream);
/* out[j] = sum_q coef[j*nq+q] * (sum of the PWS_OBJ_SLOTS slots of quantity q), j < nout <= 64; slots [nq][PWS_OBJ_SLOTS],
 * coef [nout][nq] device doubles */
int pws_objective_finalize(const double *slots, int nq, const double *coef, int nout, float *out, pws_stream_t stream);

/* ---- VGG-16 perceptual term (lib/utils.py:11-32): the 13 conv3x3+ReLU layers run on pws_conv2d_fwd /
 * pws_conv2d_bwd_data (frozen weights: no weight gradient); these are the remaining pieces.  NHWC fp32. */
/* nn.MaxPool2d(2, 2): y[n,h/2,w/2,c]; h, w even, c % 4 == 0 */
int pws_maxpool2x2_fwd(const float *x, float *y, int n, int h, int w, int c, pws_stream_t stream);
/* dx (overwritten) = dy routed to the first maximum of each window in row-major scan order (ATen's argmax), 0 elsewhere */
int pws_maxpool2x2_bwd(const float *x, const float *dy, float *dx, int n, int h, int w, int c, pws_stream_t stream);
/* the same with the element type of x / y / dy / dx given by `store` (PWS_STORE_BF16: bf16 elements, c % 8 == 0) */
int pws_maxpool2x2_fwd_s(const void *x, void *y, int n, int h, int w, int c, int store, pws_stream_t stream);
/* relu_mask != 0 (bf16 storage only): x is a ReLU output and dx is the gradient wrt the pre-activation, dx *= (x > 0) */
int pws_maxpool2x2_bwd_s(const void *x, const void *dy, void *dx, int n, int h, int w, int c, int store, int relu_mask,
                         pws_stream_t stream);
/* nn.MSELoss pieces: slots[PWS_OBJ_SLOTS] (caller-zeroed doubles) += sum (a-b)^2 ;  ga = c * (*scale) * 2 (a-b) */
int pws_sqdiff_sum(const float *a, const float *b, size_t count, double *slots, pws_stream_t stream);
int pws_sqdiff_bwd(const float *a, const float *b, size_t count, float c, const float *scale, float *ga, pws_stream_t stream);

/* ---- training-mode BatchNorm2d + activation (the use_BN variant, lib/networks_cascading.py:253-341: conv -> BatchNorm2d ->
 * LeakyReLU / ReLU in every block), NHWC fp32 [pixels][c], any c.  ws: pws_bn_ws_bytes(c) bytes of scratch.
 * fwd: stats[2c] = (batch mean, 1/sqrt(biased var + eps)) saved for backward; y = act(gamma * xhat + beta); running_mean /
 *      running_var (nullable pair) updated `repeat` times with momentum and the UNBIASED variance, as torch does per call.
 * bwd: dy is overwritten by dz; dgamma / dbeta (nullable) are ACCUMULATED (a module used by several stages adds up).
 * pixels == 1 is refused like torch ("Expected more than 1 value per channel when training"). */
size_t pws_bn_ws_bytes(int c);
int pws_bn_train_fwd(const float *z, size_t pixels, int c, const float *gamma, const float *beta, int act, float *y, float *stats,
                     float *running_mean, float *running_var, float momentum, float eps, int repeat, void *ws, size_t ws_bytes,
                     pws_stream_t stream);
int pws_bn_train_bwd(float *dy, const float *y, const float *z, const float *stats, const float *gamma, int act, size_t pixels, int c,
                     float *dgamma, float *dbeta, void *ws, size_t ws_bytes, pws_stream_t stream);

/* ---------------------------------------------------------------- whole generator */
/* Floats needed for all 46 packed layer weights + 46 biases of a generator (input_nc, ngf). */
size_t pws_netg_packed_floats(int input_nc, int ngf);
/* params: HOST array of 92 DEVICE pointers in state-dict order (weight, bias per layer; torch layouts). */
int pws_netg_pack_weights(const float *const *params, float *packed, int input_nc, int ngf,
                          pws_stream_t stream);
/* The same for a buffer that will serve forwards of ONE math mode only (a training loop re-packs after every optimizer step): `math`
 * PWS_MATH_BF16 leaves out the Winograd-domain copies of the layers that have bf16 weights (those layers never read them in that mode),
 * PWS_MATH_FP32 leaves out the bf16 copies; anything else packs everything (= pws_netg_pack_weights).  Running a forward of the other
 * mode on such a buffer reads stale copies: re-pack when the mode changes. */
int pws_netg_pack_weights_for(const float *const *params, float *packed, int input_nc, int ngf, int math,
                              pws_stream_t stream);
size_t pws_netg_workspace_bytes(int n, int input_nc, int ngf, int is_training);
/* x: [n,input_nc,256,256] NCHW.  grids: is_training ? [3][n,256,256,2] : [n,256,256,2] (stage 3).
 * resid: is_training ? [3][n,256,256,2] : ignored (may be NULL).  thetas (nullable): [3][n,6]. */
int pws_netg_forward(const float *packed, const float *x, int n, int input_nc, int ngf, int is_training,
                     int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid, float *thetas,
                     pws_stream_t stream);

/* The same forward with its mode in the ARGUMENTS instead of the process-wide pws_set_option() defaults: two generators with
 * different arithmetic may run from two host threads (or on two devices) at once.  opts == NULL: as pws_netg_forward. */
typedef struct pws_netg_opts {
    int math;       /* PWS_MATH_FP32 / PWS_MATH_BF16 */
    int store;      /* PWS_STORE_FP32 / PWS_STORE_BF16 (needs PWS_MATH_BF16 and ngf % 32 == 0) */
    int two_queues; /* 1: fork the internal second queue (forward: the stage-2/3 encoder -- stages 2 and 3 run in lockstep, their shared
                       layers as one launch of batch 2n -- and the deep decoder levels beside stage 1's encoder / decoder;
                       backward: every weight gradient beside the data-gradient chain; the queues join before the
                       call's last launch on the caller's stream / at the end of the call); 0: caller's stream only;
                       -1: the process default (PWS_OPT_TWO_QUEUES) */
    int flags;      /* bit set of PWS_NETG_*; unknown bits are refused */
    size_t x_sample_stride; /* floats between consecutive samples of the window tensor x; 0 = dense (input_nc * 256 * 256).  The
                               windows of a video are OVERLAPPING views of one plane buffer (window b = planes b .. b+30,
                               main_new.py:627-673): stride 256 * 256 reads them in place instead of from a gathered copy.
                               Inference forward only (the backward reads x dense). */
} pws_netg_opts;
#define PWS_NETG_DETERMINISTIC 1   /* backward: every weight / bias gradient element receives exactly ONE fp32 atomic add per
                                      launch (no pixel split across workgroups), so two runs give bit-identical gradients; slower */
#define PWS_NETG_PRUNE_DEAD 2      /* inference forward (is_training = 0) only: do not compute stage 1's `up2` (x122, reference
                                      lib/networks_cascading.py:171): its only consumers are `up1` (:173) and stage 2's `up_bottom1`
                                      (:196), both under `if is_training` -- the returned field is bit-identical, 2.147 of the 94.48 GFLOP
                                      per frame are not spent.  Off by default: the reference executes the layer. */
int pws_netg_forward_opts(const float *packed, const float *x, int n, int input_nc, int ngf, int is_training, int align_corners,
                          void *ws, size_t ws_bytes, float *grids, float *resid, float *thetas, const pws_netg_opts *opts,
                          pws_stream_t stream);

/* ---- training: backward of the whole generator (is_training=1 forward must have run on the SAME ws, untouched since).
 * Data-gradient weights: a second packed buffer (pws_netg_packed_dgrad_floats) filled by pws_netg_pack_weights_dgrad.
 * g_grids / g_resid: [3][n,256,256,2] each, or NULL as a whole (no gradient wrt that output list).
 * dpacked: the GRADIENT SLAB (pws_netg_grad_floats floats, overwritten): per layer in state-dict order the weight gradient in the
 * forward packed layout with the bias gradient behind it and nothing else (no Winograd / bf16 copies as in `packed`), so that a
 * data-parallel host all-reduces the slab -- or the ranges pws_netg_grad_layout reports for the layers a run finished -- IN PLACE
 * (replaces nn.DataParallel's reduce to GPU 0, lib/networks_cascading.py:51-52) and unpacks once;
 * pws_netg_unpack_grads converts it to 92 torch-layout tensors.  dx (nullable): gradient wrt the input window is not
 * needed by the reference (the window is data) and is not computed. */
size_t pws_netg_packed_dgrad_floats(int input_nc, int ngf);
size_t pws_netg_grad_floats(int input_nc, int ngf);
/* first_float[46], floats[46] (HOST): the contiguous range of the gradient slab that holds layer i's weight + bias gradient (with
 * its alignment padding, which stays zero); consecutive layers abut, so the layers a backward run reports final merge into a few
 * large ranges. */
int pws_netg_grad_layout(int input_nc, int ngf, size_t *first_float, size_t *floats);
/* final_part[46] (HOST; no GPU call): the run of an nparts-run backward (pws_netg_backward_opts / _lists with part = 0 .. nparts-1)
 * after which layer i's range of the slab is final, i.e. the run whose final_mask first reports it -- lets a data-parallel host
 * plan its messages (and a CPU rehearsal replay them) without a device. */
int pws_netg_backward_plan(int input_nc, int ngf, int nparts, unsigned char *final_part);
int pws_netg_pack_weights_dgrad(const float *const *params, float *packed_dgrad, int input_nc, int ngf,
                                pws_stream_t stream);
/* Both buffers of a training step at once (ABI 5).  math == PWS_MATH_BF16 and ngf % 64 == 0: the conv layers go torch layout -> bf16
 * forward copy + bf16 data-gradient copy in ONE pass, without the fp32 packed copies nothing reads in that mode (so the buffers then
 * serve bf16-math calls only, as pws_netg_pack_weights_for(.., PWS_MATH_BF16, ..) already implies); otherwise the two calls above. */
int pws_netg_pack_weights_train(const float *const *params, float *packed, float *packed_dgrad, int input_nc, int ngf, int math,
                                pws_stream_t stream);
size_t pws_netg_train_workspace_bytes(int n, int input_nc, int ngf);
int pws_netg_backward(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                      int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                      const float *g_grids, const float *g_resid, float *dpacked, pws_stream_t stream);
/* The same backward cut into `nparts` consecutive runs of the reversed layer tape; call part = 0 .. nparts-1 in order on the
 * same arguments.  final_mask (nullable, HOST, 46 bytes, one per layer in state-dict order): set to 1 for every layer whose
 * weight / bias gradient in `dpacked` is complete once this run has executed -- a data-parallel host unpacks and all-reduces
 * those on a second stream while the next run computes (gradient exchange overlapped with backward). */
int pws_netg_backward_part(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                           int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                           const float *g_grids, const float *g_resid, float *dpacked, int part, int nparts,
                           unsigned char *final_mask, pws_stream_t stream);
/* pws_netg_backward / pws_netg_backward_part with the mode of the forward that filled `ws` passed explicitly (opts must equal the
 * forward's; NULL: the process defaults).  part = 0, nparts = 1, final_mask = NULL is the whole backward. */
int pws_netg_backward_opts(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                           int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                           const float *g_grids, const float *g_resid, float *dpacked, int part, int nparts,
                           unsigned char *final_mask, const pws_netg_opts *opts, pws_stream_t stream);
/* pws_netg_backward_opts with the six upstream gradients as two HOST arrays of 3 DEVICE pointers ([n,256,256,2] each; an entry, or a
 * whole array, may be NULL: that output has no gradient) -- autograd hands them over as separate tensors (main_new.py:214), so no
 * stacked copy and no zero-filled stand-ins are needed. */
int pws_netg_backward_lists(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                            int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                            const float *const *g_grids, const float *const *g_resid, float *dpacked, int part, int nparts,
                            unsigned char *final_mask, const pws_netg_opts *opts, pws_stream_t stream);
/* ---- use_BN=True training (lib/cfg.py:37; lib/networks_cascading.py:253-341: BatchNorm2d after every conv, batch statistics).
 * bn_params / bn_running / dbn: flat buffers of pws_netg_bn_floats() floats, per layer in state-dict order
 * [gamma(cout) | beta(cout)], [running_mean | running_var], [dgamma | dbeta].  The forward is the is_training one (6 fields) and
 * updates bn_running (nullable) with `momentum` once per call of a module, as torch does (down_bottom1, which the reference
 * evaluates twice on the same input, is computed once and updated twice); n >= 2 (the theta head has one value per sample and
 * channel).  `packed` holds the raw (un-folded) conv weights.  fp32 math / storage only; arena:
 * pws_netg_train_workspace_bytes_bn.  dpacked's bias entries stay zero: a bias in front of a BatchNorm has no gradient. */
size_t pws_netg_bn_floats(int input_nc, int ngf);
size_t pws_netg_train_workspace_bytes_bn(int n, int input_nc, int ngf);
int pws_netg_forward_bn(const float *packed, const float *bn_params, float *bn_running, float momentum, float eps, const float *x,
                        int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid,
                        float *thetas, pws_stream_t stream);
int pws_netg_backward_bn(const float *packed, const float *packed_dgrad, const float *bn_params, float eps, const float *x, int n,
                         int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                         const float *g_grids, const float *g_resid, float *dpacked, float *dbn, pws_stream_t stream);
/* The same two with their mode in the arguments: opts->math = PWS_MATH_BF16 runs the conv contractions (forward, data and weight
 * gradients) on the bf16 matrix cores -- operands rounded while they are staged, fp32 accumulation; activations, BatchNorm
 * statistics and gradients stay fp32 (opts->store must be PWS_STORE_FP32).  opts == NULL: fp32. */
int pws_netg_forward_bn_opts(const float *packed, const float *bn_params, float *bn_running, float momentum, float eps, const float *x,
                             int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid,
                             float *thetas, const pws_netg_opts *opts, pws_stream_t stream);
int pws_netg_backward_bn_opts(const float *packed, const float *packed_dgrad, const float *bn_params, float eps, const float *x, int n,
                              int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, const float *resid,
                              const float *thetas, const float *g_grids, const float *g_resid, float *dpacked, float *dbn,
                              const pws_netg_opts *opts, pws_stream_t stream);
/* grads: HOST array of 92 DEVICE pointers (torch layouts, state-dict order), overwritten; a layer whose weight AND bias
 * pointers are NULL is skipped. */
int pws_netg_unpack_grads(const float *dpacked, float *const *grads, int input_nc, int ngf, pws_stream_t stream);

/* ---------------------------------------------------------------- measurement hooks (bench / tests only)
 * When enabled, every kernel launch of this library is bracketed by two hipEvents on the launch stream and
 * tagged with its algorithmic work.  Process-global and not thread-safe: a measurement facility, off by
 * default, never enabled by the product path. */
typedef struct pws_prof_record {
    int kernel_id;  /* index for pws_prof_kernel_name() */
    int tag;        /* caller-defined (the generator executor passes the layer index) */
    double flops;   /* algorithmic floating-point operations of this launch */
    double bytes;   /* algorithmic HBM bytes of this launch (each tensor touched once) */
    float ms;       /* hipEventElapsedTime between the two events */
} pws_prof_record;
int pws_prof_enable(int on);
/* Synchronises the recorded events, copies up to max_records records (launch order), clears the log,
 * returns the number of records that were pending (may exceed max_records) or a negative error. */
int pws_prof_collect(pws_prof_record *out, int max_records);
const char *pws_prof_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* PWSTABLE_H */

// Host-side executor of the cascading generator: sequences the HIP kernels of this library on one stream,
// with all activations in a caller-provided arena.  Mirrors UnetGenerator.forward of the reference
// (lib/networks_cascading.py:152-237) -- variable names below are the reference's -- and its autograd backward
// (loss_g.backward(), main_new.py:214).
//
// Differences in HOW (not in WHAT):
//   * activations are NHWC; a torch.cat([a, b], 1) is a list of (pointer, channels, stride) sources consumed
//     directly by the next convolution -- the reference's 43-44 concat copies per forward disappear;
//   * stage 3's first block (`x32 = down_bottom1(None, x11)`, reference :200) has the same weights and the same
//     input as stage 2's (`x22`, :178) and is bit-identical, so it is computed once (its gradient is the sum of
//     both uses, which falls out of x22 having one gradient buffer);
//   * `out` conv + tanh + tanh + permute + affine_grid + add run as one kernel (field head).
// Backward replays the forward's allocation sequence (no launches) to recover every activation's address in the
// arena, then walks the recorded ops in reverse: per layer act'/bias-grad, weight grad, data grad into the gradient
// buffers of the layer's sources (first consumer overwrites, later consumers accumulate -- no zero-fill pass).
// No allocation, no synchronisation: every launch is ordered after the caller's `stream`; the forward forks an internal
// second queue by events (stage k+1's encoder beside stage k's decoder) and joins it back before returning.
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"
#include <cstdlib>
#include <cstdio>
#include "netg_pack.h"

namespace pws {

struct Layer {
    int kind, cin, cout;
    size_t w_off, b_off;  // float offsets into the packed buffer
    size_t gw_off, gb_off;  // float offsets into the GRADIENT slab (pws_netg_grad_floats): weight gradient in the forward packed layout,
                            // bias gradient right behind it -- the slab holds nothing else, so a data-parallel host all-reduces it in place
    size_t dg_off;        // float offset into the data-gradient weight buffer (or SIZE_MAX)
    size_t ww_off;        // float offset of the Winograd-domain weights inside the packed buffer (or SIZE_MAX)
    size_t wr_off;        // ... of the ring-layout Winograd weights (conv_wring.hip; or SIZE_MAX)
    size_t wb_off;        // float offset of the bf16 weights inside the packed buffer (or SIZE_MAX)
    size_t dgb_off;       // float offset of the bf16 data-gradient weights inside the data-gradient buffer (or SIZE_MAX)
};

// planes of the forward / data-gradient packed layouts of the kinds the bf16 kernels cover (0: not covered)
static int bf16_planes(int kind) {
    switch (kind) {
    case PWS_CONV_K3S1:
    case PWS_CONV_K3S2:
    case PWS_CONVT_K3S1: return 9;
    case PWS_CONVT_K4S2: return 16;
    case PWS_CONV_K5S1: return 25;
    default: return 0;
    }
}
static int bf16_dgrad_planes(int kind) { return kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1 ? 9 : 16; }

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
static std::vector<Layer> build_layers(int input_nc, int g, size_t *total_floats, size_t *total_dgrad = nullptr, size_t *total_grad = nullptr) {
    std::vector<Layer> L;
    size_t off = 0, dg = 0, gr = 0;
    // ring: the layer runs on maps of whole 16 x 32 (or, two samples at a time, 16 x 16) units at the reference's 256 x 256 input (conv_wring.hip)
    auto add = [&](int kind, int cin, int cout, bool ring = false) {
        Layer l{kind, cin, cout, 0, 0, 0, 0, (size_t)-1, (size_t)-1, (size_t)-1, (size_t)-1, (size_t)-1};
        l.w_off = off;
        off = align_up(off + pws_packed_weight_floats(kind, cin, cout), 64);
        l.b_off = off;
        off = align_up(off + (size_t)cout, 64);
        l.gw_off = gr;
        gr = align_up(gr + pws_packed_weight_floats(kind, cin, cout), 64);
        l.gb_off = gr;
        gr = align_up(gr + (size_t)cout, 64);
        if (kind == PWS_CONV_K3S1 || kind == PWS_CONVT_K3S1) {
            l.ww_off = off;
            off = align_up(off + pws_packed_wino_floats(cin, cout), 64);
        }
        if (ring && pws_packed_wring_floats(kind, cin, cout)) {
            l.wr_off = off;
            off = align_up(off + pws_packed_wring_floats(kind, cin, cout), 64);
        }
        // (CONVT_K4S2 has a Winograd F(3x3,2x2) kernel too, conv_wino.hip MODE 1; measured slower than the direct kernel on
        //  this generator's 32..128-pixel maps, so its weights are not packed here)
        if (bf16_planes(kind) && ((cin + 15) / 16 * 16) % 32 == 0) {
            l.wb_off = off;
            off = align_up(off + pws_packed_bf16_floats(bf16_planes(kind), (cin + 15) / 16 * 16, cout), 64);
        }
        const size_t d = pws_packed_dgrad_floats(kind, cin, cout);
        if (d) l.dg_off = dg, dg = align_up(dg + d, 64);
        if (d && bf16_planes(kind) && cout % 32 == 0) {
            l.dgb_off = dg;
            dg = align_up(dg + pws_packed_bf16_floats(bf16_dgrad_planes(kind), cout, cin), 64);
        }
        L.push_back(l);
    };
    const int enc[7][2] = {{g, g}, {g, 2 * g}, {2 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}, {4 * g, 4 * g}};
    add(PWS_CONV_K5S1, input_nc, g, true);   // (ring: the F(2x2,5x5) weights of conv_first_wino.hip, where pws_packed_wring_floats covers the shape)
    for (int i = 0; i < 7; ++i) add(PWS_CONV_K3S2, enc[i][0], enc[i][1]);
    const int dec[7][2] = {{4 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 4 * g}, {8 * g, 2 * g}, {4 * g, g}, {2 * g, g}};
    for (int j = 0; j < 7; ++j) add(PWS_CONVT_K4S2, dec[j][0], dec[j][1], j >= 3);  // up7..up1 (up4..up1: inputs of 16^2..128^2)
    add(PWS_CONV_K3S1_OUT, g, 2);
    for (int i = 0; i < 7; ++i) {
        add(PWS_CONV_K3S1, enc[i][0], enc[i][0], i < 4);   // down_bottom1..4.conv_same: maps of 256^2..32^2
        add(PWS_CONV_K3S2, i == 0 ? enc[i][0] : 2 * enc[i][0], enc[i][1]);
    }
    const int ub[7][3] = {{4 * g, 4 * g, 8 * g}, {8 * g, 4 * g, 16 * g}, {8 * g, 4 * g, 16 * g}, {8 * g, 4 * g, 16 * g},
                          {8 * g, 2 * g, 16 * g}, {4 * g, g, 8 * g}, {2 * g, g, 4 * g}};  // (input_nc, output_nc, inner_nc)
    for (int j = 0; j < 7; ++j) {
        add(PWS_CONVT_K4S2, ub[j][2], ub[j][1], j >= 3);   // up_bottom4..1: inputs of 16^2..128^2
        add(PWS_CONVT_K3S1, ub[j][0], ub[j][0], j >= 3);
    }
    add(PWS_CONV_K2S1P0, 4 * g, 8 * g);
    add(PWS_CONV_K1, 8 * g, 6);
    if (total_floats) *total_floats = off;
    if (total_dgrad) *total_dgrad = dg;
    if (total_grad) *total_grad = gr;
    return L;
}

static int kind_ksize(int kind) {
    switch (kind) {
    case PWS_CONV_K5S1: return 5;
    case PWS_CONVT_K4S2: return 4;
    case PWS_CONV_K2S1P0: return 2;
    case PWS_CONV_K1: return 1;
    default: return 3;
    }
}
static unsigned off32(size_t off) { return off == (size_t)-1 ? kNoOff : (unsigned)off; }

// argument blocks of the whole-generator pack / unpack kernels (netg_pack.hip)
static void fill_pack_layers(const std::vector<Layer> &L, PackLayer *out) {
    for (int i = 0; i < L_COUNT; ++i) {
        const Layer &l = L[i];
        PackLayer &p = out[i];
        p.kind = l.kind, p.cin = l.cin, p.cin_pad = (l.cin + 15) / 16 * 16, p.cout = l.cout, p.k = kind_ksize(l.kind);
        p.planes = l.kind == PWS_CONVT_K4S2 ? 16 : p.k * p.k;
        p.dg_taps = l.dg_off == (size_t)-1 ? 0 : (l.kind == PWS_CONV_K3S1 || l.kind == PWS_CONVT_K3S1 ? 9 : 16);
        p.w_off = off32(l.w_off), p.b_off = off32(l.b_off), p.ww_off = off32(l.ww_off), p.dg_off = off32(l.dg_off), p.wr_off = off32(l.wr_off);
    }
}

// Mode of one forward / backward: carried by value from the entry point's arguments (pws_netg_opts) -- the executor reads no
// process-wide option, so two generators with different arithmetic can run from two host threads.
struct NetgOpts {
    int math, store;
    bool two_queues;
    bool deterministic = false;     // PWS_NETG_DETERMINISTIC
    size_t x_sample_stride = 0;     // floats between samples of the window (0 = dense)
    bool prune_dead = false;        // PWS_NETG_PRUNE_DEAD
};
static NetgOpts opts_defaults() { return NetgOpts{g_math, g_store, g_two_queues}; }
static int opts_from(const pws_netg_opts *o, NetgOpts *out) {
    if (!o) {
        *out = opts_defaults();
        return PWS_OK;
    }
    PWS_REQUIRE(o->math == PWS_MATH_FP32 || o->math == PWS_MATH_BF16, "pws_netg_opts: math %d", o->math);
    PWS_REQUIRE(o->store == PWS_STORE_FP32 || o->store == PWS_STORE_BF16, "pws_netg_opts: store %d", o->store);
    PWS_REQUIRE(o->store == PWS_STORE_FP32 || o->math == PWS_MATH_BF16, "pws_netg_opts: PWS_STORE_BF16 needs PWS_MATH_BF16");
    PWS_REQUIRE(o->two_queues >= -1 && o->two_queues <= 1 && (o->flags & ~(PWS_NETG_DETERMINISTIC | PWS_NETG_PRUNE_DEAD)) == 0, "pws_netg_opts: two_queues %d / flags %d",
                o->two_queues, o->flags);
    *out = NetgOpts{o->math, o->store, o->two_queues < 0 ? g_two_queues : o->two_queues != 0, (o->flags & PWS_NETG_DETERMINISTIC) != 0,
                    o->x_sample_stride, (o->flags & PWS_NETG_PRUNE_DEAD) != 0};
    return PWS_OK;
}

struct Seg {
    float *ptr;
    int c, ld;
    unsigned char *sign = nullptr;   // bf16 training: the tensor's sign bits (pws_conv_args.out_sign), c / 8 bytes per pixel; NULL = none
};
struct Tn {  // a (virtually concatenated) NHWC tensor
    Seg seg[4];
    int nseg, h, w;
};

enum { OP_CONV = 0, OP_THETA = 1, OP_FIELD = 2 };
struct Op {
    int type, layer, act, stage;
    bool nchw;
    Tn in, out;
    // use_BN training: pre-normalisation values and saved (mean, invstd) of the BatchNorms of this op
    //   conv : aux[0] = z, aux[1] = stats          field: aux[0] = z, aux[1] = stats, aux[2] = BatchNorm output
    //   theta: aux[0] = z1, aux[1] = stats1, aux[2] = z2, aux[3] = stats2
    float *aux[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Training-mode BatchNorm configuration of one forward / backward (use_BN=True, reference lib/networks_cascading.py:253-341).
// params / running: flat buffers, per layer in state-dict order [gamma(cout) | beta(cout)] resp. [running_mean | running_var].
struct BnCfg {
    const float *params = nullptr;
    float *running = nullptr;
    float momentum = 0.1f, eps = 1e-5f;
};

// Second in-order queue for the branch of the forward that does not depend on the current one (stage k+1's encoder
// runs beside stage k's decoder), plus the events that fork / join it.  Created lazily once per process (the only
// runtime objects this library owns); ordering against the caller's stream is by events only, never by a sync.
// The side QUEUE is one per device and process: the runtime deals streams to a few hardware queues round-robin, and a second side
// stream made late in a process with several streams alive (the backward's, on autograd's thread, in bench.py after its streaming
// leg) landed on the caller's hardware queue -- the two then serialise, barrier packets on top (configs[2] step 32.3 ms instead of
// 30.1; tools/_bin-style probe: 5 extra torch streams reproduce it).  So every host thread shares the stream made first; what a
// thread owns is its EVENT POOL (and a private side stream for calls made inside a graph capture: SideStream::pick).
static hipStream_t shared_side_queue(int dev) {
    static std::mutex m;
    static std::unordered_map<int, hipStream_t> per_device;
    std::lock_guard<std::mutex> lock(m);
    auto it = per_device.find(dev);
    if (it != per_device.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    per_device[dev] = s;
    return s;
}
struct SideStream {
    hipStream_t stream = nullptr;    // the device's shared side queue (eager calls)
    hipStream_t own = nullptr;       // this thread's private side stream: calls made inside a graph capture fork into it
    // Event pools, one per CALLER STREAM: a call re-records the pool's events from the first one on.  Waits made by an earlier call may
    // still be queued then -- harmless while the new record lands on the same queues as the old one (it is later in their order: the
    // queued wait can only become stricter).  With ONE pool per thread, a second call on ANOTHER stream re-recorded events that waits
    // of the first call had not consumed yet: autograd runs the backwards of all host threads on one worker thread per device, so two
    // training threads shared a pool, a weight-gradient launch of one waited for the other's stream and read its dy early (gradients
    // off by 1e-4..1e-1 of their size in 1 run of 10 of tests/test_hip_threads.py, tools/probes/thread_grad_probe.py, round 4).
    struct Pool {
        std::vector<hipEvent_t> events;
        size_t next = 0;
    };
    std::unordered_map<hipStream_t, Pool> pools;
    Pool *cur = nullptr;
    int dev = 0;
    void begin(hipStream_t st) { cur = &pools[st], cur->next = 0; }   // (references into an unordered_map survive a rehash)
    // The side queue to fork into beside `st`, or nullptr (run on one queue).  A call that is being CAPTURED never touches the shared
    // queue: forking joins the side stream to the caller's capture, and two host threads capturing at once would pull one stream into
    // two captures (both invalid) -- a captured call forks into the thread's own stream instead (a graph has no hardware queue of its
    // own to collide with: its nodes are scheduled when it is launched).  An eager call takes the shared queue, which therefore is never
    // part of anybody's capture; the capture status of `st` is the calling thread's own business, so nothing here is check-then-use
    // against another thread.
    hipStream_t pick(hipStream_t st) {
        hipStreamCaptureStatus cm = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cm) != hipSuccess) return nullptr;
        if (cm == hipStreamCaptureStatusNone) {
            if (!stream) stream = shared_side_queue(dev);
            // the private stream is made HERE, on the thread's first eager call: creating a stream while a capture is open is one of
            // the calls a capture may not survive (a host captures after an eager warm-up on the same thread)
            if (!own && hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) own = nullptr;
            return stream;
        }
        if (cm != hipStreamCaptureStatusActive) return nullptr;
        return own;   // nullptr (one queue) when this thread never made an eager call
    }
    hipEvent_t event() {
        if (!cur) return nullptr;
        if (cur->next == cur->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            cur->events.push_back(e);
        }
        return cur->events[cur->next++];
    }
};
// one event pool per host thread AND device
static SideStream &side_stream() {
    static thread_local std::unordered_map<int, SideStream> per_device;
    int dev = 0;
    (void)hipGetDevice(&dev);
    SideStream &s = per_device[dev];
    s.dev = dev;
    return s;
}

class Exec {
  public:
    Exec(const float *packed, const std::vector<Layer> &layers, int n, char *ws, size_t ws_bytes, hipStream_t st, bool dry,
         bool launch, const NetgOpts &o, const BnCfg *bn = nullptr)
        : math_(o.math), x_sstride_(o.x_sample_stride), prune_dead_(o.prune_dead), packed_(packed), L_(layers), n_(n), ws_(ws), cap_(ws_bytes), dry_(dry), launch_(launch && !dry) {
        if (bn) {
            bn_on_ = true, bn_ = *bn;
            size_t off = 0;
            for (int i = 0; i < L_COUNT; ++i) bn_off_[i] = off, off += 2 * (size_t)layers[i].cout;
            bn_total_ = off;
        }
        // bf16 storage of activations: only with bf16 math and when every conv layer is covered by the bf16 kernels
        io16_ = o.math == PWS_MATH_BF16 && o.store == PWS_STORE_BF16 && layers[L_DOWN1].cin % 32 == 0;
        streams_[0] = st, streams_[1] = st;
        if (launch_ && o.two_queues) {
            side_ = &side_stream();
            if (hipStream_t q = side_->pick(st)) streams_[1] = q, side_->begin(st);
        }
    }

    // ---- two-queue scheduling: ops go to queue `q`; fork/join are event record + wait (capturable, no host sync)
    void use(int q) { q_ = q; }
    bool two_queues() const { return streams_[1] != streams_[0]; }
    // mark(): "everything issued so far on queue `from`"; wait(): later work on queue `to` starts after that point
    hipEvent_t mark(int from) {
        if (!launch_ || !two_queues()) return nullptr;
        hipEvent_t e = side_->event();
        if (!e || hipEventRecord(e, streams_[from]) != hipSuccess) {
            if (rc_ == PWS_OK) set_error("pws_netg_forward: hipEventRecord failed"), rc_ = PWS_EHIP;
            return nullptr;
        }
        return e;
    }
    void wait(int to, hipEvent_t e) {
        if (!e) return;
        if (hipStreamWaitEvent(streams_[to], e, 0) != hipSuccess && rc_ == PWS_OK)
            set_error("pws_netg_forward: hipStreamWaitEvent failed"), rc_ = PWS_EHIP;
    }
    void order(int from, int to) { wait(to, mark(from)); }

    size_t used() const { return off_; }
    int rc() const { return rc_; }
    const std::vector<Op> &tape() const { return tape_; }
    float *splitk_ws() const { return splitk_ws_; }
    float *x_nhwc() const { return x_nhwc_; }
    bool io16() const { return io16_; }
    bool prune_dead() const { return prune_dead_; }
    int math() const { return math_; }
    int store() const { return io16_ ? PWS_STORE_BF16 : PWS_STORE_FP32; }
    float *theta_x32(int q) const { return theta_x32_[q]; }
    size_t splitk_bytes() const { return splitk_bytes_; }
    float *h_saved(int stage) const { return h_saved_[stage]; }
    bool bn_on() const { return bn_on_; }
    size_t bn_off(int layer) const { return bn_off_[layer]; }
    size_t bn_total() const { return bn_total_; }
    float *bn_ws(int q) const { return bn_ws_[q]; }
    size_t bn_ws_bytes() const { return bn_ws_bytes_; }
    const float *gamma(int layer) const { return bn_.params + bn_off_[layer]; }
    const float *beta(int layer) const { return bn_.params + bn_off_[layer] + L_[layer].cout; }
    // y = act(BatchNorm_train(z)) of `layer` on the current queue; stats saved, running statistics updated `repeat` times
    int bn_forward(int layer, const float *z, size_t pixels, int act, float *y, float *stats, int repeat) {
        const int c = L_[layer].cout;
        float *rm = bn_.running ? bn_.running + bn_off_[layer] : nullptr;
        return pws_bn_train_fwd(z, pixels, c, gamma(layer), beta(layer), act, y, stats, rm, rm ? rm + c : nullptr, bn_.momentum, bn_.eps,
                                repeat, bn_ws_[q_], bn_ws_bytes_, streams_[q_]);
    }

    float *alloc(size_t floats) {
        const size_t bytes = align_up(floats * sizeof(float), 256);
        const size_t at = off_;
        off_ += bytes;
        if (dry_) return nullptr;
        if (off_ > cap_) {
            if (rc_ == PWS_OK) {
                set_error("pws_netg: workspace too small (%zu B needed so far, %zu B given)", off_, cap_);
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

    static void out_extent(int kind, const Tn &x, int *oh, int *ow) {
        *oh = x.h, *ow = x.w;
        if (kind == PWS_CONV_K3S2) *oh = (x.h - 1) / 2 + 1, *ow = (x.w - 1) / 2 + 1;
        if (kind == PWS_CONVT_K4S2) *oh = 2 * x.h, *ow = 2 * x.w;
    }
    Tn conv(int layer, const Tn &x, int act, const float *nchw_src = nullptr, int nchw_c = 0) {
        const Layer &l = L_[layer];
        int oh, ow;
        out_extent(l.kind, x, &oh, &ow);
        Tn o{};
        o.nseg = 1, o.h = oh, o.w = ow;
        o.seg[0] = Seg{alloc((size_t)n_ * oh * ow * l.cout), l.cout, l.cout};
        // bf16 training: the layers whose backward multiplies by act'(this output) read its SIGN BITS instead of the tensor (1/16 of the
        // bytes: the data-gradient epilogues of the large maps are read-modify-writes waiting on their loads).  Asked of the maps the
        // ring kernel takes and the first layer (they write them in their epilogues; the one-shot kernel and split-K launches are followed
        // by a pass over the output) -- PWS_OPT_EXPERIMENT 12: never.
        // (The arena holds the bytes in every training layout -- the workspace size does not depend on the storage mode.)
        if (wants_sign(act, l.cout, ow)) {
            unsigned char *sg = reinterpret_cast<unsigned char *>(alloc(((size_t)n_ * oh * ow * (l.cout / 8) + 3) / 4));
            if (io16_ && g_experiment != 12) o.seg[0].sign = sg;
        }
        return run_conv(layer, x, act, o, 1, nchw_src, nchw_c);
    }
    bool wants_sign(int act, int cout, int ow) const { return training_ && !bn_on_ && act != PWS_ACT_NONE && cout % 8 == 0 && ow >= 32; }

    // ---- stages 2 and 3 in LOCKSTEP (round 4).  They run the same modules (reference :178-214) on different inputs, and every input of
    // stage 3 at a level is an output of stage 2 at that level's predecessor -- so with the three stages' tensors of a level laid out as
    // sample GROUPS of one buffer [stage 1 | stage 2 | stage 3] (each n samples), "what stages 2 and 3 read" are the two batch-2n views
    // groups (0, 1) and groups (1, 2) of it, and ONE launch of batch 2n computes both stages' outputs into groups (1, 2) of the next
    // buffer.  23 of the 68 launches of an inference forward go; what is left of the deep levels runs on twice the pixels per launch.
    // The tape still holds one op per stage (the backward is per stage, as before).
    struct Groups {   // a [groups][n][h][w][c] buffer
        float *ptr = nullptr;
        unsigned char *sign = nullptr;
        int c = 0, h = 0, w = 0;
    };
    size_t esz() const { return io16_ ? 2 : 4; }
    Groups alloc_groups(int ngroups, int h, int w, int c, int act) {
        Groups t;
        t.c = c, t.h = h, t.w = w;
        t.ptr = alloc((size_t)ngroups * n_ * h * w * c);
        if (wants_sign(act, c, w)) {
            unsigned char *sg = reinterpret_cast<unsigned char *>(alloc(((size_t)ngroups * n_ * h * w * (c / 8) + 3) / 4));
            if (io16_ && g_experiment != 12) t.sign = sg;
        }
        return t;
    }
    Tn group(const Groups &t, int g0) const {   // group g0 (read as batch 2n: groups g0 and g0 + 1)
        Tn o{};
        o.nseg = 1, o.h = t.h, o.w = t.w;
        float *p = t.ptr ? reinterpret_cast<float *>(reinterpret_cast<char *>(t.ptr) + (size_t)g0 * n_ * t.h * t.w * t.c * esz()) : nullptr;
        o.seg[0] = Seg{p, t.c, t.c, t.sign ? t.sign + (size_t)g0 * n_ * t.h * t.w * (t.c / 8) : nullptr};
        return o;
    }
    Tn shifted(const Tn &x, int k) const {   // the same (virtually concatenated) tensor k groups further
        Tn o = x;
        for (int i = 0; i < o.nseg; ++i) {
            Seg &sg = o.seg[i];
            if (sg.ptr) sg.ptr = reinterpret_cast<float *>(reinterpret_cast<char *>(sg.ptr) + (size_t)k * n_ * x.h * x.w * sg.ld * esz());
            if (sg.sign) sg.sign += (size_t)k * n_ * x.h * x.w * (sg.c / 8);
        }
        return o;
    }
    // layer on `ngroups` consecutive groups of x (batch ngroups * n) into group g0.. of dst; returns group g0's output
    Tn conv_groups(int layer, const Tn &x, int act, int ngroups, const Groups &dst, int g0) {
        return run_conv(layer, x, act, group(dst, g0), ngroups, nullptr, 0);
    }

    Tn run_conv(int layer, const Tn &x, int act, const Tn &o, int ngroups, const float *nchw_src, int nchw_c) {
        const Layer &l = L_[layer];
        const int oh = o.h, ow = o.w;
        Op op{OP_CONV, layer, act, 0, nchw_c > 0, x, o};
        if (bn_on_) op.aux[0] = alloc((size_t)n_ * oh * ow * l.cout), op.aux[1] = alloc(2 * (size_t)l.cout);
        tape_.push_back(op);
        for (int k = 1; k < ngroups; ++k) tape_.push_back(Op{OP_CONV, layer, act, 0, false, shifted(x, k), shifted(o, k)});
        if (!launch_ || rc_ != PWS_OK) return o;
        pws_conv_args a{};
        a.kind = l.kind, a.n = ngroups * n_, a.h = x.h, a.w = x.w;
        if (nchw_c > 0 && math_ == PWS_MATH_BF16 && l.wb_off != (size_t)-1 && x_nhwc_ && nchw_c <= 32) {
            // bf16 first layer: the NCHW window is re-laid once as a 32-channel NHWC source (kept for the weight gradient)
            rc_ = nchw_to_nhwc_pad_strided(nchw_src, x_sstride_, x_nhwc_, n_, nchw_c, x.h, x.w, 32, store(), streams_[q_]);
            if (rc_ != PWS_OK) return o;
            a.nsrc = 1, a.src[0] = pws_src{x_nhwc_, 32, 32};
        } else if (nchw_c > 0) {
            a.nsrc = 1, a.src_nchw = 1, a.src[0] = pws_src{nchw_src, nchw_c, (int)x_sstride_};
        } else {
            a.nsrc = x.nseg;
            for (int i = 0; i < x.nseg; ++i) a.src[i] = pws_src{x.seg[i].ptr, x.seg[i].c, x.seg[i].ld};
        }
        a.cout = l.cout, a.w_packed = packed_ + l.w_off, a.bias = packed_ + l.b_off, a.act = bn_on_ ? PWS_ACT_NONE : act;
        a.w_wino = l.ww_off != (size_t)-1 ? packed_ + l.ww_off : nullptr;
        a.w_wring = l.wr_off != (size_t)-1 ? packed_ + l.wr_off : nullptr;
        a.out = bn_on_ ? op.aux[0] : o.seg[0].ptr, a.out_ld = l.cout;
        a.ws = q_ ? splitk_ws2_ : splitk_ws_, a.ws_bytes = math_ == PWS_MATH_FP32 ? splitk_big_ : splitk_bytes_;
        if (math_ == PWS_MATH_BF16 && l.wb_off != (size_t)-1) {
            a.math = PWS_MATH_BF16, a.w_bf16 = packed_ + l.wb_off;
            // pws_netg_pack_weights_for(math = bf16) leaves this layer's Winograd copies out.  The bf16 kernels decline a call
            // with a source of channels % 32 != 0 (wb_off only needs the TOTAL cin % 32 == 0: ngf 16 / 48 in up_bottom1.conv_same);
            // pws_conv2d_fwd then falls through to the fp32 kernels, which must not find the stale Winograd pointers: the direct
            // kernel on w_packed (always packed) takes it.
            a.w_wino = a.w_wring = nullptr;
        }
        a.store = store();
        a.out_sign = o.seg[0].sign, a.out_sign_ld = l.cout / 8;
        g_prof_tag = layer;
        rc_ = pws_conv2d_fwd(&a, streams_[q_]);
        g_prof_tag = -1;
        if (bn_on_ && rc_ == PWS_OK) {
            // down_bottom1 is called twice per forward by the reference (x22, x32: same input, same weights) and computed once
            // here: its running statistics take both updates
            const int repeat = (layer == L_DB1_CS || layer == L_DB1_CS + 1) ? 2 : 1;
            rc_ = bn_forward(layer, op.aux[0], (size_t)n_ * oh * ow, act, o.seg[0].ptr, op.aux[1], repeat);
        }
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
    void theta(const Tn &x_s8, int stage, float *theta_out) {
        Op op{OP_THETA, L_FLATTEN, 0, stage, false, x_s8, Tn{}};
        if (bn_on_) {
            const int hidden = L_[L_FLATTEN].cout;
            op.aux[0] = alloc((size_t)n_ * hidden), op.aux[1] = alloc(2 * (size_t)hidden);
            op.aux[2] = alloc((size_t)n_ * 6), op.aux[3] = alloc(12);
        }
        tape_.push_back(op);
        if (!launch_ || rc_ != PWS_OK) return;
        const Layer &f = L_[L_FLATTEN], &l = L_[L_LINEAR];
        const float *xin = x_s8.seg[0].ptr;
        if (bn_on_) {
            // flatten -> BatchNorm over the n samples -> LeakyReLU -> linear -> BatchNorm -> LeakyReLU (reference :148-149 with use_BN)
            rc_ = theta_z1(xin, n_, x_s8.seg[0].c, f.cout, packed_ + f.w_off, packed_ + f.b_off, q_ ? theta_ws2_ : theta_ws_, op.aux[0],
                           streams_[q_]);
            if (rc_ == PWS_OK) rc_ = bn_forward(L_FLATTEN, op.aux[0], (size_t)n_, PWS_ACT_LRELU, h_saved_[stage], op.aux[1], 1);
            if (rc_ == PWS_OK) rc_ = theta_z2(h_saved_[stage], n_, f.cout, packed_ + l.w_off, packed_ + l.b_off, op.aux[2], streams_[q_]);
            if (rc_ == PWS_OK) rc_ = bn_forward(L_LINEAR, op.aux[2], (size_t)n_, PWS_ACT_LRELU, theta_out, op.aux[3], 1);
            return;
        }
        if (io16_) {  // the head is an fp32 GEMV: its 2x2xC input is converted once (n x 4C values)
            rc_ = pws_cvt_bf16_to_f32(xin, theta_x32_[q_], (size_t)n_ * 4 * x_s8.seg[0].c, streams_[q_]);
            if (rc_ != PWS_OK) return;
            xin = theta_x32_[q_];
        }
        rc_ = pws_theta_head_fwd_save(xin, n_, x_s8.seg[0].c, f.cout, packed_ + f.w_off, packed_ + f.b_off,
                                      packed_ + l.w_off, packed_ + l.b_off, q_ ? theta_ws2_ : theta_ws_, theta_out,
                                      h_saved_[stage], streams_[q_]);
    }
    // tanh(out(x)).permute(0,2,3,1) [+ affine_grid(theta)] (reference :174,235-237)
    void field(const Tn &x, int stage, const float *theta_k, int ac, float *resid, float *grid) {
        Op op{OP_FIELD, L_OUT, 0, stage, false, x, Tn{}};
        if (bn_on_) {
            const size_t e = (size_t)n_ * x.h * x.w * 2;
            op.aux[0] = alloc(e), op.aux[1] = alloc(4), op.aux[2] = alloc(e);
        }
        tape_.push_back(op);
        if (!launch_ || rc_ != PWS_OK) return;
        const Layer &o = L_[L_OUT];
        if (bn_on_) {
            // out conv -> BatchNorm(2) -> tanh, then the second tanh + permute + affine add (reference :128,174 with use_BN)
            rc_ = field_head_raw(x.seg[0].ptr, x.seg[0].ld, n_, x.h, x.w, x.seg[0].c, packed_ + o.w_off, packed_ + o.b_off, op.aux[0],
                                 streams_[q_]);
            if (rc_ == PWS_OK) rc_ = bn_forward(L_OUT, op.aux[0], (size_t)n_ * x.h * x.w, PWS_ACT_NONE, op.aux[2], op.aux[1], 1);
            if (rc_ == PWS_OK) rc_ = field_bn_finish(op.aux[2], theta_k, n_, x.h, x.w, ac, resid, grid, streams_[q_]);
            return;
        }
        rc_ = pws_field_head_fwd_s(x.seg[0].ptr, x.seg[0].ld, n_, x.h, x.w, x.seg[0].c, packed_ + o.w_off, packed_ + o.b_off,
                                   theta_k, ac, resid, grid, store(), streams_[q_]);
    }

    // scratch shared by all layers (launches are stream-ordered): split-K partial tiles, the theta head's partials,
    // and (training) the hidden activations of the three theta heads
    void reserve_scratch(int ngf, bool training) {
        training_ = training;
        splitk_bytes_ = (size_t)(2 * n_ > 8 ? 2 * n_ : 8) * (2u << 20);   // (2 n: stages 2 and 3 run as one launch of twice the batch)
        // fp32 math: four times that -- the Winograd ring kernel's two-class units of the transposed layers on 16 x 16 / 32 x 32 maps
        // split K four ways (33 MB of partial sums at batch 8; conv_wring.hip).  The bf16 kernels keep the smaller bound: deeper splits
        // of their small maps measured slower (configs[2] step +0.17 ms)
        // (training forwards in fp32 as well: netG(x, False) stays bit-equal to netG(x)[0][2], reference :237).  The deep splits are a
        // small-batch matter -- from batch 16 on those maps fill the chip without them -- so the extra is capped at the 48 MB batch 8
        // asks for: 2 x 176 MB of scratch at batch 64 instead of 2 x 512 MB in every arena
        splitk_big_ = splitk_bytes_ + ((size_t)48 << 20) < 4 * splitk_bytes_ ? splitk_bytes_ + ((size_t)48 << 20) : 4 * splitk_bytes_;
        splitk_ws_ = alloc(splitk_big_ / sizeof(float));
        splitk_ws2_ = alloc(splitk_big_ / sizeof(float));  // one scratch per queue: the two run concurrently
        theta_ws_ = alloc(pws_theta_head_ws_floats(n_, 4 * ngf, 8 * ngf));
        theta_ws2_ = alloc(pws_theta_head_ws_floats(n_, 4 * ngf, 8 * ngf));
        if (!splitk_ws_ || !splitk_ws2_) splitk_bytes_ = 0, splitk_big_ = 0;
        for (int s = 0; s < 3; ++s) h_saved_[s] = training ? alloc((size_t)n_ * 8 * ngf) : nullptr;
        x_nhwc_ = alloc((size_t)n_ * 256 * 256 * 32);  // bf16 math: NHWC copy of the window (unused in fp32 math)
        theta_x32_[0] = alloc((size_t)n_ * 16 * ngf), theta_x32_[1] = alloc((size_t)n_ * 16 * ngf);  // bf16 storage: fp32 copy of x_s8
        if (bn_on_) {
            bn_ws_bytes_ = pws_bn_ws_bytes(16 * ngf);   // slabs of the BatchNorm reductions, one scratch per queue
            bn_ws_[0] = alloc(bn_ws_bytes_ / sizeof(float)), bn_ws_[1] = alloc(bn_ws_bytes_ / sizeof(float));
        }
    }

  private:
    float *splitk_ws_ = nullptr, *splitk_ws2_ = nullptr, *theta_ws_ = nullptr, *theta_ws2_ = nullptr;
    float *h_saved_[3] = {nullptr, nullptr, nullptr};
    float *x_nhwc_ = nullptr;
    float *theta_x32_[2] = {nullptr, nullptr};
    bool bn_on_ = false;
    BnCfg bn_;
    size_t bn_off_[L_COUNT] = {}, bn_total_ = 0, bn_ws_bytes_ = 0;
    float *bn_ws_[2] = {nullptr, nullptr};
    bool io16_ = false, training_ = false;
    int math_ = PWS_MATH_FP32;
    size_t x_sstride_ = 0;
    bool prune_dead_ = false;
    SideStream *side_ = nullptr;
    hipStream_t streams_[2];
    int q_ = 0;
    size_t splitk_bytes_ = 0, splitk_big_ = 0;
    const float *packed_;
    const std::vector<Layer> &L_;
    int n_;
    char *ws_;
    size_t cap_, off_ = 0;
    bool dry_, launch_;
    int rc_ = PWS_OK;
    std::vector<Op> tape_;
};

// The forward with stages 2 and 3 in lockstep (Exec::Groups): every level's tensors of the three stages are the sample groups of ONE
// buffer, the layers stages 2 and 3 share run as one launch of batch 2n wherever both stages' inputs are such views.
//   encoder level k (k = 2 .. 8, maps 128 .. 1):  Ek = [x1k | x2k | x3k]   (x32 == x22: the reference computes it twice, :178,:200)
//       x2k, x3k = mpconv_{k-1}(cat(x_left, conv_same_{k-1}(x_up))),  x_up = (x1,k-1 | x2,k-1) = groups (0, 1) of E_{k-1},
//                                                                  x_left = (x2,k-1 | x3,k-1) = groups (1, 2)
//       conv_same merges from k = 3, mpconv from k = 4 (at k = 3 x_left would be (x22 | x32) = the same tensor twice)
//   decoder level l (l = 7 .. 1):  d_s(l) = cat(V_l[s], x_sl),  V_l = [u1l | v2l | v3l];  d_s(8) = x_s8
//       v_sl = mpconv_l(cat(conv_same_l(x_up), x_left)),  x_up = (d1 | d2)(l+1), x_left = (d2 | d3)(l+1): merged for l = 7 .. 2
//       (level 1 exists for stage 3 only at inference; in training its inputs d_s(2) hold x22 twice: separate launches)
// Queues: 0 = stage-1 encoder, stage-1 decoder, then the large merged decoder levels and the heads; 1 = the stage-2/3 encoder beside the
// stage-1 decoder, then the deep merged decoder levels.
static void forward_lockstep(Exec &E, const Tn &in, const float *x, int n, int input_nc, int g, int is_training, int ac, float *grids, float *resid,
                             float *th1, float *th2, float *th3) {
    const int S = 256;
    const size_t gsz = (size_t)n * S * S * 2;
    const int enc_c[9] = {0, g, g, 2 * g, 4 * g, 4 * g, 4 * g, 4 * g, 4 * g};        // channels of x_sk, k = 1 .. 8
    const int dec_c[8] = {0, g, g, 2 * g, 4 * g, 4 * g, 4 * g, 4 * g};               // channels of v_sl, l = 1 .. 7
    auto side = [&](int k) { return S >> (k - 1); };                                // map edge at level k
    // ---- buffers
    Exec::Groups Ek[9], Ck[9], Vl[8], Gl[8];
    for (int k = 2; k <= 8; ++k) Ek[k] = E.alloc_groups(k == 2 ? 2 : 3, side(k), side(k), enc_c[k], PWS_ACT_LRELU);   // (x32 == x22: E2 has no third group)
    for (int k = 3; k <= 8; ++k) Ck[k] = E.alloc_groups(2, side(k - 1), side(k - 1), enc_c[k - 1], PWS_ACT_LRELU);   // conv_same_{k-1}: level k-1 in, same size out
    for (int l = 7; l >= 2; --l) {
        Vl[l] = E.alloc_groups(3, side(l), side(l), dec_c[l], PWS_ACT_RELU);
        // conv_same of up_bottom level l: input x_up at level l + 1 (channels of d(l+1): v + skip, or the bottom x_s8), same size out
        const int cin = l == 7 ? enc_c[8] : dec_c[l + 1] + enc_c[l + 1];
        Gl[l] = E.alloc_groups(2, side(l + 1), side(l + 1), cin, PWS_ACT_RELU);
    }
    auto enc = [&](int k, int s) { return E.group(Ek[k], s - 1); };                 // x_sk
    auto dec = [&](int l, int s) {                                                   // d_s(l), l = 8 .. 2
        if (l == 8) return enc(8, s);
        Tn skip = s == 3 && l == 2 ? enc(2, 2) : enc(l, s);                          // x32 == x22
        return Exec::cat(E.group(Vl[l], s - 1), skip);
    };
    // ---- encoders.  sched (PWS_OPT_EXPERIMENT 150 + v, measured in tools/lockstep_sched.sh): how queue 1 follows queue 0
    const int sched = g_experiment >= 150 && g_experiment < 160 ? g_experiment - 150 : 0;
    auto db1 = [&](const Tn &x11) {   // down_bottom1 (:178 == :200): once, into group 1 of E2
        Tn c = E.conv(L_DB1_CS, x11, PWS_ACT_LRELU);
        E.conv_groups(L_DB1_CS + 1, c, PWS_ACT_LRELU, 1, Ek[2], 1);
    };
    auto level3 = [&]() {   // conv_same_2 on (x12 | x22) in one launch; the two mpconvs apart (both read x22 as x_left)
        E.conv_groups(L_DB1_CS + 2, enc(2, 1), PWS_ACT_LRELU, 2, Ck[3], 0);
        E.conv_groups(L_DB1_CS + 3, Exec::cat(enc(2, 2), E.group(Ck[3], 0)), PWS_ACT_LRELU, 1, Ek[3], 1);
        E.conv_groups(L_DB1_CS + 3, Exec::cat(enc(2, 2), E.group(Ck[3], 1)), PWS_ACT_LRELU, 1, Ek[3], 2);
    };
    auto enc1 = [&](int k, const Tn &x11) { E.conv_groups(L_DOWN1 + (k - 2), k == 2 ? x11 : enc(k - 1, 1), PWS_ACT_LRELU, 1, Ek[k], 0); };
    E.use(0);
    Tn x11 = E.conv(L_TRANSFER, in, PWS_ACT_LRELU, x, input_nc);
    hipEvent_t lvl[9] = {};
    if (sched == 4) {   // an event behind every level of stage 1: queue 1 follows level by level
        lvl[1] = E.mark(0);
        for (int k = 2; k <= 8; ++k) enc1(k, x11), lvl[k] = E.mark(0);
        E.use(1);
        E.wait(1, lvl[1]);
        db1(x11);
        E.wait(1, lvl[2]);
        level3();
    } else if (sched == 2) {   // the large early layers of stages 2 / 3 on queue 0 itself, queue 1 gets the merged levels only
        enc1(2, x11);
        db1(x11);
        level3();
        for (int k = 3; k <= 5; ++k) enc1(k, x11);
        E.order(0, 1);
        for (int k = 6; k <= 8; ++k) enc1(k, x11);
        E.order(0, 1);
        E.use(1);
    } else if (sched == 1) {   // queue 1 starts behind the whole stage-1 encoder
        for (int k = 2; k <= 8; ++k) enc1(k, x11);
        E.order(0, 1);
        E.use(1);
        db1(x11);
        level3();
    } else {   // 0 / 3: queue 1 starts behind x14 (its large layers beside the deeper levels of stage 1), the merged levels behind x18
        const int first = sched == 5 ? 3 : (sched == 6 ? 5 : 4);   // (5 / 6: behind x13 / x15 instead -- fp32 1815 / 1826 against 1840 f/s, bf16 6238 / 6150 against 6252)
        for (int k = 2; k <= first; ++k) enc1(k, x11);
        E.order(0, 1);
        E.use(1);
        db1(x11);
        level3();
        E.use(0);
        for (int k = first + 1; k <= 8; ++k) enc1(k, x11);
        E.order(0, 1);
        E.use(1);
    }
    if (is_training) {
        E.use(0);
        E.theta(enc(8, 1), 0, th1);
        E.use(1);
    }
    for (int k = 4; k <= 8; ++k) {   // levels 4 .. 8, merged, on queue 1
        const int db = k - 1;   // down_bottom index
        if (sched == 4) E.wait(1, lvl[k - 1]);
        E.conv_groups(L_DB1_CS + 2 * (db - 1), enc(k - 1, 1), PWS_ACT_LRELU, 2, Ck[k], 0);
        E.conv_groups(L_DB1_CS + 2 * (db - 1) + 1, Exec::cat(enc(k - 1, 2), E.group(Ck[k], 0)), PWS_ACT_LRELU, 2, Ek[k], 1);
    }
    if (sched == 4) E.wait(1, lvl[8]);   // (the bottom of stage 1: x_up of the deepest decoder level)
    // (the theta heads of stages 2 / 3 -- 30 us of GEMV launches the field heads need at the very end -- are issued on queue 1 BEHIND the deep
    //  decoder levels, where that queue has nothing else to do: in front of them they sat on the critical path)
    // ---- stage 1 decoder (reference :166-174) on queue 0
    E.use(0);
    // (PWS_NETG_PRUNE_DEAD, inference: u12 = up2's output is read by up1 and stage 2's up_bottom1 only, reference :173,:196 -- both `if is_training`)
    const int l_last = (!is_training && E.prune_dead()) ? 3 : 2;
    for (int l = 7; l >= l_last; --l) {
        E.conv_groups(L_UP7 + (7 - l), dec(l + 1, 1), PWS_ACT_RELU, 1, Vl[l], 0);
        if (l == 6) E.order(0, 1);   // u17, u16: what the deep merged levels (7 .. 5, queue 1) read of stage 1
    }
    if (is_training) {
        Tn x111 = E.up(1, dec(2, 1), nullptr);
        E.field(x111, 0, th1, ac, resid, grids);
    }
    // ---- stages 2 / 3 decoder (reference :190-198, :212-219): levels 7 .. 5 on queue 1 (behind the encoder, beside the large levels of the
    // stage-1 decoder), levels 4 .. 2 on queue 0
    auto merged_level = [&](int l) {
        E.conv_groups(L_UB7_MP + 2 * (7 - l) + 1, dec(l + 1, 1), PWS_ACT_RELU, 2, Gl[l], 0);
        E.conv_groups(L_UB7_MP + 2 * (7 - l), Exec::cat(E.group(Gl[l], 0), dec(l + 1, 2)), PWS_ACT_RELU, 2, Vl[l], 1);
    };
    if (sched == 3) {   // everything of the decoders on queue 0
        E.order(1, 0);
        E.use(0);
        for (int l = 7; l >= 5; --l) merged_level(l);
    } else {
        E.use(1);
        for (int l = 7; l >= 5; --l) merged_level(l);
        E.order(1, 0);
    }
    E.use(1);
    if (is_training) E.theta(enc(8, 2), 1, th2);
    E.theta(enc(8, 3), 2, th3);
    hipEvent_t th_done = E.mark(1);
    E.use(0);
    for (int l = 4; l >= 2; --l) merged_level(l);
    // ---- level 1 and the field heads
    E.wait(0, th_done);
    if (is_training) {
        Tn x211 = E.up_bottom(1, dec(2, 1), dec(2, 2), nullptr);
        E.field(x211, 1, th2, ac, resid ? resid + gsz : nullptr, grids ? grids + gsz : nullptr);
    }
    Tn x311 = E.up_bottom(1, dec(2, 2), dec(2, 3), nullptr);
    if (is_training)
        E.field(x311, 2, th3, ac, resid ? resid + 2 * gsz : nullptr, grids ? grids + 2 * gsz : nullptr);
    else
        E.field(x311, 2, th3, ac, nullptr, grids);
}

// Runs (or only plans) the forward over the arena of E.  thetas may be NULL (then they live in the arena).
// caller_thetas: the caller owns the thetas buffer even when `thetas` is NULL (a planning replay of the training forward: no arena space for them)
static void forward_graph(Exec &E, const float *x, int n, int input_nc, int g, int is_training, int ac, float *grids,
                          float *resid, float *thetas, bool caller_thetas = false) {
    const int S = 256;
    E.reserve_scratch(g, is_training != 0);
    const size_t gsz = (size_t)n * S * S * 2;
    float *th = (thetas || caller_thetas) ? thetas : E.alloc((size_t)3 * n * 6);
    float *th1 = th, *th2 = th ? th + (size_t)n * 6 : nullptr, *th3 = th ? th + (size_t)2 * n * 6 : nullptr;

    Tn in{};
    in.nseg = 1, in.h = S, in.w = S, in.seg[0] = Seg{nullptr, input_nc, 0};
    if (!E.bn_on() && g_experiment != 15) {   // (15: the per-stage schedule below, A/B; the BatchNorm path keeps it: statistics are per call)
        forward_lockstep(E, in, x, n, input_nc, g, is_training, ac, grids, resid, th1, th2, th3);
        return;
    }
    // Queue 0 (the caller's stream): stage-1 encoder, then the three decoders.  Queue 1: the stage-2 and stage-3 encoders,
    // which only need the previous stage's ENCODER outputs -- so stage 1's latency-bound deep layers and its decoder
    // overlap with stage 2's big conv_same layers, and stage 2's decoder overlaps with stage 3's encoder.
    // The thetas of stages 1 and 2 are only outputs of the training mode (reference :235) and are skipped otherwise.
    // ---- stage 1 encoder (reference :153-160)
    E.use(0);
    Tn x11 = E.conv(L_TRANSFER, in, PWS_ACT_LRELU, x, input_nc);
    Tn x12 = E.down(1, x11), x13 = E.down(2, x12), x14 = E.down(3, x13), x15 = E.down(4, x14);
    E.order(0, 1);  // x11..x15 ready: queue 1 may start stage 2's encoder while queue 0 finishes the deep levels
    // ---- stage 2 encoder, first levels (reference :178-181) on queue 1
    E.use(1);
    Tn x22 = E.down_bottom(1, nullptr, x11);
    Tn x23 = E.down_bottom(2, &x22, x12), x24 = E.down_bottom(3, &x23, x13), x25 = E.down_bottom(4, &x24, x14);
    E.use(0);
    Tn x16 = E.down(5, x15), x17 = E.down(6, x16), x18 = E.down(7, x17);
    E.order(0, 1);  // x16..x18
    // (the theta heads are needed by the field heads only: they are issued BEHIND the marks the other queue waits for -- 30 us of
    //  GEMV launches off the critical path of every stage)
    if (is_training) E.theta(x18, 0, th1);
    E.use(1);
    Tn x26 = E.down_bottom(5, &x25, x15), x27 = E.down_bottom(6, &x26, x16), x28 = E.down_bottom(7, &x27, x17);
    hipEvent_t enc2_done = E.mark(1);
    if (is_training) E.theta(x28, 1, th2);
    hipEvent_t th2_done = is_training ? E.mark(1) : nullptr;
    // ---- stage 1 decoder (reference :166-174) on queue 0, beside the stage-2 encoder
    E.use(0);
    Tn x177 = E.up(7, x18, &x17), x166 = E.up(6, x177, &x16), x155 = E.up(5, x166, &x15);
    Tn x144 = E.up(4, x155, &x14), x133 = E.up(3, x144, &x13), x122 = E.up(2, x133, &x12);
    if (is_training) {
        Tn x111 = E.up(1, x122, nullptr);
        E.field(x111, 0, th1, ac, resid, grids);
    }
    // ---- stage 3 encoder (reference :200-206) continues on queue 1; x32 == x22 (same weights, same input)
    E.use(1);
    const Tn &x32 = x22;
    Tn x33 = E.down_bottom(2, &x32, x22), x34 = E.down_bottom(3, &x33, x23), x35 = E.down_bottom(4, &x34, x24);
    Tn x36 = E.down_bottom(5, &x35, x25), x37 = E.down_bottom(6, &x36, x26), x38 = E.down_bottom(7, &x37, x27);
    hipEvent_t enc3_done = E.mark(1);
    // ---- stage 2 decoder (reference :190-198) on queue 0, beside the stage-3 encoder: waits for stage 2's encoder only
    E.wait(0, enc2_done);
    E.use(0);
    Tn x277 = E.up_bottom(7, x18, x28, &x27), x266 = E.up_bottom(6, x177, x277, &x26);
    Tn x255 = E.up_bottom(5, x166, x266, &x25);
    hipEvent_t dec2_deep = E.mark(0);
    // ---- stage 3 decoder (reference :212-219), deep levels: on queue 1 behind the stage-3 encoder, beside the LARGE levels of the stage-2
    // decoder on queue 0 (nine latency-bound launches on <= 8 x 8 maps under three chip-filling ones); PWS_OPT_EXPERIMENT 14: on queue 0
    // after the whole stage-2 decoder, as before
    const bool deep_on_q1 = g_experiment != 14 && !E.bn_on();
    Tn x377{}, x366{}, x355{};
    hipEvent_t dec3_deep = nullptr;
    if (deep_on_q1) {
        E.use(1);
        E.wait(1, dec2_deep);
        x377 = E.up_bottom(7, x28, x38, &x37), x366 = E.up_bottom(6, x277, x377, &x36), x355 = E.up_bottom(5, x266, x366, &x35);
        dec3_deep = E.mark(1);
        E.use(0);
    }
    Tn x244 = E.up_bottom(4, x155, x255, &x24);
    Tn x233 = E.up_bottom(3, x144, x244, &x23), x222 = E.up_bottom(2, x133, x233, &x22);
    if (is_training) {
        Tn x211 = E.up_bottom(1, x122, x222, nullptr);
        E.wait(0, th2_done);
        E.field(x211, 1, th2, ac, resid ? resid + gsz : nullptr, grids ? grids + gsz : nullptr);
    }
    E.use(1);
    E.theta(x38, 2, th3);   // (behind everything the other queue waits for)
    hipEvent_t th3_done = E.mark(1);
    E.use(0);
    // ---- stage 3 decoder, large levels: joins queue 1 (nothing is issued there after the theta head)
    E.wait(0, enc3_done);
    if (deep_on_q1) E.wait(0, dec3_deep);
    else x377 = E.up_bottom(7, x28, x38, &x37), x366 = E.up_bottom(6, x277, x377, &x36), x355 = E.up_bottom(5, x266, x366, &x35);
    Tn x344 = E.up_bottom(4, x255, x355, &x34);
    Tn x333 = E.up_bottom(3, x244, x344, &x33), x322 = E.up_bottom(2, x233, x333, &x32);
    Tn x311 = E.up_bottom(1, x222, x322, nullptr);
    E.wait(0, th3_done);
    if (is_training)
        E.field(x311, 2, th3, ac, resid ? resid + 2 * gsz : nullptr, grids ? grids + 2 * gsz : nullptr);
    else
        E.field(x311, 2, th3, ac, nullptr, grids);
}

struct GradBuf {
    float *g;
    bool written;
    bool preact;   // the buffer already holds the gradient wrt the producer's PRE-activation (act' fused into the last data-gradient
                   // call that wrote it, pws_dst.act_y): the producer then needs the bias sum only
};

// The forward passes its thetas pointer; the planning replay must consume the arena identically, so backward is
// told whether the forward kept thetas in the arena (thetas_in_arena) -- the Python host always passes its own buffer.
// part / nparts: the reversed tape is cut into nparts runs of (almost) equal op count and only run `part` is launched (the
// bookkeeping of every earlier run is replayed without launching), so that a data-parallel host can all-reduce the weight
// gradients that are already final while the later runs still compute.  final_mask (nullable, one byte per layer): 1 once no
// op of a later run contributes to that layer's gradient.
// upstream gradients of the six fields, one pointer per stage (NULL: that output has no gradient)
struct UpGrads {
    const float *grids[3] = {nullptr, nullptr, nullptr}, *resid[3] = {nullptr, nullptr, nullptr};
    UpGrads() = default;
    UpGrads(const float *g_grids, const float *g_resid, size_t gsz) {   // the stacked [3][n,256,256,2] form
        for (int k = 0; k < 3; ++k) grids[k] = g_grids ? g_grids + k * gsz : nullptr, resid[k] = g_resid ? g_resid + k * gsz : nullptr;
    }
    UpGrads(const float *const *g_grids, const float *const *g_resid) {   // HOST arrays of 3 device pointers (either may be NULL)
        for (int k = 0; k < 3; ++k) grids[k] = g_grids ? g_grids[k] : nullptr, resid[k] = g_resid ? g_resid[k] : nullptr;
    }
    bool any() const {
        for (int k = 0; k < 3; ++k)
            if (grids[k] || resid[k]) return true;
        return false;
    }
};

static int run_backward(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int g, int ac,
                        char *ws, size_t ws_bytes, const float *resid, const float *thetas, const UpGrads &up, float *dpacked,
                        hipStream_t st, bool dry, size_t *used, int part = 0,
                        int nparts = 1, unsigned char *final_mask = nullptr, const BnCfg *bn = nullptr, float *dbn = nullptr,
                        const NetgOpts *opts = nullptr) {
    const int S = 256;
    size_t total = 0, total_grad = 0;
    const std::vector<Layer> L = build_layers(input_nc, g, &total, nullptr, &total_grad);
    const NetgOpts mode = opts ? *opts : opts_defaults();
    DeterministicScope det_scope(mode.deterministic);   // PWS_NETG_DETERMINISTIC: every accumulating launcher of this call
    Exec E(packed, L, n, ws, ws_bytes, st, dry, /*launch=*/false, mode, bn);
    // (the training forward is always given a caller-owned thetas buffer: the replay must not reserve arena space for them either)
    forward_graph(E, x, n, input_nc, g, 1, ac, nullptr, nullptr, nullptr, /*caller_thetas=*/true);
    // gradient buffers, one per produced tensor, after the forward region of the arena
    std::unordered_map<const float *, GradBuf> G;
    for (size_t i = 0; i < E.tape().size(); ++i) {
        const Op &op = E.tape()[i];
        if (op.type != OP_CONV) continue;
        const Seg &o = op.out.seg[0];
        float *gp = E.alloc((size_t)n * op.out.h * op.out.w * o.c);
        if (!dry) G[o.ptr] = GradBuf{gp, false, false};
    }
    float *gz_ws = E.alloc((size_t)n * S * S * 2);
    float *th_bwd_ws = E.alloc((size_t)n * 8 * g);
    float *dtheta = E.alloc((size_t)3 * n * 6);
    float *th_dx32 = E.alloc((size_t)n * 16 * g);  // bf16 storage: fp32 gradient wrt x_s8 before it is folded into the bf16 buffer
    const size_t abb_bytes = pws_act_bwd_bias_ws_bytes(16 * g);  // slabs of the bias-gradient reduction (max cout <= 16 ngf)
    float *abb_ws = E.alloc(abb_bytes / sizeof(float));
    if (used) *used = E.used();
    if (dry) return PWS_OK;
    if (E.rc() != PWS_OK) return E.rc();

    if (part == 0 && bn && dbn) {
        hipError_t e = hipMemsetAsync(dbn, 0, E.bn_total() * sizeof(float), st);
        if (e != hipSuccess) {
            set_error("pws_netg_backward: hipMemsetAsync: %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
    }
    // BatchNorm backward of `layer`: dy -> dz in place, dgamma / dbeta accumulated
    auto bn_backward = [&](int layer, float *dy, const float *y, const float *z, const float *stats, int act, size_t pixels) {
        const int c = L[layer].cout;
        float *dg = dbn + E.bn_off(layer);
        return pws_bn_train_bwd(dy, y, z, stats, E.gamma(layer), act, pixels, c, dg, dg + c, E.bn_ws(0), E.bn_ws_bytes(), st);
    };
    if (part == 0) {
        hipError_t e = hipMemsetAsync(dpacked, 0, total_grad * sizeof(float), st);
        if (e != hipSuccess) {
            set_error("pws_netg_backward: hipMemsetAsync: %s", hipGetErrorString(e));
            return PWS_EHIP;
        }
    }
    const size_t T = E.tape().size();
    const size_t r_begin = T * (size_t)part / nparts, r_end = T * (size_t)(part + 1) / nparts;  // reversed positions of this run
    // Weight gradients are leaves of the backward graph (only the optimizer reads them): they run on the side queue, each behind an
    // event recorded where its dy is final, while this queue goes on with the data-gradient chain -- whose deep levels are ~100
    // latency-bound launches that then sit beside chip-filling weight-gradient kernels instead of between them.  The queues join at
    // the end of the call.  (PWS_OPT_EXPERIMENT 16 or two_queues == 0: one queue, as before.)
    hipStream_t wst = st;
    SideStream *wside = nullptr;
    if (mode.two_queues && g_experiment != 16) {
        wside = &side_stream();
        if (hipStream_t q = wside->pick(st)) wst = q, wside->begin(st);
        else wside = nullptr;
    }
    auto wgrad_fork = [&]() -> int {   // the side queue waits for everything issued on `st` so far
        if (wst == st) return PWS_OK;
        hipEvent_t e = wside->event();
        if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(wst, e, 0) != hipSuccess) {
            set_error("pws_netg_backward: event record / wait on the weight-gradient queue failed");
            return PWS_EHIP;
        }
        return PWS_OK;
    };
    if (final_mask) {
        // a layer is final once every op that uses it lies at a reversed position < r_end
        for (int i = 0; i < L_COUNT; ++i) final_mask[i] = 1;
        for (size_t ii = 0; ii < T; ++ii) {
            if (T - 1 - ii < r_end) continue;
            const Op &op = E.tape()[ii];
            if (op.type == OP_FIELD) final_mask[L_OUT] = 0;
            else if (op.type == OP_THETA) final_mask[L_FLATTEN] = 0, final_mask[L_LINEAR] = 0;
            else final_mask[op.layer] = 0;
        }
    }
    const size_t gsz = (size_t)n * S * S * 2;
    // bf16 storage: act' of a conv output is applied by the LAST data-gradient call that writes its gradient buffer
    // (pws_dst.act_y) instead of by a separate pass over the buffer.  A dry sweep finds, per tensor, the tape position of that
    // last writer (same control flow as the real sweep below) and the op that produced the tensor.
    std::unordered_map<const float *, size_t> last_writer, producer;
    const bool fuse_act = E.io16() && !bn && mode.math == PWS_MATH_BF16 && g_experiment != 11;
    if (fuse_act) {
        std::unordered_map<const float *, bool> wr;
        bool hd[3] = {false, false, false};
        for (size_t ii = T; ii-- > 0;) {
            const Op &op = E.tape()[ii];
            if (op.type == OP_FIELD) {
                const float *gg = up.grids[op.stage], *gr = up.resid[op.stage];
                if (!gg && !gr) continue;
                wr[op.in.seg[0].ptr] = true, last_writer[op.in.seg[0].ptr] = ii, hd[op.stage] = gg != nullptr;
            } else if (op.type == OP_THETA) {
                if (!hd[op.stage]) continue;
                wr[op.in.seg[0].ptr] = true, last_writer[op.in.seg[0].ptr] = ii;
            } else {
                producer[op.out.seg[0].ptr] = ii;
                if (!wr[op.out.seg[0].ptr] || op.nchw) continue;
                for (int i = 0; i < op.in.nseg; ++i) wr[op.in.seg[i].ptr] = true, last_writer[op.in.seg[i].ptr] = ii;
            }
        }
    }
    // act to fuse into the data-gradient call of tape position ii for the destination that is the gradient of tensor t (0: none)
    auto fused_act = [&](size_t ii, const float *t) -> int {
        if (!fuse_act) return PWS_ACT_NONE;
        const auto lw = last_writer.find(t);
        const auto pr = producer.find(t);
        if (lw == last_writer.end() || lw->second != ii || pr == producer.end()) return PWS_ACT_NONE;
        const Op &pop = E.tape()[pr->second];
        if (E.tape()[ii].type == OP_THETA) return PWS_ACT_NONE;   // the theta head's (tiny) backward does not take the option
        if (E.tape()[ii].type == OP_CONV && L[E.tape()[ii].layer].dgb_off == (size_t)-1) return PWS_ACT_NONE;   // this call runs an fp32 kernel
        return pop.act == PWS_ACT_LRELU || pop.act == PWS_ACT_RELU ? pop.act : PWS_ACT_NONE;
    };
    bool have_dtheta[3] = {false, false, false};
    int rc = PWS_OK;
    int layer_uses[L_COUNT] = {};
    for (size_t ii = 0; ii < T; ++ii)
        if (E.tape()[ii].type == OP_CONV) ++layer_uses[E.tape()[ii].layer];
    std::unordered_map<int, pws_conv_bwd_weight_args> pending;   // first weight-gradient operand pair of a layer used twice (below)
    for (size_t ii = T; ii-- > 0 && rc == PWS_OK;) {
        const Op &op = E.tape()[ii];
        const size_t r = T - 1 - ii;
        if (r >= r_end) break;
        const bool run = r >= r_begin;  // earlier runs: replay the bookkeeping (written / have_dtheta flags) only
        if (op.type == OP_FIELD) {
            const int k = op.stage;
            const float *gg = up.grids[k], *gr = up.resid[k];
            if (!gg && !gr) continue;
            const Seg &xs = op.in.seg[0];
            GradBuf &gb = G[xs.ptr];
            const Layer &o = L[L_OUT];
            if (run && bn) {
                rc = field_bwd_gz(resid + k * gsz, gg, gr, n, op.in.h, op.in.w, ac, gz_ws, nullptr, gg ? dtheta + (size_t)k * n * 6 : nullptr, st);
                if (rc == PWS_OK) rc = bn_backward(L_OUT, gz_ws, nullptr, op.aux[0], op.aux[1], PWS_ACT_NONE, (size_t)n * op.in.h * op.in.w);
                if (rc == PWS_OK)
                    rc = field_bwd_dx_dw(xs.ptr, xs.ld, gz_ws, n, op.in.h, op.in.w, xs.c, packed + o.w_off, gb.g, xs.c, gb.written ? 1 : 0,
                                         dpacked + o.gw_off, E.store(), st);
            } else if (run)
                rc = pws_field_head_bwd_act(xs.ptr, xs.ld, n, op.in.h, op.in.w, xs.c, packed + o.w_off, resid + k * gsz, gg, gr, ac, gb.g,
                                            xs.c, gb.written ? 1 : 0, dpacked + o.gw_off, dpacked + o.gb_off,
                                            gg ? dtheta + (size_t)k * n * 6 : nullptr, gz_ws, E.store(), fused_act(ii, xs.ptr), st);
            gb.written = true;
            if (fused_act(ii, xs.ptr) != PWS_ACT_NONE) gb.preact = true;
            have_dtheta[k] = gg != nullptr;
        } else if (op.type == OP_THETA) {
            const int k = op.stage;
            if (!have_dtheta[k]) continue;
            const Seg &xs = op.in.seg[0];
            GradBuf &gb = G[xs.ptr];
            const Layer &f = L[L_FLATTEN], &l = L[L_LINEAR];
            if (!run) {
                gb.written = true;
                continue;
            }
            if (bn) {
                float *dz2 = dtheta + (size_t)k * n * 6;   // overwritten by the BatchNorm backward
                rc = bn_backward(L_LINEAR, dz2, thetas + (size_t)k * n * 6, op.aux[2], op.aux[3], PWS_ACT_LRELU, (size_t)n);
                if (rc == PWS_OK) rc = theta_bwd_bn_lin(dz2, E.h_saved(k), n, f.cout, packed + l.w_off, dpacked + l.gw_off, th_bwd_ws, st);
                if (rc == PWS_OK) rc = bn_backward(L_FLATTEN, th_bwd_ws, E.h_saved(k), op.aux[0], op.aux[1], PWS_ACT_LRELU, (size_t)n);
                if (rc == PWS_OK)
                    rc = theta_bwd_flat(xs.ptr, n, xs.c, f.cout, packed + f.w_off, th_bwd_ws, dpacked + f.gw_off, gb.g, gb.written ? 1 : 0, st);
                gb.written = true;
                continue;
            }
            if (E.io16()) {
                // fp32 head on an fp32 copy of its bf16 input; its input gradient goes through an fp32 scratch
                const size_t cnt = (size_t)n * 4 * xs.c;
                rc = pws_cvt_bf16_to_f32(xs.ptr, E.theta_x32(0), cnt, st);
                if (rc == PWS_OK)
                    rc = pws_theta_head_bwd(E.theta_x32(0), n, xs.c, f.cout, packed + f.w_off, packed + l.w_off, E.h_saved(k),
                                            thetas + (size_t)k * n * 6, dtheta + (size_t)k * n * 6, dpacked + f.gw_off,
                                            dpacked + f.gb_off, dpacked + l.gw_off, dpacked + l.gb_off, th_dx32, 0, th_bwd_ws, st);
                if (rc == PWS_OK) rc = pws_cvt_f32_to_bf16(th_dx32, gb.g, cnt, gb.written ? 1 : 0, st);
            } else {
                rc = pws_theta_head_bwd(xs.ptr, n, xs.c, f.cout, packed + f.w_off, packed + l.w_off, E.h_saved(k),
                                        thetas + (size_t)k * n * 6, dtheta + (size_t)k * n * 6, dpacked + f.gw_off, dpacked + f.gb_off,
                                        dpacked + l.gw_off, dpacked + l.gb_off, gb.g, gb.written ? 1 : 0, th_bwd_ws, st);
            }
            gb.written = true;
        } else {
            const Layer &l = L[op.layer];
            const Seg &o = op.out.seg[0];
            GradBuf &go = G[o.ptr];
            if (!go.written) continue;  // nothing downstream asked for a gradient
            const size_t pixels = (size_t)n * op.out.h * op.out.w;
            // the weight-gradient call of this op (pointers only: built the same way whether this run launches it or replays it)
            auto make_wa = [&]() {
                pws_conv_bwd_weight_args wa{};
                wa.kind = l.kind, wa.n = n, wa.h = op.in.h, wa.w = op.in.w, wa.nsrc = op.in.nseg, wa.src_nchw = op.nchw ? 1 : 0;
                for (int i = 0; i < op.in.nseg; ++i) wa.src[i] = pws_src{op.in.seg[i].ptr, op.in.seg[i].c, op.in.seg[i].ld};
                if (op.nchw) wa.src[0] = pws_src{x, input_nc, 0};
                if (op.nchw && mode.math == PWS_MATH_BF16 && l.wb_off != (size_t)-1 && E.x_nhwc() && input_nc <= 32)
                    wa.src_nchw = 0, wa.src[0] = pws_src{E.x_nhwc(), 32, 32};  // the forward's NHWC copy
                wa.cout = l.cout, wa.gout = go.g, wa.gout_ld = l.cout, wa.dw_packed = dpacked + l.gw_off;
                wa.math = mode.math, wa.store = E.store();
                // (deterministic: the bias sum is an ordered pass of its own on `st`, below)
                wa.dbias = !bn && go.preact && !mode.deterministic ? dpacked + l.gb_off : nullptr;
                return wa;
            };
            // A layer that stages 2 AND 3 run (down_bottom2..7, up_bottom7..1: the same modules, reference :178-214) comes by twice: its
            // first weight gradient (stage 3, earlier in the reversed tape) waits in `pending` and goes out WITH the second one as ONE
            // launch over both operand pairs (pws_conv_bwd_weight_args.gout2: one prologue, one set of epilogue atomics -- 5-11 % less
            // than two launches; the first pair's buffers stay untouched until then: a gradient buffer is final once its producer has
            // been processed).  Returns the call to launch now (nsrc == 0: nothing -- deferred).  PWS_OPT_EXPERIMENT 89 and the kinds
            // the ring kernels do not cover run the pair as two launches inside pws_conv2d_bwd_weight.
            const bool shared = !bn && !op.nchw && layer_uses[op.layer] == 2;
            pws_conv_bwd_weight_args extra{};   // an unmergeable first pair that has to go out by itself (nsrc == 0: none)
            auto pair_up = [&](pws_conv_bwd_weight_args wa) {
                if (!shared) return wa;
                auto mine = pending.find(op.layer);
                if (mine == pending.end()) {
                    pending.emplace(op.layer, wa);
                    wa.nsrc = 0;
                    return wa;
                }
                const pws_conv_bwd_weight_args first = mine->second;
                pending.erase(mine);
                if ((first.dbias != nullptr) != (wa.dbias != nullptr)) {   // one pair's bias sum is already in: no merged bias pass
                    extra = first;
                    return wa;
                }
                for (int i = 0; i < wa.nsrc; ++i) wa.src2_ptr[i] = first.src[i].ptr;
                wa.gout2 = first.gout;
                return wa;
            };
            if (!run) {
                if (!op.nchw)
                    for (int i = 0; i < op.in.nseg; ++i) {
                        GradBuf &gi = G[op.in.seg[i].ptr];
                        gi.written = true;
                        if (fused_act(ii, op.in.seg[i].ptr) != PWS_ACT_NONE) gi.preact = true;
                    }
                (void)pair_up(make_wa());   // an earlier run launched (or deferred) it: keep `pending` as that run left it
                continue;
            }
            g_prof_tag = op.layer;
            if (bn)   // the conv bias gets no gradient: BatchNorm removes any per-channel constant (torch returns rounding noise)
                rc = bn_backward(op.layer, go.g, o.ptr, op.aux[0], op.aux[1], op.act, pixels);
            else if (!go.preact)   // (a pre-activation gradient needs the bias sum only: the weight-gradient kernel takes it along)
                rc = pws_act_bwd_bias_s(go.g, o.ptr, pixels, l.cout, op.act, dpacked + l.gb_off, E.store(), abb_ws, abb_bytes, st);
            else if (mode.deterministic)
                // the weight-gradient kernels' bias sums arrive per parity class / class pair in any order: an ordered pass instead
                // (act = NONE: sums only, slab partials + ordered reduction)
                rc = pws_act_bwd_bias_s(go.g, go.g, pixels, l.cout, PWS_ACT_NONE, dpacked + l.gb_off, E.store(), abb_ws, abb_bytes, st);
            if (rc != PWS_OK) break;
            const pws_conv_bwd_weight_args wa = pair_up(make_wa());
            if (wa.nsrc > 0 || extra.nsrc > 0) {
                rc = wgrad_fork();
                if (rc == PWS_OK && extra.nsrc > 0) rc = pws_conv2d_bwd_weight(&extra, wst);
                if (rc == PWS_OK && wa.nsrc > 0) rc = pws_conv2d_bwd_weight(&wa, wst);
            }
            if (rc != PWS_OK || op.nchw) {
                g_prof_tag = -1;
                continue;  // the window is data: no gradient wrt the first layer's input
            }
            pws_conv_bwd_data_args da{};
            da.kind = l.kind, da.n = n, da.h = op.in.h, da.w = op.in.w, da.cout = l.cout, da.gout = go.g, da.gout_ld = l.cout;
            da.w_dgrad = packed_dgrad + l.dg_off, da.ndst = op.in.nseg;
            for (int i = 0; i < op.in.nseg; ++i) {
                GradBuf &gi = G[op.in.seg[i].ptr];
                da.dst[i] = pws_dst{gi.g, op.in.seg[i].c, op.in.seg[i].c, gi.written ? 1 : 0, nullptr, 0, PWS_ACT_NONE, nullptr, 0};
                gi.written = true;
                const int fa = fused_act(ii, op.in.seg[i].ptr);
                if (fa != PWS_ACT_NONE)
                    da.dst[i].act_y = op.in.seg[i].ptr, da.dst[i].act_y_ld = op.in.seg[i].ld, da.dst[i].act = fa, gi.preact = true,
                    da.dst[i].act_sign = op.in.seg[i].sign, da.dst[i].act_sign_ld = op.in.seg[i].c / 8;
            }
            da.ws = E.splitk_ws(), da.ws_bytes = E.splitk_bytes();
            if (mode.math == PWS_MATH_BF16 && l.dgb_off != (size_t)-1) da.math = PWS_MATH_BF16, da.w_dgrad_bf16 = packed_dgrad + l.dgb_off;
            da.store = E.store();
            rc = pws_conv2d_bwd_data(&da, st);
            g_prof_tag = -1;
        }
    }
    // a deferred first pair whose partner never came (its output got no gradient): it goes out in the run that holds the layer's last op
    if (rc == PWS_OK && !pending.empty()) {
        for (auto &kv : pending) {
            size_t last_r = 0;
            for (size_t ii = 0; ii < T; ++ii)
                if (E.tape()[ii].type == OP_CONV && E.tape()[ii].layer == kv.first) { last_r = T - 1 - ii; break; }
            if (last_r < r_begin || last_r >= r_end) continue;
            rc = wgrad_fork();
            if (rc == PWS_OK) rc = pws_conv2d_bwd_weight(&kv.second, wst);
            if (rc != PWS_OK) break;
        }
    }
    if (wst != st) {   // join: whatever follows on `st` (unpack, optimizer, the next part) sees every weight gradient
        hipEvent_t e = wside->event();
        if (!e || hipEventRecord(e, wst) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) {
            if (rc == PWS_OK) set_error("pws_netg_backward: joining the weight-gradient queue failed"), rc = PWS_EHIP;
        }
    }
    return rc;
}

}  // namespace pws

using namespace pws;

extern "C" size_t pws_netg_packed_floats(int input_nc, int ngf) {
    if (input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0;
    build_layers(input_nc, ngf, &total);
    return total;
}

extern "C" size_t pws_netg_packed_dgrad_floats(int input_nc, int ngf) {
    if (input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0, dg = 0;
    build_layers(input_nc, ngf, &total, &dg);
    return dg;
}

extern "C" int pws_netg_pack_weights(const float *const *params, float *packed, int input_nc, int ngf, pws_stream_t stream) {
    return pws_netg_pack_weights_for(params, packed, input_nc, ngf, -1, stream);
}

// skip (nullable, one flag per layer): layers another launch packs (pws_netg_pack_weights_train's one-pass kernel) get no blocks here
static int pack_weights_impl(const float *const *params, float *packed, int input_nc, int ngf, int math, pws_stream_t stream, const bool *skip) {
    PWS_REQUIRE(params && packed, "pws_netg_pack_weights: NULL pointer");
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_pack_weights: ngf must be a positive multiple of 16 (got %d)",
                ngf);
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    PWS_REQUIRE(total < 0xffffffffu, "pws_netg_pack_weights: packed buffer too large for 32-bit offsets");
    // three launches over all layers (netg_pack.hip): torch layouts -> packed fp32 (+ biases), Winograd, bf16.  Alignment gaps
    // between the regions are never read, so there is no zero-fill pass.
    PackAllArgs a{};
    Bf16AllArgs b{};
    a.nlayers = b.nlayers = L_COUNT;
    fill_pack_layers(L, a.layer);
    unsigned nb = 0, nbw = 0, nbb = 0;
    for (int i = 0; i < L_COUNT; ++i) {
        PWS_REQUIRE(params[2 * i] && params[2 * i + 1], "pws_netg_pack_weights: params[%d] is NULL", 2 * i);
        a.params[2 * i] = params[2 * i], a.params[2 * i + 1] = params[2 * i + 1];
        const PackLayer &p = a.layer[i];
        const bool sk = skip && skip[i];
        a.first_block[i] = nb;
        if (sk) ;
        else if (pack_tiled(p.kind)) nb += pack_tiles(p.kind, p.cin_pad, p.cout, p.k) + (unsigned)((p.cout + 255) / 256);   // tiles + bias blocks
        else nb += (unsigned)(((size_t)p.planes * p.cin_pad * p.cout + p.cout + 255) / 256);
        a.first_block_wino[i] = nbw;
        // (math == bf16: a layer with bf16 weights runs on them -- run_conv sets a.math whenever wb_off exists -- and never reads its Winograd copies)
        const bool skip_wino = sk || (math == PWS_MATH_BF16 && L[i].wb_off != (size_t)-1);
        if ((p.ww_off != kNoOff || p.wr_off != kNoOff) && !skip_wino) nbw += (unsigned)(((size_t)p.cin_pad * p.cout + 255) / 256);
        Bf16Layer &q = b.layer[i];
        q.planes = p.planes, q.krows = p.cin_pad, q.ncols = p.cout, q.kpad = (p.cin_pad + 31) / 32 * 32, q.npad = (p.cout + 63) / 64 * 64;
        q.src_off = p.w_off, q.dst_off = sk ? kNoOff : off32(L[i].wb_off);
        b.first_block[i] = nbb;
        if (q.dst_off != kNoOff && math != PWS_MATH_FP32) nbb += (unsigned)(q.planes * (q.kpad / 32) * (q.npad / 64));
    }
    a.total_blocks = nb, a.total_blocks_wino = nbw, b.total_blocks = nbb;
    int rc = launch_pack_all(a, packed, as_stream(stream));
    if (rc == PWS_OK && nbb) rc = launch_bf16_all(b, packed, packed, as_stream(stream));
    return rc;
}

extern "C" int pws_netg_pack_weights_for(const float *const *params, float *packed, int input_nc, int ngf, int math, pws_stream_t stream) {
    return pack_weights_impl(params, packed, input_nc, ngf, math, stream, nullptr);
}

static int pack_weights_dgrad_impl(const float *const *params, float *packed_dgrad, int input_nc, int ngf, pws_stream_t stream, const bool *skip) {
    PWS_REQUIRE(params && packed_dgrad, "pws_netg_pack_weights_dgrad: NULL pointer");
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_pack_weights_dgrad: bad ngf %d", ngf);
    size_t total = 0, dg = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total, &dg);
    PWS_REQUIRE(dg < 0xffffffffu, "pws_netg_pack_weights_dgrad: buffer too large for 32-bit offsets");
    PackAllArgs a{};
    Bf16AllArgs b{};
    a.nlayers = b.nlayers = L_COUNT;
    fill_pack_layers(L, a.layer);
    unsigned nb = 0, nbb = 0;
    for (int i = 0; i < L_COUNT; ++i) {
        a.params[2 * i] = params[2 * i], a.params[2 * i + 1] = nullptr;
        const PackLayer &p = a.layer[i];
        a.first_block_dgrad[i] = nb;
        Bf16Layer &q = b.layer[i];
        q = Bf16Layer{};
        q.dst_off = kNoOff;
        b.first_block[i] = nbb;
        if (p.dg_off == kNoOff || (skip && skip[i])) continue;
        PWS_REQUIRE(params[2 * i], "pws_netg_pack_weights_dgrad: params[%d] is NULL", 2 * i);
        if (pack_tiled(p.kind)) nb += pack_tiles(p.kind, p.cin, p.cout, p.k);
        else nb += (unsigned)(((size_t)p.dg_taps * p.cin * p.cout + 255) / 256);
        q.planes = p.dg_taps, q.krows = p.cout, q.ncols = p.cin, q.kpad = (p.cout + 31) / 32 * 32, q.npad = (p.cin + 63) / 64 * 64;
        q.src_off = p.dg_off, q.dst_off = off32(L[i].dgb_off);
        if (q.dst_off != kNoOff) nbb += (unsigned)(q.planes * (q.kpad / 32) * (q.npad / 64));
    }
    a.total_blocks_dgrad = nb, b.total_blocks = nbb;
    int rc = nb ? launch_dgrad_all(a, packed_dgrad, as_stream(stream)) : PWS_OK;
    if (rc == PWS_OK) rc = launch_bf16_all(b, packed_dgrad, packed_dgrad, as_stream(stream));
    return rc;
}

extern "C" int pws_netg_pack_weights_dgrad(const float *const *params, float *packed_dgrad, int input_nc, int ngf,
                                           pws_stream_t stream) {
    return pack_weights_dgrad_impl(params, packed_dgrad, input_nc, ngf, stream, nullptr);
}

// Both buffers of a TRAINING step at once.  math == PWS_MATH_BF16: the conv layers whose padded extents are the real ones (every one of them
// at ngf % 64 == 0) go torch layout -> bf16 forward copy + bf16 data-gradient copy in ONE pass (pack16_all_kernel: one read of the 194 MB,
// no fp32 packed copies -- nothing reads them in this mode: run_conv and the data-gradient calls take the bf16 copies whenever they
// exist, and at these channel counts the bf16 kernels never decline); the heads' small tensors and any layer that does not qualify take
// the ordinary path.  Otherwise: pws_netg_pack_weights_for + pws_netg_pack_weights_dgrad.
extern "C" int pws_netg_pack_weights_train(const float *const *params, float *packed, float *packed_dgrad, int input_nc, int ngf, int math,
                                           pws_stream_t stream) {
    PWS_REQUIRE(params && packed && packed_dgrad, "pws_netg_pack_weights_train: NULL pointer");
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_pack_weights_train: ngf must be a positive multiple of 16 (got %d)", ngf);
    bool fused[L_COUNT] = {};
    int nfused = 0;
    size_t total = 0, dg = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total, &dg);
    PWS_REQUIRE(total < 0xffffffffu && dg < 0xffffffffu, "pws_netg_pack_weights_train: buffers too large for 32-bit offsets");
    Pack16Args f{};
    f.nlayers = L_COUNT;
    fill_pack_layers(L, f.layer);
    unsigned nb = 0;
    for (int i = 0; i < L_COUNT; ++i) {
        PWS_REQUIRE(params[2 * i] && params[2 * i + 1], "pws_netg_pack_weights_train: params[%d] is NULL", 2 * i);
        f.params[2 * i] = params[2 * i], f.params[2 * i + 1] = params[2 * i + 1];
        const PackLayer &p = f.layer[i];
        const Layer &l = L[i];
        const bool has_dg = l.dg_off != (size_t)-1;
        bool ok = math == PWS_MATH_BF16 && g_experiment != 120 && pack_tiled(p.kind) && l.wb_off != (size_t)-1 && p.cout % 64 == 0 && p.cin_pad % 32 == 0;
        if (ok && has_dg) ok = l.dgb_off != (size_t)-1 && p.cin % 64 == 0;   // (cout % 64 == 0 covers the data-gradient copy's k padding)
        // every source of the layer must be a multiple of 32 channels, or the bf16 kernels decline and fall back to the fp32 copies:
        // sources are sums of ngf multiples, ngf % 32 == 0 settles it
        ok = ok && ngf % 32 == 0;
        // the first layer reads the caller's NCHW window: run_conv takes its bf16 kernel only for windows of <= 32 channels (the NHWC-32 copy);
        // a wider window goes to the fp32 first-layer kernels, which read the fp32 packed copy -- so that copy must exist
        if (i == L_TRANSFER) ok = ok && input_nc <= 32;
        fused[i] = ok, nfused += ok ? 1 : 0;
        f.first_block[i] = nb;
        f.wb_off[i] = ok ? off32(l.wb_off) : kNoOff;
        f.dgb_off[i] = ok && has_dg ? off32(l.dgb_off) : kNoOff;
        if (ok) nb += pack_tiles(p.kind, p.cin_pad, p.cout, p.k) + (unsigned)((p.cout + 255) / 256);
    }
    f.total_blocks = nb;
    int rc = nfused ? launch_pack16_all(f, packed, packed_dgrad, as_stream(stream)) : PWS_OK;
    if (rc == PWS_OK) rc = pack_weights_impl(params, packed, input_nc, ngf, math, stream, nfused ? fused : nullptr);
    if (rc == PWS_OK) rc = pack_weights_dgrad_impl(params, packed_dgrad, input_nc, ngf, stream, nfused ? fused : nullptr);
    return rc;
}

extern "C" int pws_netg_backward_part(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                                      int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                                      const float *g_grids, const float *g_resid, float *dpacked, int part, int nparts,
                                      unsigned char *final_mask, pws_stream_t stream) {
    return pws_netg_backward_opts(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, ws, ws_bytes, resid, thetas, g_grids, g_resid,
                                  dpacked, part, nparts, final_mask, nullptr, stream);
}

// ---- use_BN=True training (BatchNorm2d after every conv, batch statistics): fp32 math and storage only
extern "C" size_t pws_netg_bn_floats(int input_nc, int ngf) {
    if (input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0, bn = 0;
    for (const Layer &l : build_layers(input_nc, ngf, &total)) bn += 2 * (size_t)l.cout;
    return bn;
}

extern "C" size_t pws_netg_train_workspace_bytes_bn(int n, int input_nc, int ngf) {
    if (n <= 0 || input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t used = 0;
    BnCfg bn;
    const NetgOpts o{PWS_MATH_FP32, PWS_STORE_FP32, false};
    run_backward(nullptr, nullptr, nullptr, n, input_nc, ngf, 0, nullptr, 0, nullptr, nullptr, UpGrads(), nullptr, nullptr, true,
                 &used, 0, 1, nullptr, &bn, nullptr, &o);
    return used;
}

// mode of the BatchNorm training path: fp32 storage always (the BatchNorm kernels are fp32); the conv contractions in fp32 or on the
// bf16 matrix cores (operands rounded while they are staged, fp32 accumulation), as pws_netg_opts.math says
static int bn_opts_from(const pws_netg_opts *opts, NetgOpts *o) {
    *o = NetgOpts{PWS_MATH_FP32, PWS_STORE_FP32, g_two_queues};
    if (!opts) return PWS_OK;
    NetgOpts u;
    if (int rc = opts_from(opts, &u)) return rc;
    PWS_REQUIRE(u.store == PWS_STORE_FP32 && u.x_sample_stride == 0,
                "pws_netg_*_bn_opts: the BatchNorm training path stores fp32 activations (store must be PWS_STORE_FP32) and reads a dense window");
    o->math = u.math, o->two_queues = u.two_queues, o->deterministic = u.deterministic;
    return PWS_OK;
}

extern "C" int pws_netg_forward_bn(const float *packed, const float *bn_params, float *bn_running, float momentum, float eps,
                                   const float *x, int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes,
                                   float *grids, float *resid, float *thetas, pws_stream_t stream) {
    return pws_netg_forward_bn_opts(packed, bn_params, bn_running, momentum, eps, x, n, input_nc, ngf, align_corners, ws, ws_bytes, grids, resid,
                                    thetas, nullptr, stream);
}

extern "C" int pws_netg_forward_bn_opts(const float *packed, const float *bn_params, float *bn_running, float momentum, float eps,
                                        const float *x, int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes,
                                        float *grids, float *resid, float *thetas, const pws_netg_opts *opts, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_forward_bn: bad n/input_nc/ngf %d/%d/%d", n, input_nc, ngf);
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(n >= 2, "pws_netg_forward_bn: training-mode BatchNorm needs more than 1 value per channel (the theta head has one per "
                        "sample): n >= 2, as torch");
    PWS_REQUIRE(packed && bn_params && x && ws && grids && resid && thetas, "pws_netg_forward_bn: NULL pointer");
    PWS_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "pws_netg_forward_bn: workspace must be 256-byte aligned");
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    BnCfg bn;
    bn.params = bn_params, bn.running = bn_running, bn.momentum = momentum, bn.eps = eps;
    NetgOpts o;   // opts == NULL: fp32 whatever the process defaults say
    if (int rc = bn_opts_from(opts, &o)) return rc;
    Exec E(packed, L, n, static_cast<char *>(ws), ws_bytes, as_stream(stream), false, true, o, &bn);
    forward_graph(E, x, n, input_nc, ngf, 1, align_corners, grids, resid, thetas);
    return E.rc();
}

extern "C" int pws_netg_backward_bn(const float *packed, const float *packed_dgrad, const float *bn_params, float eps, const float *x,
                                    int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, const float *resid,
                                    const float *thetas, const float *g_grids, const float *g_resid, float *dpacked, float *dbn,
                                    pws_stream_t stream) {
    return pws_netg_backward_bn_opts(packed, packed_dgrad, bn_params, eps, x, n, input_nc, ngf, align_corners, ws, ws_bytes, resid, thetas, g_grids,
                                     g_resid, dpacked, dbn, nullptr, stream);
}

extern "C" int pws_netg_backward_bn_opts(const float *packed, const float *packed_dgrad, const float *bn_params, float eps, const float *x,
                                         int n, int input_nc, int ngf, int align_corners, void *ws, size_t ws_bytes, const float *resid,
                                         const float *thetas, const float *g_grids, const float *g_resid, float *dpacked, float *dbn,
                                         const pws_netg_opts *opts, pws_stream_t stream) {
    PWS_REQUIRE(n >= 2 && input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_backward_bn: bad n/input_nc/ngf");
    PWS_REQUIRE(packed && packed_dgrad && bn_params && x && ws && resid && thetas && dpacked && dbn, "pws_netg_backward_bn: NULL pointer");
    PWS_REQUIRE(g_grids || g_resid, "pws_netg_backward_bn: no output gradient given");
    PWS_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "pws_netg_backward_bn: workspace must be 256-byte aligned");
    BnCfg bn;
    bn.params = bn_params, bn.eps = eps;
    NetgOpts o;
    if (int rc = bn_opts_from(opts, &o)) return rc;
    return run_backward(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, static_cast<char *>(ws), ws_bytes, resid, thetas,
                        UpGrads(g_grids, g_resid, (size_t)n * 256 * 256 * 2), dpacked, as_stream(stream), false, nullptr, 0, 1, nullptr, &bn, dbn, &o);
}

extern "C" int pws_netg_unpack_grads(const float *dpacked, float *const *grads, int input_nc, int ngf, pws_stream_t stream) {
    PWS_REQUIRE(dpacked && grads, "pws_netg_unpack_grads: NULL pointer");
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_unpack_grads: bad ngf %d", ngf);
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    UnpackAllArgs a{};
    a.nlayers = L_COUNT;
    fill_pack_layers(L, a.layer);
    for (int i = 0; i < L_COUNT; ++i) a.layer[i].w_off = off32(L[i].gw_off), a.layer[i].b_off = off32(L[i].gb_off);   // the gradient slab's layout
    unsigned nb = 0;
    for (int i = 0; i < L_COUNT; ++i) {
        // a layer whose two pointers are both NULL is skipped (part-wise unpacking, pws_netg_backward_part)
        PWS_REQUIRE((grads[2 * i] != nullptr) == (grads[2 * i + 1] != nullptr), "pws_netg_unpack_grads: grads[%d], grads[%d]: one is NULL",
                    2 * i, 2 * i + 1);
        a.grads[2 * i] = grads[2 * i], a.grads[2 * i + 1] = grads[2 * i + 1];
        const PackLayer &p = a.layer[i];
        a.first_block[i] = nb;
        if (grads[2 * i] && pack_tiled(p.kind)) nb += pack_tiles(p.kind, p.cin, p.cout, p.k) + (unsigned)((p.cout + 255) / 256);
        else if (grads[2 * i]) nb += (unsigned)(((size_t)p.k * p.k * p.cin * p.cout + p.cout + 255) / 256);
    }
    a.total_blocks = nb;
    if (nb == 0) return PWS_OK;
    return launch_unpack_all(a, dpacked, as_stream(stream));  // one launch: up to 46 weights + 46 biases
}

extern "C" size_t pws_netg_workspace_bytes(int n, int input_nc, int ngf, int is_training) {
    if (n <= 0 || input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    Exec E(nullptr, L, n, nullptr, 0, nullptr, /*dry=*/true, false, NetgOpts{PWS_MATH_FP32, PWS_STORE_FP32, false});   // the arena layout does not depend on the mode
    forward_graph(E, nullptr, n, input_nc, ngf, is_training, 0, nullptr, nullptr, nullptr);
    return E.used();
}

// HOST only (no GPU call): the run of an nparts-run backward after which layer i's gradient is final -- the same rule
// run_backward applies to its final_mask (a layer is final once every tape op that uses it lies at a reversed position < r_end).
extern "C" int pws_netg_backward_plan(int input_nc, int ngf, int nparts, unsigned char *final_part) {
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_backward_plan: bad input_nc / ngf %d / %d", input_nc, ngf);
    PWS_REQUIRE(nparts >= 1 && nparts <= 255 && final_part, "pws_netg_backward_plan: nparts %d (1..255) / NULL pointer", nparts);
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    Exec E(nullptr, L, 1, nullptr, 0, nullptr, /*dry=*/true, false, NetgOpts{PWS_MATH_FP32, PWS_STORE_FP32, false});
    forward_graph(E, nullptr, 1, input_nc, ngf, 1, 0, nullptr, nullptr, nullptr, /*caller_thetas=*/true);
    const size_t T = E.tape().size();
    size_t last[L_COUNT];   // largest reversed position of an op that uses the layer
    bool used[L_COUNT];
    for (int i = 0; i < L_COUNT; ++i) last[i] = 0, used[i] = false;
    auto touch = [&](int layer, size_t rpos) { last[layer] = used[layer] ? std::max(last[layer], rpos) : rpos, used[layer] = true; };
    for (size_t ii = 0; ii < T; ++ii) {
        const Op &op = E.tape()[ii];
        const size_t rpos = T - 1 - ii;
        if (op.type == OP_FIELD) touch(L_OUT, rpos);
        else if (op.type == OP_THETA) touch(L_FLATTEN, rpos), touch(L_LINEAR, rpos);
        else touch(op.layer, rpos);
    }
    for (int i = 0; i < L_COUNT; ++i) {
        int part = 0;
        while (used[i] && part < nparts - 1 && !(last[i] < T * (size_t)(part + 1) / nparts)) ++part;
        final_part[i] = (unsigned char)part;
    }
    return PWS_OK;
}

extern "C" size_t pws_netg_train_workspace_bytes(int n, int input_nc, int ngf) {
    if (n <= 0 || input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t used = 0;
    const NetgOpts o{PWS_MATH_FP32, PWS_STORE_FP32, false};   // the arena layout does not depend on the mode
    run_backward(nullptr, nullptr, nullptr, n, input_nc, ngf, 0, nullptr, 0, nullptr, nullptr, UpGrads(), nullptr, nullptr,
                 true, &used, 0, 1, nullptr, nullptr, nullptr, &o);
    return used;
}

extern "C" int pws_netg_forward_opts(const float *packed, const float *x, int n, int input_nc, int ngf, int is_training,
                                     int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid, float *thetas,
                                     const pws_netg_opts *opts, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_forward: bad n/input_nc/ngf %d/%d/%d", n, input_nc,
                ngf);
    NetgOpts o;
    if (int rc = opts_from(opts, &o)) return rc;
    if (n == 0) return PWS_OK;
    PWS_REQUIRE(packed && x && ws && grids, "pws_netg_forward: NULL pointer");
    PWS_REQUIRE(!is_training || (resid && thetas), "pws_netg_forward: resid and thetas must be given when is_training");
    PWS_REQUIRE(o.x_sample_stride == 0 || (!is_training && o.x_sample_stride % 4 == 0 && o.x_sample_stride < (1u << 30)),
                "pws_netg_opts: x_sample_stride %zu: inference forward only, a multiple of 4 floats", o.x_sample_stride);
    PWS_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "pws_netg_forward: workspace must be 256-byte aligned");
    size_t total = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total);
    Exec E(packed, L, n, static_cast<char *>(ws), ws_bytes, as_stream(stream), false, true, o);
    forward_graph(E, x, n, input_nc, ngf, is_training, align_corners, grids, resid, thetas);
    return E.rc();
}

extern "C" int pws_netg_forward(const float *packed, const float *x, int n, int input_nc, int ngf, int is_training,
                                int align_corners, void *ws, size_t ws_bytes, float *grids, float *resid, float *thetas,
                                pws_stream_t stream) {
    return pws_netg_forward_opts(packed, x, n, input_nc, ngf, is_training, align_corners, ws, ws_bytes, grids, resid, thetas, nullptr,
                                 stream);
}

static int backward_entry(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf, int align_corners,
                          void *ws, size_t ws_bytes, const float *resid, const float *thetas, const UpGrads &up, float *dpacked, int part,
                          int nparts, unsigned char *final_mask, const pws_netg_opts *opts, pws_stream_t stream) {
    PWS_REQUIRE(n >= 0 && input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_backward: bad n/input_nc/ngf");
    PWS_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "pws_netg_backward: part %d of %d", part, nparts);
    NetgOpts o;
    if (int rc = opts_from(opts, &o)) return rc;
    PWS_REQUIRE(o.x_sample_stride == 0, "pws_netg_backward: x_sample_stride is an option of the inference forward (the backward reads a dense window)");
    if (n == 0) {
        if (final_mask)
            for (int i = 0; i < L_COUNT; ++i) final_mask[i] = 1;
        return PWS_OK;
    }
    PWS_REQUIRE(packed && packed_dgrad && x && ws && resid && thetas && dpacked, "pws_netg_backward: NULL pointer");
    PWS_REQUIRE(up.any(), "pws_netg_backward: no output gradient given");
    PWS_REQUIRE((reinterpret_cast<size_t>(ws) & 255) == 0, "pws_netg_backward: workspace must be 256-byte aligned");
    return run_backward(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, static_cast<char *>(ws), ws_bytes, resid, thetas, up,
                        dpacked, as_stream(stream), false, nullptr, part, nparts, final_mask, nullptr, nullptr, &o);
}

extern "C" int pws_netg_backward_opts(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                                      int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                                      const float *g_grids, const float *g_resid, float *dpacked, int part, int nparts,
                                      unsigned char *final_mask, const pws_netg_opts *opts, pws_stream_t stream) {
    return backward_entry(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, ws, ws_bytes, resid, thetas,
                          UpGrads(g_grids, g_resid, (size_t)(n > 0 ? n : 0) * 256 * 256 * 2), dpacked, part, nparts, final_mask, opts, stream);
}

extern "C" int pws_netg_backward_lists(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                                       int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                                       const float *const *g_grids, const float *const *g_resid, float *dpacked, int part, int nparts,
                                       unsigned char *final_mask, const pws_netg_opts *opts, pws_stream_t stream) {
    return backward_entry(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, ws, ws_bytes, resid, thetas, UpGrads(g_grids, g_resid),
                          dpacked, part, nparts, final_mask, opts, stream);
}

extern "C" size_t pws_netg_grad_floats(int input_nc, int ngf) {
    if (input_nc <= 0 || ngf <= 0 || ngf % 16 != 0) return 0;
    size_t total = 0, total_grad = 0;
    build_layers(input_nc, ngf, &total, nullptr, &total_grad);
    return total_grad;
}

extern "C" int pws_netg_grad_layout(int input_nc, int ngf, size_t *first_float, size_t *floats) {
    PWS_REQUIRE(input_nc > 0 && ngf > 0 && ngf % 16 == 0, "pws_netg_grad_layout: bad ngf %d", ngf);
    PWS_REQUIRE(first_float && floats, "pws_netg_grad_layout: NULL pointer");
    size_t total = 0, total_grad = 0;
    const std::vector<Layer> L = build_layers(input_nc, ngf, &total, nullptr, &total_grad);
    for (int i = 0; i < L_COUNT; ++i) {
        first_float[i] = L[i].gw_off;
        floats[i] = (i + 1 < L_COUNT ? L[i + 1].gw_off : total_grad) - L[i].gw_off;   // weight, bias and their padding: adjacent layers abut
    }
    return PWS_OK;
}

extern "C" int pws_netg_backward(const float *packed, const float *packed_dgrad, const float *x, int n, int input_nc, int ngf,
                                 int align_corners, void *ws, size_t ws_bytes, const float *resid, const float *thetas,
                                 const float *g_grids, const float *g_resid, float *dpacked, pws_stream_t stream) {
    return pws_netg_backward_opts(packed, packed_dgrad, x, n, input_nc, ngf, align_corners, ws, ws_bytes, resid, thetas, g_grids, g_resid,
                                  dpacked, 0, 1, nullptr, nullptr, stream);
}

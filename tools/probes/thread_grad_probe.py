#!/usr/bin/env python3
"""tests/test_hip_threads.py::test_train_step_beside_inference_and_another_train_step[bf16] failed 1 run in 10 (gradient of thread 1 differs
from the serial run by 3e-4 of its size).  Repeats the threaded part in ONE process and, on a mismatch, lists WHICH tensors differ.
usage: python tools/probes/thread_grad_probe.py [rounds] [math] [variant]   variant: all | noinfer | one"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_threads as T  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402

# ---- stash of intermediate gradients, keyed by a tag the worker sets on its own thread before it runs (PROBE_STASH=1)
import threading  # noqa: E402

from pwstablenet_amd import autograd as AG  # noqa: E402
from pwstablenet_amd import objective as OB  # noqa: E402
_tls = threading.local()
STASH = {}
if os.environ.get("PROBE_STASH") == "1":
    _of, _ob = OB._Objective.forward, OB._Objective.backward
    _nf, _nb = AG._NetGTrain.forward, AG._NetGTrain.backward

    def of(ctx, *a):
        ctx._tag = getattr(_tls, "tag", None)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        return _of(ctx, *a)

    import ctypes

    def ob(ctx, *g):
        # the objective's backward, stage 0 and 1 once more by hand BEFORE the real one, every intermediate kept
        st = STASH.setdefault(ctx._tag, {})
        cfg, sv = ctx.cfg, ctx.saved_tensors
        L_ = cfg["num_layer"]
        rgb, stable, features, theta, resid = sv[:5]
        grids, fakes = sv[5:5 + L_], sv[5 + L_:5 + 2 * L_]
        m, h, w = grids[0].shape[0], cfg["size"], cfg["size"]
        n = m // 2
        lib, stq = A.lib(), A.current_stream()
        scale = g[0].reshape(1).to(torch.float32).contiguous()
        cnt = float(n) * 3 * h * w
        c_l1, c_t = 1.0 / cnt, cfg["lamd"] / cnt
        c_f = 1.0 / (cfg["number_feature"] * cfg["batch"])
        rgb_p, rgb_s = OB._planes(rgb, h, w, "rgb")
        stb_p, stb_s = OB._planes(stable, h, w, "stable")
        inter = {"scale": scale.clone(), "theta": theta.clone(), "features": features.clone(), "rgb": rgb.clone(), "stable": stable[:, :3].clone()}
        for nl in range(2):
            inter["fake%d" % nl], inter["grid%d" % nl] = fakes[nl].clone(), grids[nl].clone()
            gextra = torch.zeros_like(fakes[nl])
            scratch = torch.empty((n, 3, h, w), device=rgb.device, dtype=torch.float32)
            A.check(lib.pws_temporal_l1_bwd_det(A.ptr(fakes[nl][:n]), A.ptr(fakes[nl][n:]), A.ptr(theta), c_t, A.ptr(scale),
                                                A.ptr(gextra[:n]), A.ptr(gextra[n:]), A.ptr(scratch), n, h, w, stq), "t")
            inter["gextra%d" % nl], inter["scratch%d" % nl] = gextra.clone(), scratch.clone()
            gg = torch.empty_like(grids[nl])
            A.check(lib.pws_warp_norm_bwd(rgb_p, rgb_s, A.ptr(grids[nl]), stb_p, stb_s, c_l1, A.ptr(scale), A.ptr(gextra), A.ptr(gg), 0, m, h, w, stq), "w")
            inter["gg_warp%d" % nl] = gg.clone()
            A.check(lib.pws_feature_loss_bwd_det(A.ptr(grids[nl]), A.ptr(features), c_f, A.ptr(scale), A.ptr(gg), m, features.shape[1], h, w, stq), "f")
            inter["gg_feat%d" % nl] = gg.clone()
        st["inter"] = inter
        res = _ob(ctx, *g)
        st["obj_stream_ok"] = torch.cuda.current_stream().cuda_stream == ctx._stream
        st["obj_out"] = [None if r is None else r.clone() for r in res]
        st["obj_gloss"] = None if g[0] is None else g[0].clone()
        return res

    def nf(ctx, *a):
        ctx._tag = getattr(_tls, "tag", None)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        return _nf(ctx, *a)

    def nb(ctx, *g):
        st = STASH.setdefault(ctx._tag, {})
        st["netg_stream_ok"] = torch.cuda.current_stream().cuda_stream == ctx._stream
        st["netg_in"] = [None if r is None else r.clone() for r in g]
        return _nb(ctx, *g)
    OB._Objective.forward, OB._Objective.backward = staticmethod(of), staticmethod(ob)
    AG._NetGTrain.forward, AG._NetGTrain.backward = staticmethod(nf), staticmethod(nb)


def tagged(fn, tag):
    def run():
        _tls.tag = tag
        return fn()
    return run


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
math = sys.argv[2] if len(sys.argv) > 2 else "bf16"
variant = sys.argv[3] if len(sys.argv) > 3 else "all"
A.lib().pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
_tq = False if os.environ.get("PROBE_CU_MASK") == "1" else None
works = [T._train_work("W1", 11, math, 2, two_queues=_tq), T._train_work("W2", 12, math, 2, two_queues=_tq)]
infer_net = T.make_net("W1", 5)
x = [torch.from_numpy(synth.noise_window(2, 31, 256, seed=33 + r)).cuda() for r in range(2)]
infer = T._infer_work(infer_net, x, 8)
works = [tagged(works[0], 0), tagged(works[1], 1)]
serial = [w() for w in works] + [infer()]
torch.cuda.synchronize()
SERIAL_STASH = {k: dict(v) for k, v in STASH.items()}
serial2 = [w() for w in works]
for t in range(2):
    assert all(torch.equal(a, b) for a, b in zip(serial[t][1], serial2[t][1])), "serial runs differ from each other"
torch.cuda.synchronize()
if os.environ.get("PROBE_EAGER_INFER") != "1":
    T._capture_serially(infer_net, x)
if os.environ.get("PROBE_TWO_QUEUES") is not None:
    A.lib().pws_set_option(A.OPT_TWO_QUEUES, int(os.environ["PROBE_TWO_QUEUES"]))
names = [k for k, _ in synth.make_weights("W1", seed=1, ngf=T.NGF)]
bad = 0
for r in range(rounds):
    def dummy_gpu():   # a third thread that only keeps the GPU busy with torch kernels on its own stream
        z = torch.randn(4096, 4096, device="cuda")
        for _ in range(60):
            z = (z @ z).tanh_()
        torch.cuda.current_stream().synchronize()
        return None

    def dummy_copy():   # ... with memory traffic only
        z = torch.randn(64 << 20, device="cuda")
        for _ in range(300):
            z = z + 1.0
        torch.cuda.current_stream().synchronize()
        return None

    def dummy_host():   # ... that only competes for the interpreter lock
        import time
        t = time.perf_counter()
        k = 0
        while time.perf_counter() - t < 0.25:
            k += 1
        return None
    third = {"all": infer, "dummy": dummy_gpu, "copy": dummy_copy, "host": dummy_host}.get(variant)
    fns = works + [third] if third is not None else (works if variant == "noinfer" else [works[1], infer])
    streams = None
    if os.environ.get("PROBE_CU_MASK") == "1":   # every thread on its own share of the compute units (and one queue per call: see below)
        streams = A.cu_masked_streams([96, 96, 64][:len(fns)])
    got = T._in_threads(fns, streams=streams)
    torch.cuda.synchronize()
    idx = [0, 1] if variant != "one" else [1]
    for j, t in enumerate(idx):
        g = got[j]
        lossdiff = float((g[0] - serial[t][0]).abs().max())
        diffs = [(i, float((a - b).abs().max()), float(b.abs().max())) for i, (a, b) in enumerate(zip(g[1], serial[t][1])) if not torch.equal(a, b)]
        if (diffs or lossdiff) and SERIAL_STASH:
            ref, cur = SERIAL_STASH[t], STASH[t]
            print("   stash: objective backward on its forward's stream: %s, generator backward: %s" % (cur.get("obj_stream_ok"), cur.get("netg_stream_ok")))
            for key in sorted(cur.get("inter", {})):
                a_, b_ = cur["inter"][key], ref["inter"][key]
                if not torch.equal(a_, b_):
                    print("   inter %-10s differs: max %.4g of %.4g, %d elements" % (key, float((a_ - b_).abs().max()), float(b_.abs().max()), int((a_ != b_).sum())))
            for key in ("obj_out", "netg_in"):
                for i, (a_, b_) in enumerate(zip(cur[key], ref[key])):
                    if a_ is not None and not torch.equal(a_, b_):
                        print("   stash %s[%d] differs: max %.4g of %.4g" % (key, i, float((a_ - b_).abs().max()), float(b_.abs().max())))
        if diffs or lossdiff:
            bad += 1
            print("round %d thread %d: loss diff %.3g; %d of %d gradient tensors differ" % (r, t, lossdiff, len(diffs), len(g[1])))
            for i, d, m in diffs[:100]:
                print("     %2d %-40s max diff %.4g of max %.4g (%.2g)" % (i, names[i] if i < len(names) else "?", d, m, d / (m + 1e-30)))
print("%d mismatching (round, thread) pairs in %d rounds [%s, %s]" % (bad, rounds, math, variant))

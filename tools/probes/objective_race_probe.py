#!/usr/bin/env python3
"""Is the deterministic objective backward reproducible while OTHER streams keep the GPU busy?  (tools/probes/thread_grad_probe.py: with
three host threads the gradient the objective hands to stage 1 or 2 differs from the serial run in 1 step of 10.)
Thread 1: StabObjective forward + backward (deterministic) on fixed inputs, over and over, every result compared with the first.
Thread 2: bf16 training steps of a generator (the contention).  Thread 3 (optional): spins on the interpreter lock.
usage: python tools/probes/objective_race_probe.py [iterations] [contention: train|none] [spin: 0|1]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_threads as T  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.objective import StabObjective, u8_normalize  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402
A.lib().pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
if os.environ.get("PROBE_TWO_QUEUES") is not None:
    A.lib().pws_set_option(A.OPT_TWO_QUEUES, int(os.environ["PROBE_TWO_QUEUES"]))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
contention = sys.argv[2] if len(sys.argv) > 2 else "train_bf16"   # train_bf16 | train_fp32 | infer_bf16 | infer_fp32 | none
spin = len(sys.argv) > 3 and sys.argv[3] == "1"
items = 2
batch = [torch.from_numpy(t).cuda() for t in synth.make_train_batch(items, seed=12)]
images1, features1, _a1, images2, features2, _a2, feature_adjacent = batch
n, period = items, 30
rest = torch.empty((2 * n, images1.shape[1] - period - 1, 256, 256), device="cuda")
for half, img in enumerate((images1, images2)):
    u8_normalize(img[:, period + 1:], rest[half * n:(half + 1) * n])
features = torch.cat([features1, features2], 0).float()
rs = torch.Generator(device="cuda").manual_seed(5)
base = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 256, device="cuda"), torch.linspace(-1, 1, 256, device="cuda"), indexing="ij")[::-1], -1)
grids0 = [(base[None] + 0.05 * torch.randn(2 * n, 256, 256, 2, device="cuda", generator=rs)).contiguous() for _ in range(3)]
resid0 = [(0.02 * torch.randn(2 * n, 256, 256, 2, device="cuda", generator=rs)).contiguous() for _ in range(3)]
obj = StabObjective(batchSize=items)
stop = threading.Event()
result = {}


def objective_loop():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    bad = 0
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.default_stream())
        ref = None
        for it in range(iters):
            grids = [g.clone().requires_grad_(True) for g in grids0]
            resid = [r.clone().requires_grad_(True) for r in resid0]
            out = obj(grids, resid, rest[:, 0:3], rest[:, 3:], features, feature_adjacent, deterministic=True)
            out.loss_g.backward()
            s.synchronize()
            cur = [out[k].detach().clone() for k in ("loss_g", "loss_mse", "loss_feature", "loss_delta", "loss_g2", "loss_pixel")] + [g.grad.clone() for g in grids] + [resid[2].grad.clone()]
            if ref is None:
                ref = cur
                continue
            d = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
            if d:
                bad += 1
                print("iteration %d: entries %s differ (0-5 losses, 6-8 ggrids, 9 gresid): %s" % (
                    it, d, ["%.3g" % float((cur[i] - ref[i]).abs().max()) for i in d]), flush=True)
    result["bad"] = bad
    stop.set()


def train_loop():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    kind, math = contention.split("_")
    with torch.cuda.stream(s):
        if kind == "train":
            w = T._train_work("W1", 11, math, 2)
            while not stop.is_set():
                w()
        elif kind == "layer":   # ONE kind of launch, over and over (math = the layer: k5 | k3 | k3f32 | ct4 | pad | cat)
            import test_hip_bf16 as B
            if math == "pad":
                x = torch.randn(4, 31, 256, 256, device="cuda")
                out = torch.empty(4, 256, 256, 32, device="cuda")
                while not stop.is_set():
                    A.check(A.lib().pws_nchw_to_nhwc_pad(A.ptr(x), A.ptr(out), 4, 31, 256, 256, 32, A.current_stream()), "pad")
                    s.synchronize()
            elif math == "cat":   # torch kernels only
                x = torch.randn(4, 64, 256, 256, device="cuda")
                while not stop.is_set():
                    y = torch.cat([x, x], 1).mul_(0.5)
                    del y
                    s.synchronize()
            else:
                kname, shape, src_c, cout, store, f32 = {"k5": ("CONV_K5S1", (4, 64, 64), [32], 64, True, False), "k3": ("CONV_K3S1", (4, 64, 64), [64], 64, True, False),
                                                         "ct4": ("CONVT_K4S2", (4, 32, 32), [64], 64, True, False), "k3f32": ("CONV_K3S1", (4, 64, 64), [64], 64, False, True)}[math]
                x, wt, b, _ = B.make_case(kname, shape, src_c, cout, "probe")
                while not stop.is_set():
                    if f32:
                        import test_hip_ops as O
                        raise SystemExit("k3f32: not wired")
                    B.hip_fwd(A, kname, x, wt, b, 1, src_c, cout, 64, store=store)
        else:   # inference only: no autograd, nothing on the engine's worker thread
            net = T.make_net("W1", 5, ngf=int(os.environ.get("PROBE_NGF", "32")))
            net.module.set_math(math, store=os.environ.get("PROBE_STORE"))
            xs = [torch.from_numpy(synth.noise_window(4, 31, 256, seed=33 + r)).cuda() for r in range(2)]
            k = 0
            with torch.no_grad():
                while not stop.is_set():
                    net(xs[k & 1], False)
                    k += 1
                    if k % 16 == 0:
                        s.synchronize()


def spin_loop():
    k = 0
    while not stop.is_set():
        k += 1


ths = [threading.Thread(target=objective_loop)]
if contention != "none":
    ths.append(threading.Thread(target=train_loop))
if spin:
    ths.append(threading.Thread(target=spin_loop))
t0 = time.time()
for t in ths:
    t.start()
for t in ths:
    t.join()
print("%d of %d iterations differ from the first [contention %s, spin %s, %.0f s]" % (result.get("bad", -1), iters - 1, contention, spin, time.time() - t0))

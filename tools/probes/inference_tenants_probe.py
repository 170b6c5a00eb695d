#!/usr/bin/env python3
"""Two inference tenants on one GPU (the serving case of BASELINE configs[4] with more than one stream per device): tenant A's results on fixed
inputs, every call compared with its first, while tenant B runs bf16 inference of another generator on another stream.  (A wave executing the
gfx950 bf16 matrix instructions disturbs SOME kernels of other streams on its CUs -- DESIGN.md section 10: are the inference kernels among them?)
usage: python tools/probes/inference_tenants_probe.py [seconds]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_threads as T  # noqa: E402
from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
stop, started = threading.Event(), threading.Event()


def tenant_b():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net = T.make_net("W1", 5, ngf=64)
        net.module.set_math("bf16")
        x = torch.from_numpy(synth.noise_window(8, 31, 256, seed=77)).cuda()
        k = 0
        with torch.no_grad():
            while not stop.is_set():
                net(x, False)
                k += 1
                if k == 4:
                    started.set()
                if k % 16 == 0:
                    s.synchronize()


def tenant_a():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        frames = torch.rand(8, 3, 256, 256, device="cuda") * 2 - 1
        x = torch.from_numpy(synth.noise_window(8, 31, 256, seed=11)).cuda()
        for math, graph in (("fp32", False), ("fp32", True), ("bf16", False), ("bf16", True)):
            net = T.make_net("W2", 9, ngf=64)
            net.module.set_math(math)
            net.module.enable_graph(graph)
            with torch.no_grad():
                ref = PF.grid_sample(frames, net(x, False)).clone()
                s.synchronize()
                started.wait(120)
                t_end, it, bad = time.perf_counter() + secs, 0, 0
                while time.perf_counter() < t_end:
                    out = PF.grid_sample(frames, net(x, False))
                    bad += 0 if torch.equal(out, ref) else 1
                    it += 1
            print("tenant A %s inference%s + grid_sample beside bf16 inference of tenant B: %d of %d calls differ from the first" % (
                math, " (graph)" if graph else "", bad, it), flush=True)
    stop.set()


ths = [threading.Thread(target=tenant_a), threading.Thread(target=tenant_b)]
for t in ths:
    t.start()
for t in ths:
    t.join()

#!/usr/bin/env python3
"""Control for tools/probes/objective_race_probe.py WITHOUT this library: thread 1 runs a fixed chain of torch kernels on freshly allocated
temporaries (zeros_like / empty_like / elementwise / gather-like indexing) on its own stream and compares every result with the first;
thread 2 keeps the GPU busy with torch kernels on another stream; thread 3 spins on the interpreter lock.
usage: python tools/probes/torch_only_race_probe.py [iterations] [busy: 0|1] [spin: 0|1]"""
import sys
import threading
import time

import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
busy = not (len(sys.argv) > 2 and sys.argv[2] == "0")
spin = not (len(sys.argv) > 3 and sys.argv[3] == "0")
torch.manual_seed(0)
x = torch.randn(4, 3, 256, 256, device="cuda")
g = torch.randn(4, 256, 256, 2, device="cuda") * 0.1
idx = torch.randint(0, 256 * 256, (4, 3, 256 * 256), device="cuda")
stop = threading.Event()
started = threading.Event()
res = {}


def chain():
    ge = torch.zeros_like(x)
    ge[:2] -= torch.sign(x[:2] - x[2:])
    sc = torch.empty_like(x[:2])
    sc.copy_(torch.sign(x[2:] - x[:2]))
    ge[2:] += torch.gather(sc.reshape(2, 3, -1), 2, idx[:2]).reshape(2, 3, 256, 256)
    gg = torch.empty_like(g)
    gg.copy_((ge.sum(1)[..., None] * g).contiguous())
    gg[:, ::7, ::5] += 0.25
    return gg


def loop():
    s = torch.cuda.Stream()
    bad = 0
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.default_stream())
        ref = chain().clone()
        if busy:
            started.wait(120)
        for it in range(iters):
            cur = chain()
            if not torch.equal(cur, ref):
                bad += 1
                print("iteration %d: %d elements differ" % (it, int((cur != ref).sum())), flush=True)
    res["bad"] = bad
    stop.set()


def busy_loop():
    s = torch.cuda.Stream()
    if len(sys.argv) > 4:   # the aggressor is this library's inference in the given math mode (fp32 | bf16); the victim stays torch-only
        import os
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sys.path.insert(0, root), sys.path.insert(0, os.path.join(root, "tests"))
        import test_hip_threads as T
        from pwstablenet_amd import synth
        with torch.cuda.stream(s):
            net = T.make_net("W1", 5)
            net.module.set_math(sys.argv[4])
            xs = [torch.from_numpy(synth.noise_window(4, 31, 256, seed=33 + r)).cuda() for r in range(2)]
            k = 0
            with torch.no_grad():
                while not stop.is_set():
                    net(xs[k & 1], False)
                    k += 1
                    if k == 4:
                        started.set()
                    if k % 16 == 0:
                        s.synchronize()
        return
    with torch.cuda.stream(s):
        z = torch.randn(2048, 2048, device="cuda")
        y = torch.randn(64 << 20, device="cuda")
        started.set()
        while not stop.is_set():
            z = (z @ z).tanh_()
            y = y * 1.0001 + 0.5
            t = torch.zeros_like(y)
            t += y
            del t
        s.synchronize()


def spin_loop():
    k = 0
    while not stop.is_set():
        k += 1


ths = [threading.Thread(target=loop)]
if busy:
    ths.append(threading.Thread(target=busy_loop))
if spin:
    ths.append(threading.Thread(target=spin_loop))
t0 = time.time()
for t in ths:
    t.start()
for t in ths:
    t.join()
print("%d of %d iterations differ [busy %s, spin %s, %.0f s]" % (res.get("bad", -1), iters, busy, spin, time.time() - t0))

#!/usr/bin/env python3
"""Do the library's OWN two queues disturb each other in bf16?  (A wave executing v_mfma_f32_32x32x16_bf16 disturbs waves of other kernels on its
CU: DESIGN.md section 10.)  The bf16 forward (two queues: both run kernels of this library at the same time) and a whole deterministic
training step, over and over on fixed inputs; every result compared with the first.
usage: python tools/probes/bf16_self_determinism_probe.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_threads as T  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for ngf, n in ((32, 4), (64, 8)):
    net = T.make_net("W1", 5, ngf=ngf)
    net.module.set_math("bf16")
    x = torch.from_numpy(synth.noise_window(n, 31, 256, seed=33)).cuda()
    for tq in (True, False):
        net.module.two_queues = tq
        with torch.no_grad():
            ref = net(x, False).clone()
            bad = sum(0 if torch.equal(net(x, False), ref) else 1 for _ in range(iters))
        print("bf16 forward ngf %d batch %d, two queues %s: %d of %d runs differ from the first" % (ngf, n, tq, bad, iters))
if os.environ.get("PROBE_FULL") == "1":   # BASELINE configs[2] scale: ngf 64, 32 item pairs (the two queues overlap chip-filling launches)
    _mk = T.make_net
    T.make_net = lambda kind, seed, ngf=64: _mk(kind, seed, ngf)
    work = T._train_work("W2", 12, "bf16", 32)
else:
    work = T._train_work("W2", 12, "bf16", 2)
ref = work()
bad = 0
for _ in range(max(10, iters // 10)):
    cur = work()
    same = torch.equal(cur[0], ref[0]) and all(torch.equal(a, b) for a, b in zip(cur[1], ref[1]))
    bad += 0 if same else 1
print("bf16 deterministic training step (two queues): %d of %d steps differ from the first" % (bad, max(10, iters // 10)))

#!/usr/bin/env python3
"""Round 4: some PROCESSES run the configs[2] step 3.7x slower than others on the same box (110 vs 30 ms, every step of the process).
Does a plain device copy / a bf16 GEMM-free kernel run slow in such a process too?  python tools/probes/slow_mode_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402


def copy_rate(nbytes):
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    return 2 * nbytes * 5 / (time.perf_counter() - t0) / 1e12


t_start = time.time()
r0 = copy_rate(2 << 30)
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
net = net.cuda()
dt, _ = bench.configs2_step_leg(net, torch.device("cuda"), 32, "bf16", 4)
r1 = copy_rate(2 << 30)
free, total = torch.cuda.mem_get_info()
print("copy %.2f TB/s before, step %.1f ms, copy %.2f TB/s after; reserved %.1f GB, free %.1f GB; wall %.1f s"
      % (r0, 1e3 * dt, r1, torch.cuda.memory_reserved() / 1e9, free / 1e9, time.time() - t_start))

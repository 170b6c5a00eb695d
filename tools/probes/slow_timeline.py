#!/usr/bin/env python3
"""Does the slow mode of the configs[2] step (110 ms instead of 29) switch on / off INSIDE a process?  N steps without host syncs, a
timing event every 5 steps.  python tools/probes/slow_timeline.py [steps]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.objective import StabObjective, train_step  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
net = net.cuda()
net.module.set_math("bf16")
small = synth.make_train_batch(4, seed=500)
batch = [torch.from_numpy(t).repeat((8,) + (1,) * (t.ndim - 1))[:32].cuda() for t in small]
obj = StabObjective(batchSize=32)
opt = Adam(net.parameters(), lr=1e-6, betas=(0.5, 0.999))
def ms():
    st = torch.cuda.memory_stats()
    return "mallocs %d reserved %.1f GB" % (st.get("num_device_alloc", -1), st["reserved_bytes.all.current"] / 1e9)


train_step(net, opt, batch, obj)
torch.cuda.synchronize()
print("after the warm-up step:", ms())
evs = []
t0 = time.perf_counter()
host = []
for i in range(steps):
    if i % 5 == 0:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        host.append(time.perf_counter() - t0)
    th = time.perf_counter()
    train_step(net, opt, batch, obj)
    if i < 4:
        print("after step %d: %s (host %.0f ms)" % (i + 2, ms(), 1e3 * (time.perf_counter() - th)))
e = torch.cuda.Event(enable_timing=True)
e.record()
evs.append(e)
torch.cuda.synchronize()
per = [evs[i].elapsed_time(evs[i + 1]) / 5 for i in range(len(evs) - 1)]
print("ms per step over groups of 5:", " ".join("%.0f" % p for p in per))
print("host issue time per group (ms):", " ".join("%.0f" % (1e3 * (host[i + 1] - host[i])) for i in range(len(host) - 1)))

#!/usr/bin/env python3
"""Which host call issues the at::native fill launches of a train_step?  """
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.objective import StabObjective, train_step  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402

with contextlib.redirect_stdout(sys.stderr):
    net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
net = net.cuda()
net.module.set_math("bf16")
batch = [torch.from_numpy(t).cuda() for t in synth.make_train_batch(4, seed=500)]
obj = StabObjective(batchSize=4)
opt = Adam(net.parameters(), lr=1e-6, betas=(0.5, 0.999))
train_step(net, opt, batch, obj)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(net, opt, batch, obj)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone", "aten::add_", "aten::_foreach_add_"):
        st = [s for s in ev.stack if "pwstablenet_amd" in s or "bench.py" in s][:1]
        key = (ev.name, st[0] if st else "?")
        rows[key] = rows.get(key, 0) + 1
for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:25]:
    print("%4d  %-22s %s" % (v, k[0], k[1]))

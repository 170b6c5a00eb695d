#!/usr/bin/env python3
"""Per-kernel timing of the VGG-16 perceptual term (forward of N images + data-gradient backward) through the pws_prof_* hooks."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd.perceptual import GeneratorLoss, VGG16Features  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--math", default="bf16")
a = ap.parse_args()
crit = GeneratorLoss(VGG16Features(a.math).init_random(0)).cuda()
x = (torch.rand((a.batch, 3, 256, 256), device="cuda") * 2 - 1).requires_grad_(True)
t = torch.rand((a.batch, 3, 256, 256), device="cuda") * 2 - 1
for _ in range(2):
    crit(x, t).backward()
torch.cuda.synchronize()
A.lib().pws_prof_enable(1)
crit(x, t).backward()
A.lib().pws_prof_enable(0)
recs = A.prof_collect(1 << 14)
agg = {}
for name, tag, fl, by, ms in recs:
    e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += fl; e[2] += by; e[3] += ms  # noqa: E702
tot = sum(v[3] for v in agg.values())
for k, (c, fl, by, ms) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    print("%-36s %4d calls %9.3f ms  %7.1f TFLOP/s  %7.1f GB/s" % (k, c, ms, fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9))
print("total %.2f ms for 2 x %d forwards + %d backward" % (tot, a.batch, a.batch))

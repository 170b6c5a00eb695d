"""A bare inference loop for timelines: the bench's configs[1] forward (batch 8, graph replay, two queues) N times.
usage: python3 tools/infer_loop.py [--math bf16] [--reps 20] [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pwstablenet_amd import functional as PF, hipabi as A, synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--math", default="fp32")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
net = define_G(31, 2, 64, "normal", 0.02).cuda()
if a.math == "bf16":
    net.module.set_math("bf16")
x = torch.from_numpy(synth.noise_window(a.batch, 31, 256, seed=1)).cuda()
fr = torch.from_numpy(synth.make_frames(a.batch, 3, 256, 256, seed=2)).cuda()
with torch.no_grad():
    net.module.enable_graph(True)
    for i in range(a.reps):
        f = net(x, False)
        w = PF.grid_sample(fr, f)
torch.cuda.synchronize()
print("done", float(w.abs().mean()))

#!/usr/bin/env python3
"""Does the objective's backward read memory it has not written?  Its temporaries come from torch's caching allocator (torch.empty):
the free blocks are filled with a poison value between two identical runs; any difference = an uninitialised read.
usage: python tools/probes/objective_poison_probe.py [deterministic: 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.objective import StabObjective, u8_normalize  # noqa: E402

det = not (len(sys.argv) > 1 and sys.argv[1] == "0")
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


def run():
    grids = [g.clone().requires_grad_(True) for g in grids0]
    resid = [r.clone().requires_grad_(True) for r in resid0]
    out = obj(grids, resid, rest[:, 0:3], rest[:, 3:], features, feature_adjacent, deterministic=det)
    out.loss_g.backward()
    torch.cuda.synchronize()
    return [out[k].detach().clone() for k in ("loss_g", "loss_mse", "loss_feature", "loss_delta", "loss_g2", "loss_pixel")] + [g.grad.clone() for g in grids] + [resid[2].grad.clone()]


def poison(value):
    # every size class the objective allocates: fill, free (the blocks go back to the pool with the poison in them)
    hold = [torch.full((sz,), value, device="cuda") for sz in (2 * n * 3 * 65536, n * 3 * 65536, 2 * n * 2 * 65536, 2 * n * 65536 * 4, 1 << 20, 1 << 16, 4096, 64) for _ in range(6)]
    torch.cuda.synchronize()
    del hold


ref = run()
names = ["loss_g", "loss_mse", "loss_feature", "loss_delta", "loss_g2", "loss_pixel", "ggrid0", "ggrid1", "ggrid2", "gresid"]
for value in (float("nan"), 1e30, -7.0, 0.0):
    poison(value)
    cur = run()
    d = [(names[i], float((a - b).abs().max())) for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
    print("free blocks filled with %r: %s" % (value, d if d else "identical"))

#!/usr/bin/env python3
"""A/B of grid_sample forward (north_star: "wavefront shuffles for the bilinear gather"): the product kernel (paired 8-byte gathers
per tap row) against the row-window variant (PWS_OPT_EXPERIMENT 5: one 16-byte load per source row and plane + the next lane's
first column through a wave shuffle).  N frames of 256 x 256 x 3 per launch, median of 20 launches (hipEvents, pws_prof_*)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
L = A.lib()
for GB in (64, 256):
    big = torch.rand((GB, 3, 256, 256), device=dev) * 255
    theta = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(GB, 1)
    ramp = torch.linspace(0, 6.28, 256, device=dev)
    fields = {"translation": PF.affine_grid(theta + torch.tensor([0, 0, 0.013, 0, 0, -0.021], device=dev), (GB, 3, 256, 256)),
              "5% affine + smooth residual": PF.affine_grid(theta + 0.05 * torch.randn_like(theta), (GB, 3, 256, 256)) +
              (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1))}
    for name, grid in fields.items():
        res, outs = {}, {}
        for exp, tag in ((0, "paired gathers"), (5, "row window + shuffle")):
            L.pws_set_option(A.OPT_EXPERIMENT, exp)
            with torch.no_grad():
                for _ in range(3):
                    outs[tag] = PF.grid_sample(big, grid)
                torch.cuda.synchronize()
                L.pws_prof_enable(1)
                for _ in range(20):
                    PF.grid_sample(big, grid)
                L.pws_prof_enable(0)
            r = sorted(x[4] for x in A.prof_collect() if x[0] == "grid_sample_fwd_kernel")
            res[tag] = 1e3 * r[len(r) // 2]
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
        by = 32.0 * GB * 256 * 256
        same = float((outs["paired gathers"] - outs["row window + shuffle"]).abs().max())
        print("N=%3d %-30s %s   max |diff| %.3g" % (GB, name, "   ".join("%s %.1f us (%.0f GB/s)" % (k, v, by / v / 1e3) for k, v in res.items()), same))

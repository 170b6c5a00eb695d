#!/usr/bin/env python3
"""grid_sample forward micro-benchmark (HBM roofline): GB/s of algorithmic traffic at several batch sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402

for shape in [(8, 3, 256, 256), (64, 3, 256, 256), (256, 3, 256, 256), (16, 3, 720, 1280)]:
    n, c, h, w = shape
    img = torch.rand(shape, device="cuda") * 255
    theta = torch.tensor([1, 0, 0, 0, 1, 0], device="cuda", dtype=torch.float32).repeat(n, 1)
    theta = theta + 0.05 * torch.randn_like(theta)
    grid = PF.affine_grid(theta, (n, c, h, w)) + (2.0 / w) * torch.randn((n, h, w, 2), device="cuda")
    with torch.no_grad():
        for _ in range(3):
            PF.grid_sample(img, grid)
        torch.cuda.synchronize()
        A.lib().pws_prof_enable(1)
        for _ in range(20):
            PF.grid_sample(img, grid)
        A.lib().pws_prof_enable(0)
    r = [x for x in A.prof_collect() if x[0] == "grid_sample_fwd_kernel"]
    ms = sorted(x[4] for x in r)[len(r) // 2]
    print("variant=%s shape=%s median %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (
        os.environ.get("PWS_GS_VARIANT", "default"), shape, ms * 1e3, r[0][3] / (ms * 1e-3) / 1e9,
        100 * r[0][3] / (ms * 1e-3) / 8e12))

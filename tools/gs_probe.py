#!/usr/bin/env python3
"""One warp-kernel configuration run a few times (for rocprofv3 --pmc passes): PWS_GS_VARIANT selects the kernel variant,
PWS_GS_CASE = gs256 (256 frames 256x256) | fused32 (32 frames 720p fused) | gs720 (16 frames 720p)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402

v = int(os.environ.get("PWS_GS_VARIANT", "2"))
case = os.environ.get("PWS_GS_CASE", "gs256")
A.lib().pws_set_option(100, v)
torch.manual_seed(0)
if case in ("gs256", "gs720"):
    shape = (256, 3, 256, 256) if case == "gs256" else (16, 3, 720, 1280)
    n, c, h, w = shape
    img = torch.rand(shape, device="cuda") * 255
    theta = torch.tensor([1, 0, 0, 0, 1, 0], device="cuda", dtype=torch.float32).repeat(n, 1) + 0.05 * torch.randn((n, 6), device="cuda")
    grid = PF.affine_grid(theta, (n, c, h, w)) + (2.0 / w) * torch.randn((n, h, w, 2), device="cuda")
    fn = lambda: PF.grid_sample(img, grid)  # noqa: E731
else:
    n = 32
    img = torch.rand((n, 3, 720, 1280), device="cuda") * 255
    theta = torch.tensor([1, 0, 0, 0, 1, 0], device="cuda", dtype=torch.float32).repeat(n, 1) + 0.02 * torch.randn((n, 6), device="cuda")
    field = PF.affine_grid(theta, (n, 3, 256, 256)) + (1.0 / 256) * torch.randn((n, 256, 256, 2), device="cuda")
    fn = lambda: PF.upsample_grid_sample(img, field)  # noqa: E731
with torch.no_grad():
    for _ in range(5):
        fn()
torch.cuda.synchronize()

#!/usr/bin/env python3
"""The fused 720p resize+warp on one of two fields, a few launches (for rocprofv3 --pmc passes, tools/pmc_warp720.sh):
PWS_WARP_FIELD = smooth (2 % affine + smooth +-2 px residual: what a stabiliser emits) | generator (the random-weight generator's
field, as bench.py's 720p leg produces it: its residual jumps by up to ~100 px between neighbouring 256x256 cells)."""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402

kind = os.environ.get("PWS_WARP_FIELD", "smooth")
B = 8
dev = torch.device("cuda")
torch.manual_seed(0)
rot = [torch.rand((B, 3, 720, 1280), device=dev) * 255 for _ in range(4)]
if kind == "smooth":
    th = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(B, 1)
    ramp = torch.linspace(0, 6.28, 256, device=dev)
    field = PF.affine_grid(th + 0.02 * torch.randn_like(th), (B, 3, 256, 256)) + \
        (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1))
else:
    with contextlib.redirect_stdout(sys.stderr):
        net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
    net = net.cuda()
    x = torch.from_numpy(synth.make_window(B, 31, 256, seed=11)).cuda()
    with torch.no_grad():
        field = net(x, False).clone()
    d = (field[:, :, 1:, 0] - field[:, :, :-1, 0]).abs() * 640
    print("generator field: |d source x / d cell| mean %.1f px, max %.1f px" % (float(d.mean()), float(d.max())), file=sys.stderr)
u8 = os.environ.get("PWS_WARP_U8", "0") == "1"   # the uint8 HWC variant (what VideoStabilizer runs)
if u8:
    rot = [torch.randint(0, 256, (B, 720, 1280, 3), device=dev, dtype=torch.uint8) for _ in range(4)]
with torch.no_grad():
    for i in range(8):
        if u8:
            PF.upsample_grid_sample_u8(rot[i % 4], field, swap_rb=True)
        else:
            PF.upsample_grid_sample(rot[i % 4], field)
torch.cuda.synchronize()

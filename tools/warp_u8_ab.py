#!/usr/bin/env python3
"""The fused 720p uint8 resize+warp (upsample_grid_sample_u8_kernel) on the field a stabiliser emits, on a pure translation and on the
random-weight generator's field: microseconds per launch of 8 frames (hipEvents through pws_prof_*), inputs rotated through 4 buffers
so that they come from HBM.  A/B: the product path (v_cvt_pk_u8_f32 for float -> byte) against PWS_OPT_EXPERIMENT 4 ((int) + clamp +
shift/or).  (Rounds 2-5 compared a row-window + wave-shuffle variant here -- north_star's "wavefront shuffles for the bilinear gather";
it lost 33.6 vs 27.2 us on a pure translation and was removed with the round-6 rewrite: docs/ROUNDS.md.)"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402

B = 8
dev = torch.device("cuda")
torch.manual_seed(0)
rot = [torch.randint(0, 256, (B, 720, 1280, 3), device=dev, dtype=torch.uint8) for _ in range(4)]
th = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(B, 1)
ramp = torch.linspace(0, 6.28, 256, device=dev)
fields = {"smooth (2 % affine + +-2 px residual)": PF.affine_grid(th + 0.02 * torch.randn_like(th), (B, 3, 256, 256)) +
          (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1)),
          "pure translation": PF.affine_grid(th + torch.tensor([0, 0, 0.01, 0, 0, -0.02], device=dev), (B, 3, 256, 256))}
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
net = net.cuda()
with torch.no_grad():
    fields["random-weight generator"] = net(torch.from_numpy(synth.make_window(B, 31, 256, seed=11)).cuda(), False).clone()
L = A.lib()
for name, field in fields.items():
    res = {}
    for exp, tag in ((0, "product"), (4, "clamp + shift")):
        L.pws_set_option(A.OPT_EXPERIMENT, exp)
        with torch.no_grad():
            for i in range(8):
                PF.upsample_grid_sample_u8(rot[i % 4], field, swap_rb=True)
            torch.cuda.synchronize()
            L.pws_prof_enable(1)
            for i in range(24):
                PF.upsample_grid_sample_u8(rot[i % 4], field, swap_rb=True)
            L.pws_prof_enable(0)
        r = sorted(x[4] for x in A.prof_collect() if x[0] == "upsample_grid_sample_u8_kernel")
        res[tag] = 1e3 * r[len(r) // 2]
    L.pws_set_option(A.OPT_EXPERIMENT, 0)
    by = B * 720 * 1280 * 6.0 + 8.0 * B * 256 * 256
    print("%-40s %s" % (name, "   ".join("%s %.1f us (%.0f GB/s = %.1f %% of 8 TB/s)" % (k, v, by / v / 1e3, by / v / 1e3 / 80) for k, v in res.items())))

#!/usr/bin/env python3
"""Per-launch time of every kernel of the objective (forward + backward) at m samples, torch events round 100 back-to-back launches each:
python tools/objective_kernels_time.py [m=256]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, h, w, nf = m // 2, 256, 256, 400
L = A.lib()
L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
g = torch.Generator(device="cuda").manual_seed(3)
base = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 256, device="cuda"), torch.linspace(-1, 1, 256, device="cuda"), indexing="ij")[::-1], -1)
grid = (base[None] + 0.05 * torch.randn(m, h, w, 2, device="cuda", generator=g)).contiguous()
resid = (0.02 * torch.randn(m, h, w, 2, device="cuda", generator=g)).contiguous()
rgb, stable, fake = (torch.rand(m, 3, h, w, device="cuda", generator=g) * 2 - 1 for _ in range(3))
gextra = torch.zeros(m, 3, h, w, device="cuda")
gg, gres = torch.empty_like(grid), torch.empty_like(resid)
theta = torch.tensor([[1.01, 0.02, 0.01, -0.015, 0.99, 0.02]] * n, device="cuda")
feats = torch.rand(m, nf, 6, device="cuda", generator=g) * 1.8 - 0.9
scale = torch.ones(1, device="cuda")
slots = torch.zeros(16 * A.OBJ_SLOTS, device="cuda", dtype=torch.float64)
sp = slots.data_ptr()
q = lambda k: ctypes.c_void_p(sp + k * A.OBJ_SLOTS * 8)  # noqa: E731
st = A.current_stream()
P = A.ptr
calls = [
    ("warp_norm_fwd", lambda: L.pws_warp_norm_fwd(P(rgb), 3 * h * w, P(grid), P(fake), P(stable), 3 * h * w, q(0), m, h, w, st), m * h * w * 44),
    ("temporal_l1_fwd", lambda: L.pws_temporal_l1_fwd(P(fake[:n]), P(fake[n:]), P(theta), q(1), n, h, w, st), n * h * w * 24),
    ("feature_loss_fwd", lambda: L.pws_feature_loss_fwd(P(grid), P(feats), q(2), m, nf, h, w, st), m * nf * 32),
    ("field_smoothness", lambda: L.pws_field_smoothness(P(grid), q(3), q(4), m, h, w, st), m * h * w * 8),
    ("shape_loss_fwd", lambda: L.pws_shape_loss_fwd(P(resid), q(5), m, 256, 16, st), m * h * w * 8),
    ("temporal_l1_bwd", lambda: L.pws_temporal_l1_bwd(P(fake[:n]), P(fake[n:]), P(theta), 1e-6, P(scale), P(gextra[:n]), P(gextra[n:]), n, h, w, st), n * h * w * 72),
    ("warp_norm_bwd", lambda: L.pws_warp_norm_bwd(P(rgb), 3 * h * w, P(grid), P(stable), 3 * h * w, 1e-6, P(scale), P(gextra), P(gg), 0, m, h, w, st), m * h * w * 52),
    ("feature_loss_bwd", lambda: L.pws_feature_loss_bwd(P(grid), P(feats), 1e-3, P(scale), P(gg), m, nf, h, w, st), m * nf * 40),
    ("shape_loss_bwd", lambda: L.pws_shape_loss_bwd(P(resid), 1.0, P(scale), P(gres), m, 256, 16, st), m * h * w * 16),
]
tot = 0.0
for name, fn, nbytes in calls:
    for _ in range(3):
        A.check(fn(), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        A.check(fn(), name)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 10
    tot += us
    print("%-18s %8.1f us   %7.0f GB/s (algorithmic bytes)" % (name, us, nbytes / us / 1e3))
print("one of each: %.1f us (m = %d)" % (tot, m))

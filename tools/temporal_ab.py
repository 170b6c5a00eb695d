#!/usr/bin/env python3
"""pws_temporal_l1_bwd: one lane per pixel with memory atomics (the product, PWS_OPT_EXPERIMENT 0) against tiles with the scatter in LDS (98)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402

L, st = A.lib(), A.current_stream
for n in (32, 128):
    f1 = torch.rand((n, 3, 256, 256), device="cuda") * 2 - 1
    f2 = torch.rand((n, 3, 256, 256), device="cuda") * 2 - 1
    th = torch.tensor([1, 0, 0, 0, 1, 0], device="cuda", dtype=torch.float32).repeat(n, 1) + 0.01 * torch.randn((n, 6), device="cuda")
    g1, g2 = torch.zeros_like(f1), torch.zeros_like(f2)
    for e in (93, 101, 0, 98, 99):
        L.pws_set_option(100, e)
        for _ in range(3):
            A.check(L.pws_temporal_l1_bwd(A.ptr(f1), A.ptr(f2), A.ptr(th), 0.1, None, A.ptr(g1), A.ptr(g2), n, 256, 256, st()), "t")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            A.check(L.pws_temporal_l1_bwd(A.ptr(f1), A.ptr(f2), A.ptr(th), 0.1, None, A.ptr(g1), A.ptr(g2), n, 256, 256, st()), "t")
        e1.record()
        torch.cuda.synchronize()
        print("n=%3d exp %2d: %.1f us" % (n, e, e0.elapsed_time(e1) * 1e3 / 20))
    L.pws_set_option(100, 0)

#!/usr/bin/env python3
"""Single-layer weight-gradient micro-benchmark: python tools/wgrad_bench.py KIND N H W CIN COUT [bf16]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402


def bench(kname, n, h, w, cin, cout, bf16):
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x = torch.randn((n, h, w, cin), device="cuda")
    oh, ow = (h, w) if "S1" in kname else ((h // 2, w // 2) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    dy = torch.randn((n, oh, ow, cout), device="cuda")
    dwp = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    a = A.PwsConvBwdWeightArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.math = kind, n, h, w, 1, cout, (A.MATH_BF16 if bf16 else 0)
    a.src[0].ptr, a.src[0].channels, a.src[0].ld = x.data_ptr(), cin, cin
    a.gout, a.gout_ld, a.dw_packed = dy.data_ptr(), cout, dwp.data_ptr()
    for _ in range(3):
        A.check(L.pws_conv2d_bwd_weight(ctypes.byref(a), st), "wgrad")
    torch.cuda.synchronize()
    L.pws_prof_enable(1)
    for _ in range(10):
        A.check(L.pws_conv2d_bwd_weight(ctypes.byref(a), st), "wgrad")
    L.pws_prof_enable(0)
    r = A.prof_collect()
    ms = sorted(x_[4] for x_ in r)[len(r) // 2]
    print("wgrad %-11s n=%d %dx%d %d->%d %-5s: %8.1f us  %6.1f TFLOP/s (algorithmic)" % (
        kname, n, h, w, cin, cout, "bf16" if bf16 else "fp32", ms * 1e3, r[0][2] / (ms * 1e-3) / 1e12))


if __name__ == "__main__":
    v = sys.argv[1:]
    bench(v[0], int(v[1]), int(v[2]), int(v[3]), int(v[4]), int(v[5]), len(v) > 6 and v[6] == "bf16")

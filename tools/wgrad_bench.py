#!/usr/bin/env python3
"""Single-layer weight-gradient micro-benchmark through pws_conv2d_bwd_weight (bf16 math, bf16 storage):
python tools/wgrad_bench.py KIND N H W CIN COUT      (PWS_EXPERIMENT=k selects a measured kernel variant)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402


def main():
    kname, n, h, w, cin, cout = sys.argv[1], *map(int, sys.argv[2:7])
    L, st = A.lib(), A.current_stream()
    L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
    kind = getattr(A, kname)
    k = {"CONV_K3S1": 3, "CONV_K3S2": 3, "CONVT_K3S1": 3, "CONVT_K4S2": 4, "CONV_K5S1": 5}[kname]
    oh, ow = (h, w) if "S1" in kname else ((h // 2, w // 2) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    rot = int(os.environ.get("CONV_BENCH_ROTATE", "3"))
    xs = [torch.randn((n, h, w, cin), device="cuda").bfloat16() for _ in range(rot)]
    gs = [torch.randn((n, oh, ow, cout), device="cuda").bfloat16() for _ in range(rot)]
    dw = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    a = A.PwsConvBwdWeightArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout = kind, n, h, w, 1, cout
    a.src[0].channels, a.src[0].ld = cin, cin
    a.gout_ld, a.dw_packed, a.math, a.store = cout, dw.data_ptr(), A.MATH_BF16, 1

    def launch(i):
        a.src[0].ptr, a.gout = xs[i % rot].data_ptr(), gs[i % rot].data_ptr()
        A.check(L.pws_conv2d_bwd_weight(ctypes.byref(a), st), "wgrad")
    for i in range(3):
        launch(i)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    taps = 4 if kname == "CONVT_K4S2" else k * k
    fl = 2.0 * n * oh * ow * cin * cout * taps
    print("%-10s n=%d %dx%d %d->%d wgrad bf16 exp=%s: %8.1f us  %6.1f TFLOP/s" % (kname, n, h, w, cin, cout,
          os.environ.get("PWS_EXPERIMENT", "0"), us, fl / us * 1e-6))


if __name__ == "__main__":
    main()

"""The field head's forward launch alone (csrc/head.hip): time and algorithmic GB/s per storage mode and batch.
usage: python tools/head_bench.py [n=64] [bf16|fp32]     PWS_EXPERIMENT: 90 = the vector kernel, 92 = one tile per workgroup"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pwstablenet_amd import hipabi as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
store = sys.argv[2] if len(sys.argv) > 2 else "bf16"
L, st = A.lib(), A.current_stream()
L.pws_set_option(A.OPT_EXPERIMENT, int(os.environ.get("PWS_EXPERIMENT", "0")))
h = w = 256
c = 64
xs = [(torch.randn((n, h, w, c), device="cuda") * 0.5).to(torch.bfloat16 if store == "bf16" else torch.float32) for _ in range(3)]
wout = torch.randn((9, c, 2), device="cuda") * 0.05
bout = torch.randn(2, device="cuda") * 0.1
theta = torch.randn((n, 6), device="cuda") * 0.1
grids = [torch.empty((n, h, w, 2), device="cuda") for _ in range(3)]
resid = [torch.empty((n, h, w, 2), device="cuda") for _ in range(3)]


def launch(i):
    A.check(L.pws_field_head_fwd_s(ctypes.c_void_p(xs[i % 3].data_ptr()), c, n, h, w, c, A.ptr(wout), A.ptr(bout), A.ptr(theta), 0, A.ptr(resid[i % 3]), A.ptr(grids[i % 3]),
                                   A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "field_head")


for i in range(3):
    launch(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for i in range(reps):
    launch(i)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
mb = n * h * w * (c * (2 if store == "bf16" else 4) + 16) / 1e6
print("field head n=%d %s exp=%s: %.1f us per launch, %.0f MB algorithmic -> %.2f TB/s" % (n, store, os.environ.get("PWS_EXPERIMENT", "0"), us, mb, mb / us))

"""field head: second cut (PWS_OPT_EXPERIMENT 0) against the first (35), element by element."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pwstablenet_amd import hipabi as A
L, st = A.lib(), A.current_stream()
for store in ("fp32", "bf16"):
    for (n, h, w) in ((1, 16, 32), (1, 32, 64), (2, 37, 45), (1, 256, 256)):
        c = 64
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn((n, h, w, c), device="cuda", generator=g) * 0.5).to(torch.bfloat16 if store == "bf16" else torch.float32)
        wout = torch.randn((9, c, 2), device="cuda", generator=g) * 0.05
        bout = torch.randn(2, device="cuda", generator=g) * 0.1
        out = {}
        for e in (0, 35):
            L.pws_set_option(A.OPT_EXPERIMENT, e)
            res = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(ctypes.c_void_p(x.data_ptr()), c, n, h, w, c, A.ptr(wout), A.ptr(bout), None, 0, A.ptr(res), None,
                                           A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "fh")
            out[e] = res.cpu().numpy()
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
        d = np.abs(out[0] - out[35])
        bad = np.argwhere(~(d < 1e-5))
        print(store, (n, h, w), "max diff %.3g, %d of %d differ" % (np.nanmax(d), len(bad), d.size), bad[:6].tolist())

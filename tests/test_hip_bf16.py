"""GPU parity of the bf16 matrix-core path (PWS_MATH_BF16: BASELINE configs 3/4, "bf16 with MFMA convs").

What bf16 math means here: conv operands (activations / gradients and weights) are rounded to bf16 (round to nearest
even) as they enter LDS; products are exact in fp32 and summed in fp32; everything in memory stays fp32.  So:
  * TIGHT check: on operands that are already bf16-representable the bf16 kernels must agree with an fp32 reference
    (PyTorch CPU) up to fp32 summation order: 3e-5 of the tensor's max magnitude (K up to 9216 terms).
  * STATED bf16 tolerance on arbitrary fp32 operands: each operand carries a relative rounding error <= 2^-9, a product
    <= 2^-8; for sums of K random-sign terms the error grows like sqrt(K) while the result's magnitude does too, so the
    per-layer bound is 1.5e-2 of the tensor's max magnitude (measured ~3e-3); whole generator: stage-3 field 2e-3 absolute in
    normalised coordinates for the W1 weights, 6e-2 for the saturating W2 set (measured 2.8e-4 / 1.7e-2), warped frame
    mean error < 2e-3 of the 0..255 range; training gradients: cosine > 0.995, per-tensor relative L2 error < 15 %.
"""
import ctypes
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu

KINDS = {"CONV_K3S1": ("conv", 3, 1, 1), "CONV_K3S2": ("conv", 3, 2, 1), "CONVT_K3S1": ("convT", 3, 1, 1),
         "CONVT_K4S2": ("convT", 4, 2, 1), "CONV_K5S1": ("conv", 5, 1, 2)}
PLANES = {"CONV_K3S1": 9, "CONV_K3S2": 9, "CONVT_K3S1": 9, "CONVT_K4S2": 16, "CONV_K5S1": 25}
TIGHT, STATED = 3e-5, 1.5e-2

CASES = [
    ("CONV_K3S1", (2, 20, 37), [32], 64), ("CONV_K3S1", (1, 33, 16), [64, 32], 96), ("CONV_K3S1", (3, 8, 8), [32], 16),
    ("CONV_K3S1", (5, 4, 4), [32], 32), ("CONV_K3S1", (18, 2, 2), [64], 64), ("CONV_K3S1", (2, 16, 16), [64, 64], 64),
    ("CONV_K3S1", (1, 64, 64), [64], 64),
    ("CONV_K3S2", (2, 40, 34), [32, 32], 64), ("CONV_K3S2", (2, 16, 16), [32], 32), ("CONV_K3S2", (3, 8, 8), [64], 32),
    ("CONV_K3S2", (17, 4, 4), [32, 32, 32], 16), ("CONV_K3S2", (2, 21, 9), [32], 16), ("CONV_K3S2", (1, 64, 64), [64], 128),
    ("CONV_K3S2", (2, 24, 40), [64, 32], 256), ("CONV_K3S2", (1, 16, 16), [32], 128),   # >= 128 outputs: the 64 x 128-channel (8-wave) weight gradient
    ("CONVT_K3S1", (2, 18, 21), [32, 64], 48), ("CONVT_K3S1", (2, 4, 4), [32], 16), ("CONVT_K3S1", (2, 2, 2), [256], 256),
    ("CONVT_K4S2", (2, 17, 19), [32, 32, 64], 64), ("CONVT_K4S2", (2, 8, 8), [32], 32), ("CONVT_K4S2", (3, 4, 4), [32, 32], 16),
    ("CONVT_K4S2", (20, 2, 2), [32], 64), ("CONVT_K4S2", (1, 32, 32), [128, 64], 64), ("CONVT_K4S2", (2, 16, 24), [64, 32], 128),
    ("CONV_K5S1", (1, 23, 40), [32], 64), ("CONV_K5S1", (2, 16, 16), [32], 64), ("CONV_K5S1", (1, 9, 70), [64], 32),
]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def bf16r(t):
    return t.bfloat16().float()


def relerr(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-12)


def torch_layer(kname, x, w, b, act):
    kind, k, s, p = KINDS[kname]
    y = (F.conv2d if kind == "conv" else F.conv_transpose2d)(x, w, b, stride=s, padding=p)
    return F.leaky_relu(y, 0.2) if act == 1 else F.relu(y)


def make_case(kname, shape, src_c, cout, tag):
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout, tag)).encode()))
    cin, kk, is_t = sum(src_c), KINDS[kname][1], KINDS[kname][0] == "convT"
    x = torch.from_numpy(rs.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((cin, cout, kk, kk) if is_t else (cout, cin, kk, kk)) / np.sqrt(cin * kk)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
    return x, wt, b, rs


# kernels of the bf16 path: the tiled kernel, the one-shot kernel of the deep levels (bf16 storage, <= 256 pixels per class, workspace
# given: csrc/conv_skinny16.hip), the persistent LDS-ring kernel
BF16_CONV_KERNELS = ("conv_bf16_kernel", "conv_skinny16_kernel", "conv_ring_kernel")
STORE_TOL = 5e-3   # bf16 storage of the result: relative rounding 2^-9 of each value (<= 2e-3 of max|y|) on top of TIGHT


def hip_fwd(A, kname, x, wt, b, act, src_c, cout, ws_mb, store=False, expect=None):
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    n, cin, h, w = x.shape
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    d_w = wt.cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_w), A.ptr(wp), kind, cin, cout, st), "pack")
    planes = PLANES[kname]
    cin_pad = (cin + 15) // 16 * 16
    wb = torch.empty(L.pws_packed_bf16_floats(planes, cin_pad, cout), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), planes, cin_pad, cout, st), "pack_bf16")
    xs = nhwc(x)
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, len(src_c), cout, act
    keep, c0 = [], 0
    for i, c in enumerate(src_c):
        t = xs[..., c0:c0 + c].contiguous().cuda()
        if store:
            t = t.bfloat16()
        keep.append(t)
        a.src[i].ptr, a.src[i].channels, a.src[i].ld = t.data_ptr(), c, c
        c0 += c
    oh, ow = (h, w) if "S1" in kname else (((h - 1) // 2 + 1, (w - 1) // 2 + 1) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    out = torch.full((n, oh, ow, cout), float("nan"), device="cuda", dtype=torch.bfloat16 if store else torch.float32)
    a.store = A.STORE_BF16 if store else A.STORE_FP32
    d_b = b.cuda()
    a.w_packed, a.bias, a.out, a.out_ld = wp.data_ptr(), d_b.data_ptr(), out.data_ptr(), cout
    a.math, a.w_bf16 = A.MATH_BF16, wb.data_ptr()
    if ws_mb:
        ws = torch.empty(ws_mb << 20, dtype=torch.uint8, device="cuda")
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    L.pws_prof_enable(1)
    A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv bf16")
    L.pws_prof_enable(0)
    names = [r[0] for r in A.prof_collect()]
    assert len(names) == 1 and names[0] in ((expect,) if expect else BF16_CONV_KERNELS), names  # a bf16 kernel really ran (no silent fp32 path)
    torch.cuda.synchronize()
    return out.float().cpu()


@pytest.mark.parametrize("kname,shape,src_c,cout", [c for c in CASES if c[3] % 2 == 0])
def test_bf16_storage_conv_forward(hip, kname, shape, src_c, cout):
    """PWS_STORE_BF16: sources and result are bf16 tensors (8-channel 16-byte staging loads, channel-pair dword stores)."""
    x, wt, b, _ = make_case(kname, shape, src_c, cout, "s")
    xr, wr = bf16r(x), bf16r(wt)
    want = nhwc(torch_layer(kname, xr, wr, b, 1)).numpy()
    for ws_mb in (0, 64):
        got = hip_fwd(hip, kname, xr, wt, b, 1, src_c, cout, ws_mb, store=True).numpy()
        assert not np.isnan(got).any()
        assert relerr(got, want) < STORE_TOL, relerr(got, want)


@pytest.mark.parametrize("kname,shape,src_c,cout", CASES)
@pytest.mark.parametrize("act", [1, 2])
def test_bf16_conv_forward(hip, kname, shape, src_c, cout, act):
    x, wt, b, _ = make_case(kname, shape, src_c, cout, "f")
    xr, wr = bf16r(x), bf16r(wt)
    want_tight = nhwc(torch_layer(kname, xr, wr, b, act)).numpy()
    want_fp32 = nhwc(torch_layer(kname, x, wt, b, act)).numpy()
    for ws_mb in (0, 64):
        got = hip_fwd(hip, kname, xr, wt, b, act, src_c, cout, ws_mb).numpy()  # weights are rounded by the pack kernel
        assert not np.isnan(got).any()
        assert relerr(got, want_tight) < TIGHT, relerr(got, want_tight)
    got = hip_fwd(hip, kname, x, wt, b, act, src_c, cout, 64).numpy()         # activations rounded while staging
    assert relerr(got, want_tight) < TIGHT, relerr(got, want_tight)
    assert relerr(got, want_fp32) < STATED, relerr(got, want_fp32)


@pytest.mark.parametrize("store", [False, True])
@pytest.mark.parametrize("kname,shape,src_c,cout", [c for c in CASES if c[3] % 32 == 0 and c[0] != "CONV_K5S1"])
def test_bf16_conv_data_gradient(hip, kname, shape, src_c, cout, store):
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "d")
    n, cin, h, w = x.shape
    wr = bf16r(wt)
    xg = x.clone().requires_grad_(True)
    kd, k, s_, p_ = KINDS[kname]
    y = (F.conv2d if kd == "conv" else F.conv_transpose2d)(xg, wr, None, stride=s_, padding=p_)
    dy = bf16r(torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)))
    y.backward(dy)
    ref = nhwc(xg.grad).numpy()
    d_dy = nhwc(dy).cuda()
    if store:
        d_dy = d_dy.bfloat16()
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    d_w = wt.cuda()
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(d_w), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    planes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(planes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), planes, cout, cin, st), "pack_bf16")
    for ws_mb in (0, 64):
        da = A.PwsConvBwdDataArgs()
        da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
        da.gout, da.gout_ld, da.w_dgrad, da.ndst = d_dy.data_ptr(), cout, wdg.data_ptr(), len(src_c)
        da.math, da.w_dgrad_bf16 = A.MATH_BF16, wdb.data_ptr()
        da.store = A.STORE_BF16 if store else A.STORE_FP32
        outs = []
        for i, c in enumerate(src_c):
            acc = 1 if i == 1 else 0
            o = torch.full((n, h, w, c), 0.5 if acc else float("nan"), device="cuda", dtype=torch.bfloat16 if store else torch.float32)
            outs.append(o)
            da.dst[i].ptr, da.dst[i].channels, da.dst[i].ld, da.dst[i].accumulate = o.data_ptr(), c, c, acc
        if ws_mb:
            wsb = torch.empty(ws_mb << 20, device="cuda", dtype=torch.uint8)
            da.ws, da.ws_bytes = wsb.data_ptr(), wsb.numel()
        L.pws_prof_enable(1)
        A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "bwd_data bf16")
        L.pws_prof_enable(0)
        names = [r[0] for r in A.prof_collect()]
        assert len(names) == 1 and names[0] in BF16_CONV_KERNELS, names
        torch.cuda.synchronize()
        c0 = 0
        for i, c in enumerate(src_c):
            got = outs[i].float().cpu().numpy() - (0.5 if i == 1 else 0.0)
            assert not np.isnan(got).any()
            err = np.abs(got - ref[..., c0:c0 + c]).max() / np.abs(ref).max()
            assert err < (2 * STORE_TOL if store else TIGHT), err   # accumulation into 0.5 rounds once more in bf16
            c0 += c


@pytest.mark.parametrize("act", ["ACT_LRELU", "ACT_RELU"])
@pytest.mark.parametrize("kname,shape,src_c,cout", [c for c in CASES if c[3] % 32 == 0 and c[0] != "CONV_K5S1"])
def test_bf16_data_gradient_with_fused_activation_gradient(hip, kname, shape, src_c, cout, act):
    """pws_dst.act_y: the epilogue multiplies the (accumulated) gradient of a destination by act'(y) of the forward tensor it
    belongs to -- checked against the plain call followed by the elementwise product (LeakyReLU: one more bf16 rounding in
    the two-step version, so one bf16 ulp; ReLU: bit-identical).  With and without K split, aligned (16-byte epilogue) and
    unaligned (dword epilogue: ld = c + 2) forward tensors; fp32 storage rejects the option."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "d")
    n, cin, h, w = x.shape
    kd, k, s_, p_ = KINDS[kname]
    oh, ow = (h, w) if s_ == 1 else ((h // 2, w // 2) if kd == "conv" else (2 * h, 2 * w))
    d_dy = torch.from_numpy(rs.standard_normal((n, oh, ow, cout)).astype(np.float32)).cuda().bfloat16()
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(wt.cuda()), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    planes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(planes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), planes, cout, cin, st), "pack_bf16")
    slope = 0.2 if act == "ACT_LRELU" else 0.0
    for ws_mb, pad in ((0, 0), (64, 0), (0, 2)):
        ys = [torch.from_numpy(rs.standard_normal((n, h, w, c + pad)).astype(np.float32)).cuda().bfloat16() for c in src_c]

        def run(fused):
            da = A.PwsConvBwdDataArgs()
            da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
            da.gout, da.gout_ld, da.w_dgrad, da.ndst = d_dy.data_ptr(), cout, wdg.data_ptr(), len(src_c)
            da.math, da.w_dgrad_bf16, da.store = A.MATH_BF16, wdb.data_ptr(), A.STORE_BF16
            outs = []
            for i, c in enumerate(src_c):
                acc = 1 if i == 1 else 0
                o = torch.full((n, h, w, c), 0.5 if acc else float("nan"), device="cuda", dtype=torch.bfloat16)
                outs.append(o)
                da.dst[i].ptr, da.dst[i].channels, da.dst[i].ld, da.dst[i].accumulate = o.data_ptr(), c, c, acc
                if fused and i != 2:   # a third destination, if any, stays plain
                    da.dst[i].act_y, da.dst[i].act_y_ld, da.dst[i].act = ys[i].data_ptr(), c + pad, getattr(A, act)
            if ws_mb:
                wsb = torch.empty(ws_mb << 20, device="cuda", dtype=torch.uint8)
                da.ws, da.ws_bytes = wsb.data_ptr(), wsb.numel()
            A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "bwd_data")
            torch.cuda.synchronize()
            return [o.float() for o in outs]
        plain, fused = run(False), run(True)
        for i, c in enumerate(src_c):
            want = plain[i]
            if i != 2:
                yv = ys[i][..., :c].float()
                want = want * torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, slope))
            want = want.bfloat16().float()
            got = fused[i]
            assert not torch.isnan(got).any()
            if act == "ACT_RELU":
                assert torch.equal(got, want), (i, ws_mb, pad)
            else:
                err = float((got - want).abs().max() / want.abs().max())
                assert err < 2 ** -7, (i, ws_mb, pad, err)     # the two-step product rounds twice
    da = A.PwsConvBwdDataArgs()
    da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
    d32 = d_dy.float()
    da.gout, da.gout_ld, da.w_dgrad, da.ndst = d32.data_ptr(), cout, wdg.data_ptr(), 1
    o = torch.empty((n, h, w, cin), device="cuda")
    da.dst[0].ptr, da.dst[0].channels, da.dst[0].ld = o.data_ptr(), cin, cin
    da.dst[0].act_y, da.dst[0].act_y_ld, da.dst[0].act = o.data_ptr(), cin, getattr(A, act)
    assert L.pws_conv2d_bwd_data(ctypes.byref(da), st) == -22 and b"bf16 storage" in L.pws_last_error()


@pytest.mark.parametrize("act,acc,c", [("ACT_RELU", 1, 32), ("ACT_LRELU", 1, 64), ("ACT_RELU", 0, 64), ("ACT_NONE", 1, 96)])
def test_field_head_backward_bf16_storage_and_fused_activation_gradient(hip, act, acc, c):
    """bf16 storage: the data gradient of the field head (LDS-transposed, 16-byte accesses) against the fp32-storage kernel on
    the same bf16 values, and pws_field_head_bwd_act -- dx (after the accumulation) times act'(x) -- against the plain call
    followed by the product; weight / bias / theta gradients are those of the plain call.  A pixel count that is not a
    multiple of the 256-pixel workgroup, channel counts below / at / above one 64-channel pass."""
    A = hip
    L, st = A.lib(), A.current_stream()
    rs = np.random.RandomState(7)
    n, h, w = 2, 23, 41
    x = torch.from_numpy(rs.standard_normal((n, h, w, c)).astype(np.float32)).cuda().bfloat16()
    w_out = torch.from_numpy((rs.standard_normal((9, c, 2)) * 0.05).astype(np.float32)).cuda()
    resid = torch.from_numpy(np.tanh(rs.standard_normal((n, h, w, 2))).astype(np.float32) * 0.5).cuda()
    gg = torch.from_numpy(rs.standard_normal((n, h, w, 2)).astype(np.float32)).cuda()
    old = torch.from_numpy(rs.standard_normal((n, h, w, c)).astype(np.float32)).cuda().bfloat16()

    def run(fn, store, *extra):
        xs, dx = (x, old.clone()) if store == A.STORE_BF16 else (x.float(), old.float())
        dw, db, dth = torch.zeros((9, c, 2), device="cuda"), torch.zeros(2, device="cuda"), torch.zeros((n, 6), device="cuda")
        ws = torch.empty(n * h * w * 2, device="cuda")
        A.check(fn(A.ptr(xs), c, n, h, w, c, A.ptr(w_out), A.ptr(resid), A.ptr(gg), None, 0, A.ptr(dx), c, acc, A.ptr(dw), A.ptr(db),
                   A.ptr(dth), A.ptr(ws), store, *extra, st), "field bwd")
        torch.cuda.synchronize()
        return dx.float(), dw, db, dth
    ref32 = run(L.pws_field_head_bwd_s, A.STORE_FP32)
    # c == 64 takes the fused matrix-core kernel (dx and dW from one read of x; gz and W_out enter the products as bf16);
    # PWS_OPT_EXPERIMENT 91 keeps the VALU kernels (fp32 products), which every other channel count runs anyway
    L.pws_set_option(A.OPT_EXPERIMENT, 91)
    try:
        plain = run(L.pws_field_head_bwd_s, A.STORE_BF16)
        fused = run(L.pws_field_head_bwd_act, A.STORE_BF16, getattr(A, act)) if act != "ACT_NONE" else None
    finally:
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
    assert torch.equal(plain[0], ref32[0].bfloat16().float())       # same fp32 sums, one rounding
    for a_, b_ in zip(ref32[1:], plain[1:]):
        np.testing.assert_allclose(b_.cpu().numpy(), a_.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(a_.abs().max()))
    slope = 0.2 if act == "ACT_LRELU" else 0.0
    xv = x.float()
    if act != "ACT_NONE":
        want = (plain[0] * torch.where(xv > 0, torch.ones_like(xv), torch.full_like(xv, slope))).bfloat16().float()
        if act == "ACT_RELU":
            assert torch.equal(fused[0], want)
        else:
            assert float((fused[0] - want).abs().max() / want.abs().max()) < 2 ** -7   # the two-step product rounds twice
        for a_, b_ in zip(plain[1:], fused[1:]):
            np.testing.assert_allclose(b_.cpu().numpy(), a_.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(a_.abs().max()))   # atomics order
        xf = x.float()
        assert L.pws_field_head_bwd_act(A.ptr(xf), c, n, h, w, c, A.ptr(w_out), A.ptr(resid), A.ptr(gg), None, 0, A.ptr(xf.clone()), c, 0,
                                        None, None, None, A.ptr(torch.empty(n * h * w * 2, device="cuda")), A.STORE_FP32, getattr(A, act),
                                        st) == -22
    if c == 64:
        # ---- the matrix-core kernel against a float64 model of ITS arithmetic: gz and W_out rounded to bf16, exact products
        L.pws_prof_enable(1)
        got = run(L.pws_field_head_bwd_act, A.STORE_BF16, getattr(A, act))
        L.pws_prof_enable(0)
        assert [r[0] for r in A.prof_collect()] == ["field_head_bwd_kernels"]
        ws = torch.empty(n * h * w * 2, device="cuda")     # gz of the same call (the kernel's input), recomputed by the plain call
        dxs = old.clone()
        A.check(L.pws_field_head_bwd_s(A.ptr(x), c, n, h, w, c, A.ptr(w_out), A.ptr(resid), A.ptr(gg), None, 0, A.ptr(dxs), c, acc, None, None,
                                       None, A.ptr(ws), A.STORE_BF16, st), "gz")
        gz16 = ws.view(n, h, w, 2).bfloat16().double().permute(0, 3, 1, 2)                    # n, 2, h, w
        w16 = w_out.bfloat16().double()                                                      # [tap][c][o]
        kern = w16.view(3, 3, c, 2).flip(0, 1).permute(2, 3, 0, 1).contiguous()               # [c][o][ky][kx] = W[(2-ky)*3 + 2-kx][c][o]
        dxm = torch.nn.functional.conv2d(gz16, kern, padding=1).permute(0, 2, 3, 1)          # n, h, w, c
        if acc:
            dxm = dxm + old.double()
        if act != "ACT_NONE":
            dxm = dxm * torch.where(xv > 0, torch.ones_like(xv), torch.full_like(xv, slope)).double()
        err = (got[0].double() - dxm).abs()
        assert bool((err <= 2.0 ** -7 * dxm.abs() + 1e-5).all()), float((err / (dxm.abs() + 1e-3)).max())    # one bf16 rounding (+ fp32 sum order)
        xp = torch.nn.functional.pad(xv.double().permute(0, 3, 1, 2), (1, 1, 1, 1))           # n, c, h+2, w+2
        dwm = torch.stack([torch.einsum("nchw,nohw->co", xp[:, :, ty:ty + h, tx:tx + w], gz16) for ty in range(3) for tx in range(3)])
        np.testing.assert_allclose(got[1].cpu().numpy(), dwm.cpu().numpy(), rtol=0, atol=2e-5 * float(dwm.abs().max()))
        # ... and within bf16 accuracy of the fp32-product kernels
        assert float((got[0] - (fused[0] if fused is not None else plain[0])).abs().max() / plain[0].abs().max()) < 2 ** -6
        np.testing.assert_allclose(got[1].cpu().numpy(), plain[1].cpu().numpy(), rtol=0, atol=4e-3 * float(plain[1].abs().max()))
        for a_, b_ in zip(plain[2:], got[2:]):   # bias / theta gradients: the unchanged gz kernel
            np.testing.assert_allclose(b_.cpu().numpy(), a_.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(a_.abs().max()))


DEEP_CASES = [
    ("CONV_K3S1", (8, 8, 8), [64], 64), ("CONVT_K3S1", (32, 4, 4), [64, 32], 96), ("CONV_K3S1", (64, 2, 2), [32], 64),
    ("CONV_K3S1", (7, 4, 4), [32], 32),            # fewer samples than the 4 x 4 x 16 tile holds
    ("CONVT_K4S2", (8, 8, 8), [64], 64), ("CONVT_K4S2", (20, 4, 4), [32, 32], 32), ("CONVT_K4S2", (64, 2, 2), [32], 64),
    ("CONV_K3S2", (8, 16, 16), [32], 64), ("CONV_K3S2", (18, 8, 8), [64, 32], 32),   # data gradient = sub-pixel classes over an 8 x 8 / 4 x 4 dy
]


@pytest.mark.parametrize("kname,shape,src_c,cout", DEEP_CASES)
def test_bf16_deep_map_tiles(hip, kname, shape, src_c, cout):
    """The 256-pixel tiles of the deep maps (8 x 8 x 4 samples, 4 x 4 x 16, 2 x 2 x 64; PWS_OPT_EXPERIMENT 96 takes them whenever the
    map fits, whatever the batch and K): forward (fp32 and bf16 storage, with and without K split), data gradient, fused act'."""
    L = hip.lib()
    L.pws_set_option(100, 96)
    try:
        if kname != "CONV_K3S2":
            test_bf16_storage_conv_forward(hip, kname, shape, src_c, cout)
            test_bf16_conv_forward(hip, kname, shape, src_c, cout, 1)
        if cout % 32 == 0:
            for store in (False, True):
                test_bf16_conv_data_gradient(hip, kname, shape, src_c, cout, store)
            test_bf16_data_gradient_with_fused_activation_gradient(hip, kname, shape, src_c, cout, "ACT_LRELU")
    finally:
        L.pws_set_option(100, 0)


@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 64, 64), (1, 16, 16)])
def test_field_head_forward_bf16_storage_matrix_core_kernel(hip, shape):
    """Field head on bf16-stored activations of 64 channels: the matrix-core kernel (pointwise 64 -> 18 product + 9-point stencil,
    weights as three bf16 terms; PWS_OPT_EXPERIMENT 90 switches it off) against the VALU kernel and against torch on the same
    rounded input -- both in fp32 arithmetic on exact bf16 inputs, so they agree to summation order."""
    A = hip
    L, st = A.lib(), A.current_stream()
    n, h, w = shape
    c = 64
    rs = np.random.RandomState(h * 100 + w)
    x = bf16r(torch.from_numpy(rs.standard_normal((n, c, h, w)).astype(np.float32)))
    wo = torch.from_numpy((rs.standard_normal((2, c, 3, 3)) / 12).astype(np.float32))
    bo = torch.from_numpy(rs.standard_normal(2).astype(np.float32))
    theta = torch.from_numpy((np.array([1, 0, 0, 0, 1, 0], np.float32) + 0.1 * rs.standard_normal((n, 6))).astype(np.float32))
    ref_res = torch.tanh(torch.tanh(F.conv2d(x.double(), wo.double(), bo.double(), padding=1))).permute(0, 2, 3, 1)
    ref_grid = ref_res + F.affine_grid(theta.double().view(n, 2, 3), (n, 3, h, w), align_corners=False)
    po = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1_OUT, c, 2), device="cuda")
    d_wo, d_bo, d_th = wo.cuda(), bo.cuda(), theta.cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_wo), A.ptr(po), A.CONV_K3S1_OUT, c, 2, st), "pack")
    d_x = nhwc(x).cuda().bfloat16()
    got = {}
    try:
        for exp in (0, 90):
            L.pws_set_option(100, exp)
            res = torch.full((n, h, w, 2), float("nan"), device="cuda")
            grid = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(A.ptr(d_x), c, n, h, w, c, A.ptr(po), A.ptr(d_bo), A.ptr(d_th), 0, A.ptr(res), A.ptr(grid),
                                           A.STORE_BF16, st), "field")
            got[exp] = (res.cpu().numpy(), grid.cpu().numpy())
    finally:
        L.pws_set_option(100, 0)
    for exp in (0, 90):
        np.testing.assert_allclose(got[exp][0], ref_res.numpy(), rtol=0, atol=5e-6)   # measured 3.0e-6 (fp32 sums of 576 terms, tanh twice)
        np.testing.assert_allclose(got[exp][1], ref_grid.numpy(), rtol=0, atol=6e-6)
    assert not np.array_equal(got[0][0], got[90][0]) or n * h * w < 512   # two different kernels ran


@pytest.mark.parametrize("store", ["bf16", "fp32"])
def test_field_head_persistent_walk_equals_one_tile_per_workgroup(hip, oracle, store):
    """The matrix-core field-head kernels are persistent from 3 workgroups per CU upwards (weights in registers, the next tile loaded
    under the stencil): 6 x 200 x 250 pixels = 1248 tiles make every workgroup walk two tiles, ragged at the right / bottom border.
    Bit-identical to one tile per workgroup (PWS_OPT_EXPERIMENT 92) and within the kernels' tolerance of the C oracle."""
    A = hip
    L, st = A.lib(), A.current_stream()
    n, h, w, c = 6, 200, 250, 64
    rs = np.random.RandomState(92)
    x = rs.standard_normal((n, c, h, w)).astype(np.float32)
    if store == "bf16":
        x = bf16r(torch.from_numpy(x)).numpy()
    wo = (rs.standard_normal((2, c, 3, 3)) / 12).astype(np.float32)
    bo = rs.standard_normal(2).astype(np.float32)
    theta = (np.array([1, 0, 0, 0, 1, 0], np.float32) + 0.1 * rs.standard_normal((n, 6))).astype(np.float32)
    ref_res = nhwc(torch.from_numpy(np.tanh(oracle.conv2d(x, wo, bo, 1, 1, oracle.ACT_TANH)))).numpy()
    ref_grid = ref_res + oracle.affine_grid(theta, h, w)
    po = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1_OUT, c, 2), device="cuda")
    d_wo, d_bo, d_th = torch.from_numpy(wo).cuda(), torch.from_numpy(bo).cuda(), torch.from_numpy(theta).cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_wo), A.ptr(po), A.CONV_K3S1_OUT, c, 2, st), "pack")
    d_x = nhwc(torch.from_numpy(x)).cuda().to(torch.bfloat16 if store == "bf16" else torch.float32)
    got = {}
    try:
        for exp in (0, 92):
            L.pws_set_option(A.OPT_EXPERIMENT, exp)
            res = torch.full((n, h, w, 2), float("nan"), device="cuda")
            grid = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(A.ptr(d_x), c, n, h, w, c, A.ptr(po), A.ptr(d_bo), A.ptr(d_th), 0, A.ptr(res), A.ptr(grid),
                                           A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "field")
            got[exp] = (res.cpu().numpy(), grid.cpu().numpy())
    finally:
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
    assert np.array_equal(got[0][0], got[92][0]) and np.array_equal(got[0][1], got[92][1])
    np.testing.assert_allclose(got[0][0], ref_res, rtol=0, atol=2e-5)
    np.testing.assert_allclose(got[0][1], ref_grid, rtol=0, atol=5e-5)


@pytest.mark.parametrize("store", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 37, 45), (1, 256, 256), (7, 200, 250), (24, 256, 256)])
def test_field_head_second_cut_gives_the_bits_of_the_first(hip, store, shape):
    """Round 5: field_head_v2_kernel (csrc/head.hip: 16 x 32 tiles, the matrix instruction transposed -- weights as the A operand --, three / two
    rotating register sets of pixel blocks requested ahead, tile loop unrolled so that every register index is a constant) against the first cut
    (PWS_OPT_EXPERIMENT 35): same products, same sums in the same order -- bit for bit, on one tile, ragged borders, workgroups that walk 1, 2
    and 5 tiles (the register sets rotate through every phase), with and without theta / resid."""
    A = hip
    L, st = A.lib(), A.current_stream()
    n, h, w = shape
    c = 64
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + h)
    x = (torch.randn((n, h, w, c), device="cuda", generator=g) * 0.5).to(torch.bfloat16 if store == "bf16" else torch.float32)
    wout = torch.randn((9, c, 2), device="cuda", generator=g) * 0.05
    bout = torch.randn(2, device="cuda", generator=g) * 0.1
    theta = torch.randn((n, 6), device="cuda", generator=g) * 0.1
    out = {}
    try:
        for e in (0, 35):
            L.pws_set_option(A.OPT_EXPERIMENT, e)
            res = torch.full((n, h, w, 2), float("nan"), device="cuda")
            grid = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(ctypes.c_void_p(x.data_ptr()), c, n, h, w, c, A.ptr(wout), A.ptr(bout), A.ptr(theta), 0, A.ptr(res), A.ptr(grid),
                                           A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "field head")
            only = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(ctypes.c_void_p(x.data_ptr()), c, n, h, w, c, A.ptr(wout), None, None, 1, None, A.ptr(only),
                                           A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "field head")
            out[e] = (res.clone(), grid.clone(), only.clone())
    finally:
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
    for a, b in zip(out[0], out[35]):
        assert not torch.isnan(a).any() and torch.equal(a, b)


@pytest.mark.parametrize("store,ld", [("bf16", 96), ("bf16", 72), ("fp32", 80), ("fp32", 68)])
def test_field_head_c64_with_padded_rows_through_the_c_abi(hip, oracle, store, ld):
    """pws_field_head_fwd_s called directly with c == 64, ld > c and a map that is no multiple of the 16-pixel tile,
    for both matrix-core kernels (bf16 / fp32 storage) and their PWS_OPT_EXPERIMENT 90 fallbacks, against the C oracle.  The
    columns c..ld-1 of every pixel row hold NaN: a kernel that reads past its 64 channels shows."""
    A = hip
    L, st = A.lib(), A.current_stream()
    n, h, w, c = 2, 21, 35, 64
    rs = np.random.RandomState(ld)
    x = rs.standard_normal((n, c, h, w)).astype(np.float32)
    if store == "bf16":
        x = bf16r(torch.from_numpy(x)).numpy()
    wo = (rs.standard_normal((2, c, 3, 3)) / 12).astype(np.float32)
    bo = rs.standard_normal(2).astype(np.float32)
    theta = (np.array([1, 0, 0, 0, 1, 0], np.float32) + 0.1 * rs.standard_normal((n, 6))).astype(np.float32)
    ref_res = nhwc(torch.from_numpy(np.tanh(oracle.conv2d(x, wo, bo, 1, 1, oracle.ACT_TANH)))).numpy()
    ref_grid = ref_res + oracle.affine_grid(theta, h, w)
    po = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1_OUT, c, 2), device="cuda")
    d_wo, d_bo, d_th = torch.from_numpy(wo).cuda(), torch.from_numpy(bo).cuda(), torch.from_numpy(theta).cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_wo), A.ptr(po), A.CONV_K3S1_OUT, c, 2, st), "pack")
    wide = torch.full((n, h, w, ld), float("nan"), device="cuda", dtype=torch.bfloat16 if store == "bf16" else torch.float32)
    wide[..., :c] = nhwc(torch.from_numpy(x)).cuda().to(wide.dtype)
    try:
        for exp in (0, 90):
            L.pws_set_option(A.OPT_EXPERIMENT, exp)
            res = torch.full((n, h, w, 2), float("nan"), device="cuda")
            grid = torch.full((n, h, w, 2), float("nan"), device="cuda")
            A.check(L.pws_field_head_fwd_s(A.ptr(wide), ld, n, h, w, c, A.ptr(po), A.ptr(d_bo), A.ptr(d_th), 0, A.ptr(res), A.ptr(grid),
                                           A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st), "field")
            np.testing.assert_allclose(res.cpu().numpy(), ref_res, rtol=0, atol=2e-5, err_msg="experiment %d" % exp)
            np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=0, atol=5e-5, err_msg="experiment %d" % exp)
    finally:
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
    # the ABI refuses what its 16-byte operand loads cannot take, instead of reading misaligned
    bad = wide.view(-1)[1:]   # one element (2 or 4 bytes) past a 16-byte boundary
    assert L.pws_field_head_fwd_s(A.ptr(bad), ld, n, h - 1, w, c, A.ptr(po), A.ptr(d_bo), A.ptr(d_th), 0, A.ptr(res), A.ptr(grid),
                                  A.STORE_BF16 if store == "bf16" else A.STORE_FP32, st) == -22


def test_bf16_falls_back_to_fp32_for_uncovered_shapes(hip):
    """Sources that are not multiples of 32 channels run the exact fp32 kernel even when bf16 math is requested."""
    A = hip
    x, wt, b, _ = make_case("CONV_K3S1", (1, 8, 8), [16], 16, "u")
    L, st = A.lib(), A.current_stream()
    wp = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1, 16, 16), device="cuda")
    d_w = wt.cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_w), A.ptr(wp), A.CONV_K3S1, 16, 16, st), "pack")
    wb = torch.empty(L.pws_packed_bf16_floats(9, 16, 16), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), 9, 16, 16, st), "pack_bf16")
    xs = nhwc(x).cuda()
    out = torch.empty((1, 8, 8, 16), device="cuda")
    d_b = b.cuda()
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = A.CONV_K3S1, 1, 8, 8, 1, 16, 1
    a.src[0].ptr, a.src[0].channels, a.src[0].ld = xs.data_ptr(), 16, 16
    a.w_packed, a.bias, a.out, a.out_ld, a.math, a.w_bf16 = wp.data_ptr(), d_b.data_ptr(), out.data_ptr(), 16, A.MATH_BF16, wb.data_ptr()
    A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv")
    want = nhwc(torch_layer("CONV_K3S1", x, wt, b, 1)).numpy()
    assert relerr(out.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("store", [False, True])
@pytest.mark.parametrize("kname,shape,src_c,cout", [c for c in CASES if c[3] % 8 == 0])
def test_bf16_conv_weight_gradient(hip, kname, shape, src_c, cout, store):
    """dW from bf16-rounded x and dy (fp32 accumulation, fp32 atomics) against PyTorch-CPU autograd on the same rounded operands."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "w")
    n, cin, h, w = x.shape
    xr = bf16r(x)
    wg = wt.clone().requires_grad_(True)
    kd, k, s_, p_ = KINDS[kname]
    y = (F.conv2d if kd == "conv" else F.conv_transpose2d)(xr, wg, None, stride=s_, padding=p_)
    dy = bf16r(torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)))
    y.backward(dy)
    xs = nhwc(x)  # unrounded: the kernel rounds while staging (fp32 storage) / .bfloat16() rounds the same way (bf16 storage)
    wa = A.PwsConvBwdWeightArgs()
    wa.kind, wa.n, wa.h, wa.w, wa.nsrc, wa.cout, wa.math = kind, n, h, w, len(src_c), cout, A.MATH_BF16
    wa.store = A.STORE_BF16 if store else A.STORE_FP32
    keep, c0 = [], 0
    for i, c in enumerate(src_c):
        t = xs[..., c0:c0 + c].contiguous().cuda()
        if store:
            t = t.bfloat16()
        keep.append(t)
        wa.src[i].ptr, wa.src[i].channels, wa.src[i].ld = t.data_ptr(), c, c
        c0 += c
    d_dy = nhwc(dy).cuda()
    if store:
        d_dy = d_dy.bfloat16()
    dwp = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    wa.gout, wa.gout_ld, wa.dw_packed = d_dy.data_ptr(), cout, dwp.data_ptr()
    # dbias: the column sums of dy taken along by the same kernel (accumulated: starts from 0.25 here)
    db = torch.full((cout,), 0.25, device="cuda")
    wa.dbias = db.data_ptr()
    L.pws_prof_enable(1)
    A.check(L.pws_conv2d_bwd_weight(ctypes.byref(wa), st), "bwd_weight bf16")
    L.pws_prof_enable(0)
    assert [r[0] for r in A.prof_collect()] == ["wgrad_bf16_kernel"]
    dw = torch.empty(tuple(wt.shape), device="cuda")
    A.check(L.pws_unpack_conv_weight(A.ptr(dwp), A.ptr(dw), kind, cin, cout, st), "unpack")
    err = relerr(dw.cpu().numpy(), wg.grad.numpy())
    assert err < 1e-4, err
    want_db = dy.double().sum(dim=(0, 2, 3)).numpy()
    np.testing.assert_allclose(db.cpu().numpy() - 0.25, want_db, rtol=0, atol=2e-5 * np.abs(dy.numpy()).sum(axis=(0, 2, 3)).max())


SKINNY16_CASES = [
    ("CONV_K3S1", (5, 4, 4), [32], 32), ("CONV_K3S1", (18, 2, 2), [64], 64), ("CONV_K3S1", (3, 8, 8), [32, 64], 96),
    ("CONVT_K3S1", (7, 1, 1), [64], 40), ("CONV_K3S2", (6, 8, 8), [64], 64), ("CONV_K3S2", (3, 5, 7), [32, 32], 72),
    ("CONV_K3S2", (9, 2, 2), [128], 128), ("CONVT_K4S2", (5, 4, 4), [64, 32], 64), ("CONVT_K4S2", (20, 1, 1), [128], 32),
    ("CONVT_K4S2", (2, 3, 4), [32], 96),
]


@pytest.mark.parametrize("kname,shape,src_c,cout", SKINNY16_CASES)
def test_bf16_one_shot_kernel_forward(hip, kname, shape, src_c, cout):
    """conv_skinny16_kernel (csrc/conv_skinny16.hip: one LDS-DMA burst of bf16 weights per workgroup, A operands from global memory,
    K split + the ordinary reduce epilogue) on every forward kind it covers, taken for every size up to 4096 pixels (PWS_OPT_EXPERIMENT
    72) and switched off (71 -> conv_bf16_kernel): both against PyTorch-CPU on the same rounded operands.  Its data-gradient kinds run
    in test_bf16_conv_data_gradient (the small-map cases with bf16 storage and a workspace)."""
    A = hip
    L = A.lib()
    x, wt, b, _ = make_case(kname, shape, src_c, cout, "sk16")
    xr, wr = bf16r(x), bf16r(wt)
    want = nhwc(torch_layer(kname, xr, wr, b, 1)).numpy()
    got = {}
    try:
        for e, name in ((72, "conv_skinny16_kernel"), (71, "conv_bf16_kernel")):
            L.pws_set_option(100, e)
            got[e] = hip_fwd(A, kname, xr, wt, b, 1, src_c, cout, 64, store=True, expect=name).numpy()
            assert not np.isnan(got[e]).any()
            assert relerr(got[e], want) < STORE_TOL, (name, relerr(got[e], want))
    finally:
        L.pws_set_option(100, 0)
    assert relerr(got[72], got[71]) < STORE_TOL   # same products; fp32 summation order and the bf16 rounding of the result


@pytest.mark.parametrize("kname,shape,src_c,cout", [
    ("CONV_K3S1", (2, 32, 48), [64, 64], 96),      # two input-channel blocks, cout 96: the second 64-block has one 32-channel plane
    ("CONVT_K3S1", (1, 16, 16), [64], 64),         # one tile per workgroup: the stream is prologue + tail only
    ("CONV_K3S1", (3, 16, 32), [96], 32),          # 96 input channels: block 1 has one plane; 32 output channels: one dy plane
    ("CONVT_K4S2", (2, 16, 32), [128, 64], 64),    # 4 parity classes of 2x2 taps (2 + 2 per wave half), virtual concat
    ("CONV_K3S1", (8, 64, 64), [128], 128),        # 128 tiles x 4 blocks: several tiles per workgroup, pixel split over workgroups
    ("CONVT_K4S2", (3, 24, 48), [64], 96),         # class pairs per workgroup on 8 x 16 tiles: 24 rows = 3 tiles; cout 96
    ("CONV_K5S1", (2, 32, 48), [32], 64),          # the first layer's shape: one 32-channel plane, 25 taps over 4 tap groups (7 / 7 / 7 / 4)
    ("CONV_K5S1", (1, 16, 16), [32], 96),          # one tile; cout 96: the second output block has one dy plane
    # round 4: class pairs on 128 x 64-channel workgroups (WrgCfg<..., WIDE>): matrix wave = (32-channel plane, class) x both output halves
    ("CONVT_K4S2", (2, 16, 32), [128], 64),        # one channel block, 2 class rows: few workgroups, plain 3-D grid
    ("CONVT_K4S2", (3, 24, 48), [128, 128], 96),   # two channel blocks over a virtual concat; cout 96: the second block has one dy plane
    ("CONVT_K4S2", (16, 32, 32), [256], 64),       # 256 tiles: XCD-grouped 1-D grid (64 pixel splits x 4 workgroups sharing tiles)
    ("CONVT_K4S2", (5, 40, 16), [128], 32),        # 32 output channels: the second dy half is masked; 5 tile rows per sample
    # round 4: the stride-2 kind on the ring (WrgCfg<..., S2>): 4 x 16 output tiles, halo rows stored [even | odd columns]
    ("CONV_K3S2", (2, 32, 64), [64], 64),          # 16 x 32 outputs: 4 x 2 tiles per sample; left / top padding, right / bottom edge inside
    ("CONV_K3S2", (3, 16, 32), [64, 64], 96),      # two input blocks over a virtual concat; cout 96: the second block has one dy plane
    ("CONV_K3S2", (8, 64, 64), [128], 128),        # 4 channel blocks, pixel split over workgroups
    ("CONV_K3S2", (1, 8, 32), [96], 32),           # ONE tile per sample (the whole map), 96 input channels, one dy plane
])
def test_bf16_weight_gradient_ring(hip, kname, shape, src_c, cout):
    """The persistent LDS-ring weight-gradient kernel (csrc/wgrad_ring.hip: bf16 storage, LDS-DMA tile stream; PWS_OPT_EXPERIMENT 81
    takes it whatever the number of tiles, 80 switches it off) against PyTorch-CPU autograd on the same rounded operands and
    against wgrad_bf16_kernel; dbias taken along."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "wr")
    n, cin, h, w = x.shape
    xr = bf16r(x)
    wg = wt.clone().requires_grad_(True)
    kd, k, s_, p_ = KINDS[kname]
    y = (F.conv2d if kd == "conv" else F.conv_transpose2d)(xr, wg, None, stride=s_, padding=p_)
    dy = bf16r(torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)))
    y.backward(dy)
    xs = nhwc(x)
    wa = A.PwsConvBwdWeightArgs()
    wa.kind, wa.n, wa.h, wa.w, wa.nsrc, wa.cout, wa.math = kind, n, h, w, len(src_c), cout, A.MATH_BF16
    wa.store = A.STORE_BF16
    keep, c0 = [], 0
    for i, c in enumerate(src_c):
        t = xs[..., c0:c0 + c].contiguous().cuda().bfloat16()
        keep.append(t)
        wa.src[i].ptr, wa.src[i].channels, wa.src[i].ld = t.data_ptr(), c, c
        c0 += c
    d_dy = nhwc(dy).cuda().bfloat16()
    want_db = dy.double().sum(dim=(0, 2, 3)).numpy()
    got = {}
    try:
        for force, name in ((81, "wgrad_ring_kernel"), (80, "wgrad_bf16_kernel")):
            L.pws_set_option(100, force)
            dwp = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
            wa.gout, wa.gout_ld, wa.dw_packed = d_dy.data_ptr(), cout, dwp.data_ptr()
            db = torch.full((cout,), 0.25, device="cuda")
            wa.dbias = db.data_ptr()
            L.pws_prof_enable(1)
            A.check(L.pws_conv2d_bwd_weight(ctypes.byref(wa), st), "bwd_weight bf16")
            L.pws_prof_enable(0)
            assert [r[0] for r in A.prof_collect()] == [name]
            dw = torch.empty(tuple(wt.shape), device="cuda")
            A.check(L.pws_unpack_conv_weight(A.ptr(dwp), A.ptr(dw), kind, cin, cout, st), "unpack")
            got[force] = dw.cpu().numpy()
            err = relerr(got[force], wg.grad.numpy())
            assert err < 1e-4, (name, err)
            np.testing.assert_allclose(db.cpu().numpy() - 0.25, want_db, rtol=0, atol=2e-5 * np.abs(dy.numpy()).sum(axis=(0, 2, 3)).max())
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    assert relerr(got[81], got[80]) < 2e-5   # same products, fp32 summation order only


# ---------------------------------------------------------------------------------------------- whole generator
def _make_net(kind, ngf):
    from pwstablenet_amd import synth
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, ngf, "normal", 0.02).cuda()
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(kind, ngf=ngf)})
    return net


FIELD_TOL = {"W1": 2e-3, "W2": 6e-2}   # measured 2.8e-4 / 1.7e-2 (W2 saturates the tanh heads: |field| up to 1.9)   # stated bf16 tolerance of the field, normalised coordinates (image = [-1, 1])


@pytest.mark.parametrize("store", ["bf16", "fp32"])
@pytest.mark.parametrize("kind", ["W1", "W2"])
def test_netg_bf16_inference_vs_fp32(hip, kind, store):
    """Stage-3 field with bf16 conv math against the fp32 path (itself pinned to the reference goldens): stated tolerance,
    and the warped frame (0..255 frames / 255) through the same grid_sample."""
    from pwstablenet_amd import functional as PF
    from pwstablenet_amd import synth
    net = _make_net(kind, 64)
    x = torch.from_numpy(synth.make_window(2, 31, seed=7)).cuda()
    frames = torch.from_numpy(synth.make_frames(2, 3, 256, 256, seed=7)).cuda()
    with torch.no_grad():
        f32 = net(x, False).clone()
        net.module.set_math("bf16", store=store)
        hip.lib().pws_prof_enable(1)
        f16 = net(x, False).clone()
        hip.lib().pws_prof_enable(0)
        names = [r[0] for r in hip.prof_collect()]
        net.module.set_math("fp32")
        again = net(x, False)
    assert sum(names.count(k) for k in BF16_CONV_KERNELS) >= 40, names   # every covered layer ran on the bf16 matrix cores
    assert torch.equal(again, f32)                         # switching back restores the exact fp32 path
    err = (f16 - f32).abs().max().item()
    assert err < FIELD_TOL[kind], err
    w32, w16 = PF.grid_sample(frames, f32), PF.grid_sample(frames, f16)
    werr = ((w16 - w32).abs() / 255.0)
    assert werr.mean().item() < 2e-3, werr.mean().item()
    # ---- the stated bf16 tolerance of the WARPED FRAME (north_star: "a stated bf16 tolerance"), as a bound, not a mean.
    # grid_sample evaluates the bilinear interpolant of the zero-padded frame, a continuous piecewise-bilinear function whose
    # slope along x (y) is at most the largest step between horizontal (vertical) neighbours.  Moving the sample point by
    # (dix, diy) pixels therefore moves the value by at most Gx |dix| + Gy |diy|:
    #   interior (both sample points >= 0 and <= S-1, so every tap of both is a frame pixel): Gx, Gy = largest neighbour step
    #             of the frame itself;
    #   border band (a tap of either evaluation falls in the zero padding): the step from a border pixel to 0 joins in,
    #             Gx = Gy = max(largest neighbour step, largest border value).
    # With the field tolerance above (|d field| <= FIELD_TOL, i.e. <= 128 FIELD_TOL pixels) this gives the two numbers asserted
    # at the end.  Every pixel is checked against its own bound first (which pins the kernels, not just the field error).
    S = 256
    px32, px16 = ((f32 + 1) * S - 1) / 2, ((f16 + 1) * S - 1) / 2                       # sample points in pixels (align_corners=False)
    d = (px16 - px32).abs()                                                             # (n, S, S, 2): |dix|, |diy|
    inside = ((px32 >= 0) & (px32 <= S - 1) & (px16 >= 0) & (px16 <= S - 1)).all(-1)    # (n, S, S)
    fr = frames / 255.0
    gx = (fr[..., :, 1:] - fr[..., :, :-1]).abs().amax().item()
    gy = (fr[..., 1:, :] - fr[..., :-1, :]).abs().amax().item()
    edge = max(fr[..., 0, :].amax().item(), fr[..., -1, :].amax().item(), fr[..., :, 0].amax().item(), fr[..., :, -1].amax().item())
    gb = max(gx, gy, edge)
    w_int = torch.where(inside.unsqueeze(1), werr, torch.zeros_like(werr))
    w_brd = torch.where(inside.unsqueeze(1), torch.zeros_like(werr), werr)
    slack = 2e-5   # the two fp32 evaluations themselves (~W/2 * 2^-24 px of coordinate rounding)
    bound_int = (gx * d[..., 0] + gy * d[..., 1]).unsqueeze(1) + slack
    bound_brd = (gb * (d[..., 0] + d[..., 1])).unsqueeze(1) + slack
    assert bool((w_int <= bound_int).all()), float((w_int - bound_int).max())
    assert bool((w_brd <= bound_brd).all()), float((w_brd - bound_brd).max())
    tol_int, tol_brd = FIELD_TOL[kind] * 128 * (gx + gy), FIELD_TOL[kind] * 128 * 2 * gb
    print("bf16 math / %s storage, %s: field max err %.3g; warped frame (0..1 scale): mean err %.3g, interior max %.3g (stated bound "
          "%.3g), border band max %.3g over %.2f %% of the pixels (stated bound %.3g); frame steps gx %.3g gy %.3g edge %.3g" % (
              store, kind, err, werr.mean().item(), w_int.max().item(), tol_int, w_brd.max().item(),
              100.0 * (1 - inside.float().mean().item()), tol_brd, gx, gy, edge))
    assert w_int.max().item() <= tol_int and w_brd.max().item() <= tol_brd


@pytest.mark.parametrize("ngf,n", [(16, 32), (48, 16)])
def test_netg_bf16_math_with_sources_off_the_32_channel_grid_never_reads_the_winograd_copies(hip, ngf, n):
    """ngf 16 / 48: up_bottom1.conv_same reads cat(ngf, ngf) -- cin % 32 == 0, so the layer has bf16 weights and the bf16-mode
    pack leaves its Winograd copies out, but the bf16 kernels decline sources of channels % 32 != 0 and pws_conv2d_fwd falls through
    to the fp32 kernels: from wblocks >= 2048 that used to be the Winograd kernel on weights nobody packed.  The packed buffer is
    poisoned with NaN before the bf16-mode pack, and the weights are changed once (a stale copy of the fp32-mode pack would pass
    otherwise)."""
    from pwstablenet_amd import synth
    net = _make_net("W1", ngf)
    x = torch.from_numpy(synth.make_window(n, 31, seed=11)).cuda()
    with torch.no_grad():
        net(x[:2], False)                                   # fp32-mode pack (Winograd copies of the OLD weights)
        for p in net.parameters():
            if p.dim() == 4:
                p.mul_(1.25)
        f32 = net(x, False).clone()
        net.module.set_math("bf16", store="fp32")
        nfl = hip.lib().pws_netg_packed_floats(31, ngf)
        net.module._packed = torch.full((nfl,), float("nan"), device="cuda")
        net.module._packed_key = None
        f16 = net(x, False).clone()
    assert bool(torch.isfinite(f16).all())
    err = (f16 - f32).abs().max().item()
    assert err < FIELD_TOL["W1"], err


@pytest.mark.parametrize("store", ["bf16", "fp32"])
def test_netg_bf16_training_step_gradients_vs_fp32(hip, store):
    """One training forward + backward (smooth field loss) in bf16 math: loss within 1 % of fp32, cosine similarity > 0.995 over
    all 48.5 M gradient entries, and every parameter tensor's gradient within 15 % relative L2 error of its fp32 gradient (bf16
    rounding of activations and gradients accumulates through the 30+ layers between a deep weight and the loss)."""
    from pwstablenet_amd import synth
    net = _make_net("W1", 64)
    x = torch.from_numpy(synth.make_window(2, 31, seed=11)).cuda()
    tgt = torch.from_numpy(np.random.RandomState(5).standard_normal((2, 256, 256, 2)).astype(np.float32) * 0.1).cuda()

    def step(math):
        net.module.set_math(math, store=store if math == "bf16" else None)
        net.zero_grad(set_to_none=True)
        grids, resid = net(x)
        loss = sum(((g - tgt) ** 2).mean() for g in grids) + sum((r ** 2).mean() for r in resid)
        loss.backward()
        return loss.item(), [p.grad.clone() for p in net.parameters()]

    l32, g32 = step("fp32")
    hip.lib().pws_prof_enable(1)
    l16, g16 = step("bf16")
    hip.lib().pws_prof_enable(0)
    if store == "bf16":   # the sign-bit epilogues (PWS_OPT_EXPERIMENT 12: off) change which bytes are read, not one result
        hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 12)
        try:
            l12, g12 = step("bf16")
        finally:
            hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 0)
        if not getattr(net.module, "deterministic", False):
            net.module.deterministic = True   # (atomics' arrival order moves the weight gradients' last bits otherwise)
            try:
                ld0, gd0 = step("bf16")
                hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 12)
                ld12, gd12 = step("bf16")
            finally:
                hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 0)
                net.module.deterministic = False
            assert ld0 == ld12 and all(torch.equal(a, b) for a, b in zip(gd0, gd12))
    names = [r[0] for r in hip.prof_collect()]
    net.module.set_math("fp32")
    nconv = sum(names.count(k) for k in BF16_CONV_KERNELS)
    assert names.count("wgrad_bf16_kernel") >= 40 and nconv >= 80, (names.count("wgrad_bf16_kernel"), nconv)
    assert abs(l16 - l32) < 1e-2 * abs(l32), (l16, l32)
    dot = sum((a * b).sum().item() for a, b in zip(g16, g32))
    n16 = sum((a * a).sum().item() for a in g16) ** 0.5
    n32 = sum((b * b).sum().item() for b in g32) ** 0.5
    assert dot / (n16 * n32) > 0.995, dot / (n16 * n32)
    names_p = [k for k, _ in net.named_parameters()]
    rel = [(((a - b).norm() / b.norm().clamp_min(1e-20)).item(), k) for a, b, k in zip(g16, g32, names_p)]
    worst, worst_name = max(rel)
    print("bf16 training (%s storage): loss %.6g vs %.6g, grad cosine %.5f, worst per-tensor relative L2 error %.3g (%s)" % (
        store, l16, l32, dot / (n16 * n32), worst, worst_name))
    for r_, k in sorted(rel)[-6:]:
        print("   %-44s %.4f" % (k, r_))
    assert worst < 0.15, (worst, worst_name)


@pytest.mark.parametrize("ngf", [64, 32])
def test_one_pass_training_pack_gives_the_bits_of_the_four_pass_pack(hip, ngf):
    """pws_netg_pack_weights_train (bf16 math): torch layouts -> bf16 forward + bf16 data-gradient copies in one pass, no fp32 packed
    copies of the conv layers.  A deterministic training step on it must give the fields, the loss and all 92 gradients BIT FOR BIT as on
    the buffers the separate calls pack (PWS_OPT_EXPERIMENT 120 switches the one-pass kernel off); both packed buffers are poisoned with
    NaN first, so a read of a copy the one-pass kernel does not make would show.  ngf 32: layers of 32 output channels do not qualify
    (padded to 64 in the bf16 layout) and take the ordinary path next to the ones that do."""
    from pwstablenet_amd import synth
    net = _make_net("W1", ngf)
    net.module.set_math("bf16")
    net.module.deterministic = True
    x = torch.from_numpy(synth.make_window(2, 31, seed=21)).cuda()
    tgt = torch.from_numpy(np.random.RandomState(6).standard_normal((2, 256, 256, 2)).astype(np.float32) * 0.1).cuda()
    L = hip.lib()

    def step(exp):
        L.pws_set_option(hip.OPT_EXPERIMENT, exp)
        try:
            m = net.module
            m._packed = torch.full((L.pws_netg_packed_floats(31, ngf),), float("nan"), device="cuda")
            m._packed_dgrad = torch.full((L.pws_netg_packed_dgrad_floats(31, ngf),), float("nan"), device="cuda")
            m._packed_key = m._packed_dgrad_key = None
            net.zero_grad(set_to_none=True)
            L.pws_prof_enable(1)
            grids, resid = net(x)
            loss = sum(((g - tgt) ** 2).mean() for g in grids) + sum((r ** 2).mean() for r in resid)
            loss.backward()
            L.pws_prof_enable(0)
            names = [r[0] for r in hip.prof_collect()]
            return [g.detach().clone() for g in grids], loss.item(), [p.grad.clone() for p in net.parameters()], names
        finally:
            L.pws_set_option(hip.OPT_EXPERIMENT, 0)

    g1, l1, d1, _ = step(0)
    g0, l0, d0, _ = step(120)
    assert np.isfinite(l1) and l1 == l0
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))
    assert all(torch.equal(a, b) for a, b in zip(d1, d0))
    # and the inference path on the one-pass buffers (same weights version: no re-pack)
    with torch.no_grad():
        f_train_pack = net(x, False).clone()
        net.module._packed_key = None
        f_own_pack = net(x, False)
    assert torch.equal(f_train_pack, f_own_pack)


def test_training_pack_keeps_the_fp32_first_layer_for_wide_windows(hip):
    """input_nc = 64 (cin_pad % 32 == 0, so the first layer has bf16 weights), bf16 math, fp32 storage, training: run_conv takes the bf16
    first-layer kernel only for windows of <= 32 channels (the NHWC-32 copy), so this window goes to the fp32 first-layer kernels -- which
    read the fp32 packed copy.  pws_netg_pack_weights_train must therefore not treat layer 0 as bf16-only (round-5 review: it did, and the
    forward read whatever torch.empty had left there).  Both packed buffers are poisoned with NaN first."""
    from pwstablenet_amd import synth
    from pwstablenet_amd.lib.networks_cascading import define_G
    ngf, nc = 32, 64
    net = define_G(nc, 2, ngf, "normal", 0.02).cuda()
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", ngf=ngf, input_nc=nc)})
    x = torch.from_numpy(synth.make_window(2, nc, seed=31)).cuda()
    tgt = torch.from_numpy(np.random.RandomState(7).standard_normal((2, 256, 256, 2)).astype(np.float32) * 0.1).cuda()
    L = hip.lib()

    def step(math):
        m = net.module
        m.set_math(math, store="fp32" if math == "bf16" else None)
        m._packed = torch.full((L.pws_netg_packed_floats(nc, ngf),), float("nan"), device="cuda")
        m._packed_dgrad = torch.full((L.pws_netg_packed_dgrad_floats(nc, ngf),), float("nan"), device="cuda")
        m._packed_key = m._packed_dgrad_key = None
        net.zero_grad(set_to_none=True)
        grids, resid = net(x)
        loss = sum(((g - tgt) ** 2).mean() for g in grids) + sum((r ** 2).mean() for r in resid)
        loss.backward()
        return [g.detach().clone() for g in grids], loss.item(), [p.grad.clone() for p in net.parameters()]

    g32, l32, d32 = step("fp32")
    g16, l16, d16 = step("bf16")
    net.module.set_math("fp32")
    assert np.isfinite(l16) and all(bool(torch.isfinite(g).all()) for g in g16) and all(bool(torch.isfinite(d).all()) for d in d16)
    assert abs(l16 - l32) < 1e-2 * abs(l32), (l16, l32)
    assert max((a - b).abs().max().item() for a, b in zip(g16, g32)) < FIELD_TOL["W1"]


def _sign_bytes(y_nhwc_bf16):
    """Bit (c & 7) of byte [pixel][c / 8] = (y[pixel][c] > 0): pws_conv_args.out_sign's layout, from the bf16 tensor itself."""
    pos = (y_nhwc_bf16.float() > 0).to(torch.uint8).cpu().numpy()
    n, h, w, c = pos.shape
    return np.packbits(pos.reshape(n, h, w, c // 8, 8), axis=-1, bitorder="little")[..., 0]


@pytest.mark.parametrize("kname,shape,src_c,cout,expect", [
    ("CONV_K3S1", (2, 128, 128), [32, 32], 64, "conv_ring_kernel"),      # the ring kernel writes the bytes in its epilogue
    ("CONVT_K4S2", (2, 64, 64), [64], 72, "conv_ring_kernel"),           # parity classes, cout ending inside a 64-channel block
    ("CONV_K3S2", (2, 128, 128), [64], 64, "conv_ring_kernel"),
    ("CONV_K3S1", (3, 12, 20), [32], 40, "conv_bf16_kernel"),            # another kernel + the pass over its output
    ("CONVT_K4S2", (2, 4, 4), [64], 64, "conv_skinny16_kernel"),
])
def test_forward_sign_bits_equal_the_sign_of_the_stored_output(hip, kname, shape, src_c, cout, expect):
    """pws_conv_args.out_sign (ABI version 3): the sign bits of the ROUNDED bf16 output, one byte per 8 channels, whichever kernel
    produced the output; rows of out_sign_ld > cout / 8 bytes keep their padding."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "sg")
    n, cin, h, w = x.shape
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight(A.ptr(wt.cuda()), A.ptr(wp), kind, cin, cout, st), "pack")
    wb = torch.empty(L.pws_packed_bf16_floats(PLANES[kname], (cin + 15) // 16 * 16, cout), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), PLANES[kname], (cin + 15) // 16 * 16, cout, st), "pack_bf16")
    xs = nhwc(x)
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, len(src_c), cout, A.ACT_LRELU
    keep, c0 = [], 0
    for i, c in enumerate(src_c):
        t = xs[..., c0:c0 + c].contiguous().cuda().bfloat16()
        keep.append(t)
        a.src[i].ptr, a.src[i].channels, a.src[i].ld = t.data_ptr(), c, c
        c0 += c
    oh, ow = (h, w) if "S1" in kname else (((h - 1) // 2 + 1, (w - 1) // 2 + 1) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    out = torch.full((n, oh, ow, cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    sld = cout // 8 + 3
    sign = torch.full((n, oh, ow, sld), 0xA5, device="cuda", dtype=torch.uint8)
    d_b = b.cuda()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    a.store, a.math, a.w_bf16 = A.STORE_BF16, A.MATH_BF16, wb.data_ptr()
    a.w_packed, a.bias, a.out, a.out_ld, a.ws, a.ws_bytes = wp.data_ptr(), d_b.data_ptr(), out.data_ptr(), cout, ws.data_ptr(), ws.numel()
    a.out_sign, a.out_sign_ld = sign.data_ptr(), sld
    L.pws_set_option(A.OPT_EXPERIMENT, 21 if expect == "conv_ring_kernel" else 0)   # (21: the ring kernel also for launches of few units)
    try:
        L.pws_prof_enable(1)
        A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv with sign bits")
        L.pws_prof_enable(0)
    finally:
        L.pws_set_option(A.OPT_EXPERIMENT, 0)
    assert [r[0] for r in A.prof_collect()] == [expect]
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    got = sign.cpu().numpy()
    assert np.array_equal(got[..., :cout // 8], _sign_bytes(out)) and (got[..., cout // 8:] == 0xA5).all()
    # fp32 storage has no use for them and says so
    a.store = A.STORE_FP32
    assert L.pws_conv2d_fwd(ctypes.byref(a), st) == -22 and b"out_sign" in L.pws_last_error()


@pytest.mark.parametrize("kname,shape,src_c,cout", [
    ("CONVT_K4S2", (4, 128, 128), [64, 64], 64),     # k4 s2 data gradient on the ring (256 units): planes, two destinations
    ("CONV_K3S2", (4, 128, 128), [64, 32], 64),      # sub-pixel classes
    ("CONV_K3S1", (6, 128, 128), [32, 64], 64),
    ("CONV_K3S1", (3, 12, 20), [32], 64),            # not on the ring: act_sign is ignored, act_y read
])
@pytest.mark.parametrize("act", ["ACT_LRELU", "ACT_RELU"])
def test_data_gradient_with_sign_bits_is_bit_identical_to_reading_the_forward_tensor(hip, kname, shape, src_c, cout, act):
    """pws_dst.act_sign: the ring kernel's sign-bit epilogue (one byte per 8 channels, more requests in flight) gives the same bytes as
    the epilogue that reads act_y; a destination without act' beside one with it; PWS_OPT_EXPERIMENT 12 = sign bits ignored."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "sgd")
    n, cin, h, w = x.shape
    kd, k, s_, p_ = KINDS[kname]
    oh, ow = (h, w) if s_ == 1 else ((h // 2, w // 2) if kd == "conv" else (2 * h, 2 * w))
    d_dy = torch.from_numpy(rs.standard_normal((n, oh, ow, cout)).astype(np.float32)).cuda().bfloat16()
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(wt.cuda()), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    planes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(planes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), planes, cout, cin, st), "pack_bf16")
    ys = [torch.from_numpy(rs.standard_normal((n, h, w, c)).astype(np.float32)).cuda().bfloat16() for c in src_c]
    signs = [torch.from_numpy(_sign_bytes(y)).cuda() for y in ys]
    wsb = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)

    def run(with_sign, exp):
        L.pws_set_option(A.OPT_EXPERIMENT, exp)
        da = A.PwsConvBwdDataArgs()
        da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
        da.gout, da.gout_ld, da.w_dgrad, da.ndst = d_dy.data_ptr(), cout, wdg.data_ptr(), len(src_c)
        da.math, da.w_dgrad_bf16, da.store = A.MATH_BF16, wdb.data_ptr(), A.STORE_BF16
        da.ws, da.ws_bytes = wsb.data_ptr(), wsb.numel()
        outs = []
        for i, c in enumerate(src_c):
            acc = 1 if i == 0 else 0
            o = torch.full((n, h, w, c), 0.25 if acc else float("nan"), device="cuda", dtype=torch.bfloat16)
            outs.append(o)
            da.dst[i].ptr, da.dst[i].channels, da.dst[i].ld, da.dst[i].accumulate = o.data_ptr(), c, c, acc
            if i == 0:   # the second destination, if any, stays plain
                da.dst[i].act_y, da.dst[i].act_y_ld, da.dst[i].act = ys[i].data_ptr(), c, getattr(A, act)
                if with_sign:
                    da.dst[i].act_sign, da.dst[i].act_sign_ld = signs[i].data_ptr(), c // 8
        try:
            L.pws_prof_enable(1)
            A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "bwd_data")
            L.pws_prof_enable(0)
        finally:
            L.pws_set_option(A.OPT_EXPERIMENT, 0)
        ran.append([r[0] for r in A.prof_collect()])
        torch.cuda.synchronize()
        return [o.float().cpu() for o in outs]
    ran = []
    ref, got, off = run(False, 0), run(True, 0), run(True, 12)
    assert ran[0] == ran[1] == ran[2] and (ran[0] == ["conv_ring_kernel"]) == (shape[1] >= 128), ran
    for r_, g_, o_ in zip(ref, got, off):
        assert not torch.isnan(r_).any()
        assert torch.equal(r_, g_) and torch.equal(r_, o_)
    if ran[0] == ["conv_ring_kernel"]:
        # round 4: on 32-wide tiles the kinds without a tap mask run FOUR matrix waves of 4 tile rows (RgCfg<..., MT = 4>); PWS_OPT_EXPERIMENT 105
        # = the eight-wave kernel.  Same products, another fp32 summation order, one rounding to bf16
        for exp, with_sign in ((105, True), (105, False)):
            eight = run(with_sign, exp)
            assert ran[-1] == ["conv_ring_kernel"]
            for r_, e_ in zip(ref, eight):
                assert relerr(e_.numpy(), r_.numpy()) < 2.0 ** -7   # of the largest value: at most one bf16 ulp apart


def test_nchw_to_nhwc_pad_and_first_layer_bf16(hip):
    """The 31-channel NCHW window becomes a 32-channel NHWC source (zero pad channel); the bf16 k5 kernel on it matches
    the fp32 reference on bf16-rounded operands."""
    A = hip
    L, st = A.lib(), A.current_stream()
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.standard_normal((2, 31, 19, 37)).astype(np.float32))
    d_x = x.cuda()
    out = torch.full((2, 19, 37, 32), float("nan"), device="cuda")
    A.check(L.pws_nchw_to_nhwc_pad(A.ptr(d_x), A.ptr(out), 2, 31, 19, 37, 32, st), "nchw_to_nhwc_pad")
    got = out.cpu()
    assert torch.equal(got[..., :31], x.permute(0, 2, 3, 1)) and (got[..., 31] == 0).all()
    wt = torch.from_numpy((rs.standard_normal((64, 31, 5, 5)) / 28.0).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(64).astype(np.float32))
    x32 = torch.cat([x, torch.zeros(2, 1, 19, 37)], 1)
    w32 = torch.cat([wt, torch.zeros(64, 1, 5, 5)], 1)
    y = hip_fwd(A, "CONV_K5S1", bf16r(x32), w32, b, 1, [32], 64, 0).numpy()
    want = nhwc(torch_layer("CONV_K5S1", bf16r(x), bf16r(wt), b, 1)).numpy()
    assert relerr(y, want) < TIGHT


@pytest.mark.parametrize("kname,shape,src_c,cout,force", [
    ("CONVT_K4S2", (2, 16, 32), [128], 64, 81),        # class pairs on 128 x 64-channel workgroups: both operand pairs in one ring launch
    ("CONV_K3S1", (3, 16, 32), [64, 64], 96, 81),      # 3x3 on the ring, virtual concat, cout 96
    ("CONV_K3S2", (2, 32, 64), [64], 64, 81),          # stride-2 kind on the ring (ring of 3: the tile walk crosses from pair 1 to pair 2)
    ("CONV_K5S1", (2, 32, 48), [32], 64, 81),          # the first layer's shape on the ring
    ("CONV_K3S1", (2, 16, 16), [32], 32, 0),           # not covered by the ring (32 input channels), 16 x 16 tiles: two launches inside the call
    ("CONVT_K4S2", (2, 8, 8), [64], 32, 80),           # ring switched off, 8 x 8 tiles: ONE launch of wgrad_bf16_kernel over both pairs (round 6)
    ("CONV_K3S1", (3, 8, 8), [64, 32], 72, 0),         # the same on the 3x3 kind, virtual concat, cout ending inside a block
    ("CONVT_K4S2", (4, 4, 4), [64], 64, 0),            # 4 x 4 maps: tiles of 4 samples, a pair of 4 + 4 samples = two whole tiles
    ("CONVT_K3S1", (16, 2, 2), [64, 32], 64, 0),       # 2 x 2 maps: tiles of 16 samples
    ("CONVT_K4S2", (6, 4, 4), [64], 64, 0),            # 6 samples in tiles of 4: a tile would straddle the pairs -> two launches
    ("CONV_K3S1", (2, 8, 8), [64], 64, 188),           # merged launch switched off: two launches inside the call
])
def test_bf16_weight_gradient_of_a_layer_used_twice(hip, kname, shape, src_c, cout, force):
    """pws_conv_bwd_weight_args.gout2 / src2_ptr (round 4): stages 2 and 3 of the generator run the same modules, so a shared layer's weight
    gradient is the sum over TWO operand pairs of the same geometry; given both, the ring kernels walk both tensors' tiles in one
    launch.  Against the sum of two single-pair calls (same products, fp32 summation order) and PyTorch-CPU autograd; dbias over both."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    kd, k, s_, p_ = KINDS[kname]
    pairs = []
    for tag in ("p1", "p2"):
        x, wt, b, rs = make_case(kname, shape, src_c, cout, tag)
        xr = bf16r(x)
        wg = wt.clone().requires_grad_(True)
        y = (F.conv2d if kd == "conv" else F.conv_transpose2d)(xr, wg, None, stride=s_, padding=p_)
        dy = bf16r(torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)))
        y.backward(dy)
        pairs.append((x, dy, wg.grad.numpy(), wt))
    n, cin, h, w = pairs[0][0].shape
    want_dw = pairs[0][2] + pairs[1][2]
    want_db = (pairs[0][1].double().sum(dim=(0, 2, 3)) + pairs[1][1].double().sum(dim=(0, 2, 3))).numpy()

    def args(pair, second=None):
        wa = A.PwsConvBwdWeightArgs()
        wa.kind, wa.n, wa.h, wa.w, wa.nsrc, wa.cout, wa.math, wa.store = kind, n, h, w, len(src_c), cout, A.MATH_BF16, A.STORE_BF16
        keep, c0 = [], 0
        xs = nhwc(pair[0])
        xs2 = nhwc(second[0]) if second is not None else None
        for i, c in enumerate(src_c):
            t = xs[..., c0:c0 + c].contiguous().cuda().bfloat16()
            keep.append(t)
            wa.src[i].ptr, wa.src[i].channels, wa.src[i].ld = t.data_ptr(), c, c
            if xs2 is not None:
                t2 = xs2[..., c0:c0 + c].contiguous().cuda().bfloat16()
                keep.append(t2)
                wa.src2_ptr[i] = t2.data_ptr()
            c0 += c
        g = nhwc(pair[1]).cuda().bfloat16()
        keep.append(g)
        wa.gout, wa.gout_ld = g.data_ptr(), cout
        if second is not None:
            g2 = nhwc(second[1]).cuda().bfloat16()
            keep.append(g2)
            wa.gout2 = g2.data_ptr()
        return wa, keep

    def run(calls):
        dwp = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
        db = torch.zeros(cout, device="cuda")
        names = []
        for wa, _keep in calls:
            wa.dw_packed, wa.dbias = dwp.data_ptr(), db.data_ptr()
            L.pws_prof_enable(1)
            A.check(L.pws_conv2d_bwd_weight(ctypes.byref(wa), st), "bwd_weight pair")
            L.pws_prof_enable(0)
            names += [r[0] for r in A.prof_collect()]
        dw = torch.empty(tuple(pairs[0][3].shape), device="cuda")
        A.check(L.pws_unpack_conv_weight(A.ptr(dwp), A.ptr(dw), kind, cin, cout, st), "unpack")
        return dw.cpu().numpy(), db.cpu().numpy(), names
    try:
        L.pws_set_option(100, force)
        both, db_both, names = run([args(pairs[0], pairs[1])])
        two, db_two, _ = run([args(pairs[0]), args(pairs[1])])
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    two_launches = force == 188 or (kname, shape) in (("CONVT_K4S2", (6, 4, 4)), ("CONV_K3S1", (2, 16, 16)))
    assert names == (["wgrad_ring_kernel"] if force == 81 else ["wgrad_bf16_kernel"] * (2 if two_launches else 1)), names
    assert relerr(both, want_dw) < 1e-4 and relerr(both, two) < 2e-5
    tol = 2e-5 * (np.abs(pairs[0][1].numpy()).sum(axis=(0, 2, 3)).max() + np.abs(pairs[1][1].numpy()).sum(axis=(0, 2, 3)).max())
    np.testing.assert_allclose(db_both, want_db, rtol=0, atol=tol)
    np.testing.assert_allclose(db_two, want_db, rtol=0, atol=tol)

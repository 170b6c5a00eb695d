"""Seeded random sweep over the convolution dispatch with bf16 math and bf16 activation storage (the configs[2] / [3] path): shapes
drawn around the selectors' thresholds (ring kernel / one-shot kernel / tiled kernel; whole and ragged tiles; 1..4 sources of the
virtual concat; cout ending inside a 64-channel block), forward, data gradient (with accumulation into a destination; act'
of a forward tensor is covered by tests/test_hip_bf16.py) and weight gradient.  Every result is checked against PyTorch on the same rounded operands, and the
second-generation kernels against the first-generation ones (PWS_OPT_EXPERIMENT 20 / 71 / 80 switch them off, 21 / 81 take the ring kernels wherever they are covered, 186 / 187 the ring kernel's small units)."""
import ctypes
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KINDS = {"CONV_K3S1": ("conv", 3, 1, 1), "CONV_K3S2": ("conv", 3, 2, 1), "CONVT_K3S1": ("convT", 3, 1, 1), "CONVT_K4S2": ("convT", 4, 2, 1)}
PLANES = {"CONV_K3S1": 9, "CONV_K3S2": 9, "CONVT_K3S1": 9, "CONVT_K4S2": 16}
TOL = 6e-3   # bf16 storage of the result (2^-9 relative per value) + fp32 summation order, relative to max |reference|


def _cases():
    rs = np.random.RandomState(20260)
    out = []
    names = list(KINDS)
    for i in range(56):
        kname = names[i % 4]
        # map sizes around the tile shapes: 1..8 (one-shot / small tiles), 16 / 32 multiples (ring), ragged ones (tiled kernel)
        h = int(rs.choice([1, 2, 3, 4, 5, 8, 8, 12, 16, 16, 20, 32, 32, 48]))
        w = int(rs.choice([1, 2, 4, 4, 7, 8, 8, 16, 16, 24, 32, 32, 64]))
        n = int(rs.choice([1, 2, 3, 4, 8, 8, 16]))
        if n * h * w > 16384:
            n = max(1, 16384 // (h * w))
        nsrc = int(rs.choice([1, 1, 2, 3]))
        src_c = [int(rs.choice([32, 32, 64, 96])) for _ in range(nsrc)]
        cout = int(rs.choice([32, 64, 64, 96, 128, 160]))
        out.append((kname, (n, h, w), src_c, cout))
    return out


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def bf16r(t):
    return t.bfloat16().float()


def relerr(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("kname,shape,src_c,cout", _cases())
def test_bf16_storage_dispatch_sweep(hip, kname, shape, src_c, cout):
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    kd, k, s_, p_ = KINDS[kname]
    n, h, w = shape
    cin = sum(src_c)
    g = torch.Generator().manual_seed(zlib.crc32(repr((kname, shape, src_c, cout)).encode()))
    x = bf16r(torch.randn((n, cin, h, w), generator=g))
    wt = torch.randn((cin, cout, k, k) if kd == "convT" else (cout, cin, k, k), generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    wr = bf16r(wt)
    conv = F.conv2d if kd == "conv" else F.conv_transpose2d
    # ---- reference: PyTorch-CPU fp32 on the same rounded operands: forward, gradients wrt x and wrt w
    xg = x.clone().requires_grad_(True)
    wg = wr.clone().requires_grad_(True)
    y = conv(xg, wg, b, stride=s_, padding=p_)
    want_y = nhwc(F.leaky_relu(y, 0.2)).detach()
    dy = bf16r(torch.randn(tuple(y.shape), generator=g))
    (gx, gw) = torch.autograd.grad(y, (xg, wg), dy)
    want_dx, want_dw = nhwc(gx), gw
    oh, ow = y.shape[2], y.shape[3]

    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    d_w = wt.cuda()
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight(A.ptr(d_w), A.ptr(wp), kind, cin, cout, st), "pack")
    wb = torch.empty(L.pws_packed_bf16_floats(PLANES[kname], cin, cout), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), PLANES[kname], cin, cout, st), "pack_bf16")
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(d_w), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    dplanes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(dplanes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), dplanes, cout, cin, st), "pack_bf16 dgrad")
    xs = nhwc(x).cuda()
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(xs[..., c0:c0 + c].contiguous().bfloat16())
        c0 += c
    d_dy = nhwc(dy).cuda().bfloat16()
    d_b = b.cuda()

    def forward():
        a = A.PwsConvArgs()
        a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, len(src_c), cout, 1
        for i, t in enumerate(srcs):
            a.src[i].ptr, a.src[i].channels, a.src[i].ld = t.data_ptr(), src_c[i], src_c[i]
        out = torch.full((n, oh, ow, cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        a.store, a.math, a.w_bf16 = A.STORE_BF16, A.MATH_BF16, wb.data_ptr()
        a.w_packed, a.bias, a.out, a.out_ld = wp.data_ptr(), d_b.data_ptr(), out.data_ptr(), cout
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "fwd")
        return out.float()

    def dgrad():
        da = A.PwsConvBwdDataArgs()
        da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
        da.gout, da.gout_ld, da.w_dgrad, da.ndst = d_dy.data_ptr(), cout, wdg.data_ptr(), len(src_c)
        da.math, da.w_dgrad_bf16, da.store = A.MATH_BF16, wdb.data_ptr(), A.STORE_BF16
        outs = []
        for i, c in enumerate(src_c):
            acc = 1 if i == 1 else 0   # the second destination accumulates into 0.5
            o = torch.full((n, h, w, c), 0.5 if acc else float("nan"), device="cuda", dtype=torch.bfloat16)
            outs.append(o)
            da.dst[i].ptr, da.dst[i].channels, da.dst[i].ld, da.dst[i].accumulate = o.data_ptr(), c, c, acc
        da.ws, da.ws_bytes = ws.data_ptr(), ws.numel()
        A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "dgrad")
        return torch.cat([o.float() - (0.5 if i == 1 else 0.0) for i, o in enumerate(outs)], dim=3)

    def wgrad():
        wa = A.PwsConvBwdWeightArgs()
        wa.kind, wa.n, wa.h, wa.w, wa.nsrc, wa.cout, wa.math, wa.store = kind, n, h, w, len(src_c), cout, A.MATH_BF16, A.STORE_BF16
        for i, t in enumerate(srcs):
            wa.src[i].ptr, wa.src[i].channels, wa.src[i].ld = t.data_ptr(), src_c[i], src_c[i]
        dwp = torch.zeros(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
        wa.gout, wa.gout_ld, wa.dw_packed = d_dy.data_ptr(), cout, dwp.data_ptr()
        A.check(L.pws_conv2d_bwd_weight(ctypes.byref(wa), st), "wgrad")
        dw = torch.empty(tuple(wt.shape), device="cuda")
        A.check(L.pws_unpack_conv_weight(A.ptr(dwp), A.ptr(dw), kind, cin, cout, st), "unpack")
        return dw

    res = {}
    try:
        # product dispatch; first-generation conv kernel; ring kernel for every launch it covers (not only the long ones); no one-shot
        # kernel; no ring weight gradient; ring weight gradient for every launch it covers
        # (round 6) 186 / 187: the ring kernel's units of 256 / 128 pixels wherever a tile shape of that size divides the map (else the launch falls through
        # to the other kernels, which is the dispatch being swept too); 188: a shared layer's operand pairs as two weight-gradient launches
        for e in (0, 20, 21, 186, 187, 71, 80, 81):
            L.pws_set_option(100, e)
            res[e] = (forward(), dgrad(), wgrad()) if e in (0, 20) else (forward(), dgrad(), None) if e in (21, 186, 187, 71) else (None, None, wgrad())
    finally:
        L.pws_set_option(100, 0)
    torch.cuda.synchronize()
    for e, (fy, fdx, fdw) in res.items():
        if fy is not None:
            assert not torch.isnan(fy).any() and not torch.isnan(fdx).any(), e
            assert relerr(fy, want_y) < TOL, (e, "forward", relerr(fy, want_y))
            assert relerr(fdx, want_dx) < 2 * TOL, (e, "data gradient", relerr(fdx, want_dx))   # + one more rounding of the accumulated half
        if fdw is not None:
            assert relerr(fdw, want_dw) < 2e-4, (e, "weight gradient", relerr(fdw, want_dw))


def _cases32():
    rs = np.random.RandomState(777)
    out = []
    names = list(KINDS)
    for i in range(40):
        kname = names[i % 4]
        h = int(rs.choice([1, 2, 4, 4, 8, 8, 16, 16, 20, 32, 32, 48, 64]))
        w = int(rs.choice([1, 2, 4, 4, 8, 8, 16, 16, 24, 32, 32, 64, 64]))
        n = int(rs.choice([1, 2, 2, 4, 8, 8]))
        if n * h * w > 8192:
            n = max(1, 8192 // (h * w))
        nsrc = int(rs.choice([1, 1, 2, 3]))
        src_c = [int(rs.choice([16, 32, 32, 48, 64])) for _ in range(nsrc)]
        cout = int(rs.choice([32, 64, 64, 96, 128]))
        out.append((kname, (n, h, w), src_c, cout))
    return out


@pytest.mark.parametrize("kname,shape,src_c,cout", _cases32())
def test_fp32_dispatch_sweep(hip, kname, shape, src_c, cout):
    """The fp32 forward dispatch (configs[1]: Winograd ring / one-shot / fp32 ring / tiled kernels) on random shapes around its
    thresholds, against PyTorch-CPU and against the plain tiled kernel (PWS_OPT_EXPERIMENT 50: no Winograd ring, 70: no one-shot
    kernel, 22: no fp32 ring kernel, 23: fp32 ring kernel wherever it is covered)."""
    A = hip
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    kd, k, s_, p_ = KINDS[kname]
    n, h, w = shape
    cin = sum(src_c)
    g = torch.Generator().manual_seed(zlib.crc32(repr((kname, shape, src_c, cout, 32)).encode()))
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cin, cout, k, k) if kd == "convT" else (cout, cin, k, k), generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    conv = F.conv2d if kd == "conv" else F.conv_transpose2d
    want = nhwc(F.leaky_relu(conv(x.double(), wt.double(), b.double(), stride=s_, padding=p_), 0.2))
    oh, ow = want.shape[1], want.shape[2]
    d_w = wt.cuda()
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight(A.ptr(d_w), A.ptr(wp), kind, cin, cout, st), "pack")
    a = A.PwsConvArgs()
    keep = []
    if kname == "CONVT_K4S2":
        ww = torch.empty(L.pws_packed_wino_ct4_floats(cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wino_ct4(A.ptr(wp), A.ptr(ww), cin, cout, st), "pack_wino_ct4")
        a.w_wino = ww.data_ptr()
        keep.append(ww)
    elif kname != "CONV_K3S2":
        ww = torch.empty(L.pws_packed_wino_floats(cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wino(A.ptr(wp), A.ptr(ww), cin, cout, st), "pack_wino")
        a.w_wino = ww.data_ptr()
        keep.append(ww)
    if L.pws_packed_wring_floats(kind, cin, cout):
        wr = torch.empty(L.pws_packed_wring_floats(kind, cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), kind, cin, cout, st), "pack_wring")
        a.w_wring = wr.data_ptr()
        keep.append(wr)
    xs = nhwc(x).cuda()
    c0 = 0
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, len(src_c), cout, 1
    for i, c in enumerate(src_c):
        t = xs[..., c0:c0 + c].contiguous()
        keep.append(t)
        a.src[i].ptr, a.src[i].channels, a.src[i].ld = t.data_ptr(), c, c
        c0 += c
    d_b = b.cuda()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    a.w_packed, a.bias, a.out_ld, a.ws, a.ws_bytes = wp.data_ptr(), d_b.data_ptr(), cout, ws.data_ptr(), ws.numel()
    res = {}
    try:
        for e in (0, 50, 70, 22, 23):
            L.pws_set_option(100, e)
            out = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
            a.out = out.data_ptr()
            A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "fwd fp32")
            res[e] = out
    finally:
        L.pws_set_option(100, 0)
    torch.cuda.synchronize()
    for e, out in res.items():
        assert not torch.isnan(out).any(), e
        assert relerr(out, want) < 5e-5, (e, relerr(out, want))   # Winograd transforms: a few 1e-6 of max; reordered fp32 sums

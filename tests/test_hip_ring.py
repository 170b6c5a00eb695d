"""GPU parity of the persistent LDS-ring bf16 convolution kernel (csrc/conv_ring.hip) through pws_conv2d_fwd /
pws_conv2d_bwd_data with bf16 storage: every kind it covers (3x3 s1, transposed 3x3 s1, 3x3 s2 as parity planes, transposed 4x4 s2
as parity classes, and the data gradients of all four), virtual concats of up to four sources, output channel counts that end
inside a 64-channel block, scatter over several gradient destinations with overwrite / accumulate / fused act'(y).

Reference: PyTorch-CPU fp32 arithmetic on bf16-rounded operands (only the summation order differs) with the tolerance of a
bf16-stored result (tests/test_hip_bf16.py: STORE_TOL), and the first-generation kernel (conv_bf16_kernel) on the same launch.
PWS_OPT_EXPERIMENT 21 forces the ring kernel for launches too small to fill 256 persistent workgroups, 20 disables it."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu

from test_hip_bf16 import KINDS, STORE_TOL, bf16r, make_case, nhwc, relerr, torch_layer  # noqa: E402

PLANES = {"CONV_K3S1": 9, "CONV_K3S2": 9, "CONVT_K3S1": 9, "CONVT_K4S2": 16, "CONV_K5S1": 25}
# (kind, (n, h, w) of the layer INPUT, sources, cout): the logical map (output, or the class / plane grid) is a multiple of 16 x 32
FWD = [
    ("CONV_K3S1", (2, 16, 32), [32], 64), ("CONV_K3S1", (1, 32, 64), [64, 32], 96), ("CONV_K3S1", (3, 16, 64), [32, 64, 32, 32], 128),
    ("CONVT_K3S1", (2, 32, 32), [64], 64), ("CONVT_K3S1", (1, 16, 32), [32, 32], 200),
    ("CONV_K3S2", (2, 32, 64), [32, 32], 64), ("CONV_K3S2", (1, 64, 64), [64], 128), ("CONV_K3S2", (3, 32, 64), [32], 72),
    ("CONVT_K4S2", (2, 16, 32), [32, 64], 64), ("CONVT_K4S2", (1, 32, 32), [128, 64], 64), ("CONVT_K4S2", (2, 16, 32), [32], 40),
    # 16-wide maps: tiles of 16 x 16 x 2 samples; 8 x 8 maps: tiles of 8 x 8 x 8 samples (2x2-tap kinds)
    ("CONV_K3S1", (2, 16, 16), [64], 64), ("CONVT_K3S1", (4, 32, 16), [32, 32], 72), ("CONVT_K4S2", (4, 16, 16), [32, 32], 96),
    ("CONV_K3S2", (2, 32, 32), [64], 64), ("CONVT_K4S2", (8, 8, 8), [64], 64), ("CONV_K3S2", (8, 16, 16), [32], 64),
    # round 6: the first layer's kind (5x5 on a 32-channel NHWC source, weights resident in LDS, 8 x 32 tiles): borders on every side, several tiles per
    # workgroup, cout ending inside the block
    ("CONV_K5S1", (2, 16, 64), [32], 64), ("CONV_K5S1", (1, 24, 32), [32], 64), ("CONV_K5S1", (3, 8, 96), [32], 40), ("CONV_K5S1", (5, 64, 64), [32], 64),
]
# data gradients: (kind, (n, h, w) of the forward INPUT = extent of dx, forward sources = destinations, forward cout)
BWD = [
    ("CONV_K3S1", (2, 16, 32), [64], 64), ("CONV_K3S1", (1, 32, 64), [64, 32], 96), ("CONVT_K3S1", (2, 16, 32), [32, 64, 32], 64),
    ("CONV_K3S2", (2, 32, 64), [32, 32], 64), ("CONV_K3S2", (1, 64, 128), [64], 32), ("CONV_K3S2", (1, 32, 64), [128], 96),
    ("CONVT_K4S2", (2, 16, 32), [32, 64], 64), ("CONVT_K4S2", (1, 32, 32), [128, 64], 32), ("CONVT_K4S2", (1, 16, 64), [64], 96),
    ("CONV_K3S1", (2, 16, 16), [64], 64), ("CONVT_K4S2", (2, 16, 16), [64, 64], 32), ("CONV_K3S2", (2, 32, 32), [32, 32], 64),
    ("CONVT_K4S2", (8, 8, 8), [64], 64), ("CONV_K3S2", (8, 16, 16), [64], 32),
]

# Round 6: units of 256 pixels (tiles 16 x 16, 8 x 8 x 4 samples) and 128 pixels (8 x 8 x 2 samples, 4 x 4 x 8 samples) for the maps that give too few
# units of 512.  PWS_OPT_EXPERIMENT 186 / 187 force that unit size; shapes chosen so that exactly that tile shape applies (several tiles per map,
# cout ending inside a 64-channel block, virtual concats).
FWD_SMALL = [
    ("CONV_K3S1", (1, 16, 32), [32], 64, 186), ("CONVT_K3S1", (3, 16, 16), [64], 128, 186), ("CONVT_K4S2", (1, 16, 16), [64, 32], 72, 186),
    ("CONV_K3S2", (1, 32, 64), [32, 32], 64, 186),
    ("CONV_K3S1", (4, 8, 8), [64], 64, 186), ("CONV_K3S1", (8, 8, 24), [32], 40, 186), ("CONVT_K4S2", (4, 8, 8), [32, 32], 96, 186),
    ("CONV_K3S2", (4, 16, 16), [64], 64, 186),
    ("CONV_K3S1", (2, 8, 8), [64, 32], 64, 187), ("CONVT_K3S1", (6, 8, 16), [32], 72, 187), ("CONVT_K4S2", (2, 8, 8), [64], 64, 187),
    ("CONV_K3S2", (2, 16, 16), [32, 32], 128, 187),
    ("CONV_K3S1", (8, 4, 4), [64], 64, 187), ("CONVT_K3S1", (16, 4, 12), [32], 64, 187), ("CONVT_K4S2", (8, 4, 4), [32, 64], 64, 187),
    ("CONV_K3S2", (8, 8, 8), [64], 96, 187),
]
BWD_SMALL = [
    ("CONV_K3S1", (1, 16, 32), [64], 64, 186), ("CONVT_K3S1", (3, 16, 16), [32, 64, 32], 64, 186), ("CONV_K3S2", (1, 32, 64), [32, 32], 64, 186),
    ("CONVT_K4S2", (1, 16, 16), [64, 64], 32, 186),
    ("CONV_K3S1", (4, 8, 8), [64], 96, 186), ("CONV_K3S2", (4, 16, 16), [64], 32, 186), ("CONVT_K4S2", (4, 8, 8), [32, 64], 64, 186),
    ("CONV_K3S1", (2, 8, 8), [64, 32], 64, 187), ("CONV_K3S2", (2, 16, 16), [32, 32], 64, 187), ("CONVT_K4S2", (2, 8, 8), [64], 64, 187),
    ("CONV_K3S1", (8, 4, 4), [64], 64, 187), ("CONV_K3S2", (8, 8, 8), [128], 96, 187), ("CONVT_K4S2", (8, 4, 4), [32, 64], 64, 187),
]


@pytest.fixture()
def force(hip):
    L = hip.lib()

    def set_(v):
        assert L.pws_set_option(100, v) == 0
    yield set_
    set_(0)


def _fwd(A, kname, x, wt, b, act, src_c, cout):
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    n, cin, h, w = x.shape
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    d_w = wt.cuda()
    A.check(L.pws_pack_conv_weight(A.ptr(d_w), A.ptr(wp), kind, cin, cout, st), "pack")
    wb = torch.empty(L.pws_packed_bf16_floats(PLANES[kname], cin, cout), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), PLANES[kname], cin, cout, st), "pack_bf16")
    xs = nhwc(x)
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, len(src_c), cout, act
    keep, c0 = [], 0
    for i, c in enumerate(src_c):
        pad = 8 * (i % 2)      # a source whose pixel stride exceeds its channel count (a slice of a wider tensor)
        t = torch.zeros((n, h, w, c + pad), dtype=torch.bfloat16, device="cuda")
        t[..., :c] = xs[..., c0:c0 + c].cuda().bfloat16()
        keep.append(t)
        a.src[i].ptr, a.src[i].channels, a.src[i].ld = t.data_ptr(), c, c + pad
        c0 += c
    oh, ow = (h, w) if "S1" in kname else ((h // 2, w // 2) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    out = torch.full((n, oh, ow, cout + 8), float("nan"), device="cuda", dtype=torch.bfloat16)   # out_ld = cout + 8
    d_b = b.cuda()
    a.store, a.math, a.w_bf16 = A.STORE_BF16, A.MATH_BF16, wb.data_ptr()
    a.w_packed, a.bias, a.out, a.out_ld = wp.data_ptr(), d_b.data_ptr(), out.data_ptr(), cout + 8
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    L.pws_prof_enable(1)
    A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv")
    L.pws_prof_enable(0)
    names = [r[0] for r in A.prof_collect()]
    torch.cuda.synchronize()
    assert torch.isnan(out[..., cout:].float()).all(), "the kernel wrote beyond its cout channels"
    return out[..., :cout].float().cpu(), names


@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("kname,shape,src_c,cout,unit", [c + (21,) for c in FWD] + FWD_SMALL)
def test_ring_forward(hip, force, kname, shape, src_c, cout, unit, act):
    x, wt, b, _ = make_case(kname, shape, src_c, cout, "ring")
    xr, wr = bf16r(x), bf16r(wt)
    want = nhwc(torch_layer(kname, xr, wr, b, act)).numpy()
    force(unit)
    got, names = _fwd(hip, kname, xr, wt, b, act, src_c, cout)
    assert names == ["conv_ring_kernel"], names
    force(20)
    old, names = _fwd(hip, kname, xr, wt, b, act, src_c, cout)
    assert names == ["conv_bf16_kernel"] or (unit != 21 and "conv_ring_kernel" not in names), names   # (the smallest maps are the one-shot kernel's)
    got, old = got.numpy(), old.numpy()
    assert not np.isnan(got).any()
    e_ref, e_old = relerr(got, want), relerr(got, old)
    assert e_ref < STORE_TOL, e_ref
    assert e_old < 2 ** -7, e_old      # both round the same fp32 sums (another K order): at most one bf16 ulp apart


def _bwd(A, kname, d_dy, wt, shape, src_c, cout, ys, act, stale):
    L, st = A.lib(), A.current_stream()
    kind = getattr(A, kname)
    n, h, w = shape
    cin = sum(src_c)
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    d_w = wt.cuda()
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(d_w), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    planes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(planes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), planes, cout, cin, st), "pack_bf16")
    da = A.PwsConvBwdDataArgs()
    da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
    da.gout, da.gout_ld, da.w_dgrad, da.ndst = d_dy.data_ptr(), d_dy.shape[-1], wdg.data_ptr(), len(src_c)
    da.math, da.w_dgrad_bf16, da.store = A.MATH_BF16, wdb.data_ptr(), A.STORE_BF16
    outs = []
    for i, c in enumerate(src_c):
        acc = 1 if i == 1 else 0
        o = stale[i].clone() if acc else torch.full((n, h, w, c), float("nan"), device="cuda", dtype=torch.bfloat16)
        outs.append(o)
        da.dst[i].ptr, da.dst[i].channels, da.dst[i].ld, da.dst[i].accumulate = o.data_ptr(), c, c, acc
        if act and i != 2:
            da.dst[i].act_y, da.dst[i].act_y_ld, da.dst[i].act = ys[i].data_ptr(), ys[i].shape[-1], act
    ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    da.ws, da.ws_bytes = ws.data_ptr(), ws.numel()
    L.pws_prof_enable(1)
    A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "bwd_data")
    L.pws_prof_enable(0)
    names = [r[0] for r in A.prof_collect()]
    torch.cuda.synchronize()
    return [o.float().cpu() for o in outs], names


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("kname,shape,src_c,cout,unit", [c + (21,) for c in BWD] + BWD_SMALL)
def test_ring_data_gradient(hip, force, kname, shape, src_c, cout, unit, act):
    x, wt, b, rs = make_case(kname, shape, src_c, cout, "ringd")
    n, cin, h, w = x.shape
    wr = bf16r(wt)
    xg = x.clone().requires_grad_(True)
    kd, k, s_, p_ = KINDS[kname]
    y = (F.conv2d if kd == "conv" else F.conv_transpose2d)(xg, wr, None, stride=s_, padding=p_)
    dy = bf16r(torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)))
    y.backward(dy)
    ref = nhwc(xg.grad)
    d_dy = torch.zeros(tuple(nhwc(dy).shape[:3]) + (cout + 8,), dtype=torch.bfloat16, device="cuda")   # gout_ld = cout + 8
    d_dy[..., :cout] = nhwc(dy).cuda().bfloat16()
    ys = [torch.from_numpy(rs.standard_normal((n, h, w, c + 8)).astype(np.float32)).cuda().bfloat16() for c in src_c]
    stale = [torch.from_numpy(rs.standard_normal((n, h, w, c)).astype(np.float32)).cuda().bfloat16() for c in src_c]
    force(unit)
    got, names = _bwd(hip, kname, d_dy, wt, shape, src_c, cout, ys, act, stale)
    assert names == ["conv_ring_kernel"], names
    force(20)
    old, names = _bwd(hip, kname, d_dy, wt, shape, src_c, cout, ys, act, stale)
    assert names == ["conv_bf16_kernel"] or (unit != 21 and "conv_ring_kernel" not in names), names
    slope = {0: 1.0, 1: 0.2, 2: 0.0}[act]
    c0 = 0
    for i, c in enumerate(src_c):
        want = ref[..., c0:c0 + c].clone()
        if i == 1:
            want = want + stale[i].float().cpu()
        if act and i != 2:
            yv = ys[i][..., :c].float().cpu()
            want = want * torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, slope))
        assert not torch.isnan(got[i]).any()
        scale = float(ref.abs().max())
        err = float((got[i] - want).abs().max()) / scale
        assert err < 2 * STORE_TOL, (i, err)
        assert float((got[i] - old[i]).abs().max()) / scale < 2 ** -6, i
        c0 += c


def test_ring_whole_generator_matches_first_generation_kernels(hip, force):
    """Batch 8, ngf 64 (every 256^2 .. 32^2 layer qualifies for the ring kernel): the bf16-storage training forward and backward
    with and without it -- fields, loss and all gradients agree far inside the bf16 tolerance of either."""
    from pwstablenet_amd import synth
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
    net = net.cuda()
    net.module.set_math("bf16")
    x = torch.from_numpy(synth.make_window(8, 31, 256, seed=11)).cuda()
    tgt = torch.from_numpy(np.random.RandomState(5).standard_normal((8, 256, 256, 2)).astype(np.float32) * 0.1).cuda()
    res = {}
    for tag, exp in (("ring", 0), ("v1", 20)):
        force(exp)
        net.zero_grad(set_to_none=True)
        hip.lib().pws_prof_enable(1)
        grids, resid = net(x)
        loss = sum(((g - tgt) ** 2).mean() for g in grids) + sum((r ** 2).mean() for r in resid)
        loss.backward()
        hip.lib().pws_prof_enable(0)
        names = [r[0] for r in hip.prof_collect(1 << 16)]
        res[tag] = (names.count("conv_ring_kernel"), [g.detach().clone() for g in grids], float(loss), [p.grad.clone() for p in net.parameters()])
    assert res["ring"][0] >= 20 and res["v1"][0] == 0, (res["ring"][0], res["v1"][0])
    for a, b in zip(res["ring"][1], res["v1"][1]):
        assert float((a - b).abs().max()) < 2e-3
    assert abs(res["ring"][2] - res["v1"][2]) < 2e-3 * abs(res["v1"][2])
    dot = sum((a.double() * b.double()).sum().item() for a, b in zip(res["ring"][3], res["v1"][3]))
    na = sum((a.double() ** 2).sum().item() for a in res["ring"][3]) ** 0.5
    nb = sum((b.double() ** 2).sum().item() for b in res["v1"][3]) ** 0.5
    print("ring vs first-generation kernels: %d ring launches, loss %.6g vs %.6g, gradient cosine %.6f" % (
        res["ring"][0], res["ring"][2], res["v1"][2], dot / (na * nb)))
    assert dot / (na * nb) > 0.999

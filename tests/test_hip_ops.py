"""GPU parity tests: every HIP entry point, called through the C ABI (include/pwstable.h), against the CPU
oracle on the same seeded inputs and against the golden vectors generated from the reference.

Tolerances (fp32): convolutions accumulate K = taps*cin products in a different order than the oracle, so the
bound scales with sum|a*b|: 2e-5 * sqrt(K)-ish absolute at O(1) activations; stated per test.
"""
import ctypes
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from pwstablenet_amd import synth  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1))


def nchw(a):
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2))


def run_conv(A, kind, srcs_nhwc, w_torch, bias, act, cout, nchw_src=None, ws_mb=0, wino=False, nchw_channels=None):
    """srcs_nhwc: list of numpy NHWC arrays (virtual concat) or nchw_src: one NCHW array."""
    L = A.lib()
    st = A.current_stream()
    cin = w_torch.shape[1] if kind in (A.CONV_K3S1, A.CONV_K3S2, A.CONV_K5S1) else w_torch.shape[0]
    wt = dev(w_torch)
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda", dtype=torch.float32)
    A.check(L.pws_pack_conv_weight(A.ptr(wt), A.ptr(wp), kind, cin, cout, st), "pack")
    args = A.PwsConvArgs()
    args.kind = kind
    keep = []
    if nchw_src is not None:
        t = dev(nchw_src)
        keep.append(t)
        n, _, h, w = nchw_src.shape
        args.nsrc, args.src_nchw = 1, 1
        args.src[0].ptr, args.src[0].channels, args.src[0].ld = t.data_ptr(), nchw_src.shape[1], 0
        if nchw_channels is not None:   # the first `nchw_channels` planes of every sample, read in place (ld = sample stride)
            args.src[0].channels, args.src[0].ld = nchw_channels, nchw_src.shape[1] * h * w
    else:
        n, h, w, _ = srcs_nhwc[0].shape
        args.nsrc = len(srcs_nhwc)
        for i, s in enumerate(srcs_nhwc):
            t = dev(s)
            keep.append(t)
            args.src[i].ptr, args.src[i].channels, args.src[i].ld = t.data_ptr(), s.shape[3], s.shape[3]
    args.n, args.h, args.w = n, h, w
    oh, ow = h, w
    if kind == A.CONV_K3S2:
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if kind == A.CONVT_K4S2:
        oh, ow = 2 * h, 2 * w
    out = torch.full((n, oh, ow, cout), float("nan"), device="cuda", dtype=torch.float32)
    b = dev(bias) if bias is not None else None
    args.cout, args.w_packed, args.bias, args.act = cout, wp.data_ptr(), (b.data_ptr() if b is not None else None), act
    args.out, args.out_ld = out.data_ptr(), cout
    if wino and kind == A.CONV_K5S1:
        pass   # (only the ring-layout F(2x2,5x5) weights below)
    elif wino and kind == A.CONVT_K4S2:
        ww = torch.empty(L.pws_packed_wino_ct4_floats(cin, cout), device="cuda", dtype=torch.float32)
        A.check(L.pws_pack_conv_weight_wino_ct4(A.ptr(wp), A.ptr(ww), cin, cout, st), "pack_wino_ct4")
        args.w_wino = ww.data_ptr()
    elif wino:
        ww = torch.empty(L.pws_packed_wino_floats(cin, cout), device="cuda", dtype=torch.float32)
        A.check(L.pws_pack_conv_weight_wino(A.ptr(wp), A.ptr(ww), cin, cout, st), "pack_wino")
        args.w_wino = ww.data_ptr()
    if wino and L.pws_packed_wring_floats(kind, cin, cout):   # ring-layout Winograd weights (conv_wring.hip)
        wr = torch.empty(L.pws_packed_wring_floats(kind, cin, cout), device="cuda", dtype=torch.float32)
        A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), kind, cin, cout, st), "pack_wring")
        args.w_wring = wr.data_ptr()
    if ws_mb:
        ws = torch.empty(ws_mb << 20, device="cuda", dtype=torch.uint8)
        args.ws, args.ws_bytes = ws.data_ptr(), ws.numel()
    A.check(L.pws_conv2d_fwd(ctypes.byref(args), st), "pws_conv2d_fwd")
    torch.cuda.synchronize()
    return out.cpu().numpy()


CONV_CASES = [
    # kind name, (n,h,w), source channel list, cout
    ("CONV_K3S1", (2, 20, 37), [16], 64),
    ("CONV_K3S1", (1, 33, 16), [32, 16], 96),       # 2 sources, cout not a multiple of 64
    ("CONV_K3S1", (3, 8, 8), [16], 16),             # small-tile configs
    ("CONV_K3S1", (5, 4, 4), [32], 32),
    ("CONV_K3S1", (18, 2, 2), [16], 64),
    ("CONV_K3S2", (2, 40, 34), [16, 16], 64),
    ("CONV_K3S2", (2, 16, 16), [32], 32),
    ("CONV_K3S2", (3, 8, 8), [16], 32),
    ("CONV_K3S2", (17, 4, 4), [16, 16, 16], 16),
    ("CONV_K5S1", (1, 23, 40), [16], 64),
    ("CONVT_K3S1", (2, 18, 21), [16, 32], 48),
    ("CONVT_K3S1", (2, 4, 4), [16], 16),
    ("CONVT_K4S2", (2, 17, 19), [32, 16, 16], 64),
    ("CONVT_K4S2", (2, 8, 8), [16], 32),
    ("CONVT_K4S2", (3, 4, 4), [16, 16], 16),
    ("CONVT_K4S2", (20, 2, 2), [32], 64),
    # deep-layer shapes: K large, M tiny -> split-K path when a workspace is given
    ("CONV_K3S1", (2, 16, 16), [64, 64], 64),
    ("CONV_K3S1", (3, 4, 4), [128], 64),
    ("CONV_K3S2", (2, 16, 16), [64, 48], 128),
    ("CONV_K3S2", (4, 4, 4), [256], 64),
    ("CONVT_K3S1", (2, 2, 2), [256], 256),
    ("CONVT_K4S2", (2, 8, 8), [128, 64, 64], 64),
    ("CONVT_K4S2", (3, 2, 2), [256], 128),
]


@pytest.mark.parametrize("kname,shape,src_c,cout", CONV_CASES)
@pytest.mark.parametrize("act", [1, 2])
def test_conv_kinds_vs_oracle(hip, oracle, kname, shape, src_c, cout, act):
    A = hip
    kind = getattr(A, kname)
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout)).encode()))
    cin = sum(src_c)
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    k = {"CONV_K3S1": 3, "CONV_K3S2": 3, "CONV_K5S1": 5, "CONVT_K3S1": 3, "CONVT_K4S2": 4}[kname]
    is_t = kname.startswith("CONVT")
    wshape = (cin, cout, k, k) if is_t else (cout, cin, k, k)
    wt = (rs.standard_normal(wshape) / np.sqrt(cin * k * k / (4 if kname == "CONVT_K4S2" else 1))).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    s, p = {"CONV_K3S1": (1, 1), "CONV_K3S2": (2, 1), "CONV_K5S1": (1, 2), "CONVT_K3S1": (1, 1), "CONVT_K4S2": (2, 1)}[kname]
    oact = {1: oracle.ACT_LRELU, 2: oracle.ACT_RELU}[act]
    ref = (oracle.conv_transpose2d if is_t else oracle.conv2d)(x, wt, b, s, p, oact)
    xs = nhwc(x)
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(np.ascontiguousarray(xs[..., c0:c0 + c]))
        c0 += c
    for ws_mb in (0, 64):  # without and with split-K scratch
        got = run_conv(A, kind, srcs, wt, b, act, cout, ws_mb=ws_mb)
        assert not np.isnan(got).any(), "kernel left output elements unwritten"
        # outputs are O(1); K <= 2304 products of N(0,1)*N(0,1/K): 5e-5 abs covers the reordered fp32 sum
        np.testing.assert_allclose(nchw(got), ref, rtol=0, atol=5e-5)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv_epilogue_activation_on_non_finite_values(hip, oracle, act):
    """The epilogues' activation is branch-free arithmetic (common.h: act_apply = max(v, (slope v) & keep)); it must treat
    +-inf, NaN and +-0 exactly like the reference's LeakyReLU(0.2) / ReLU / identity.  A bias of +-inf / NaN / 0 on a zero input makes
    the pre-activation exactly that value."""
    A = hip
    cout, cin = 16, 16
    x = np.zeros((1, 8, 8, cin), np.float32)
    wt = np.ones((cout, cin, 3, 3), np.float32)
    b = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1.5, -1.5, 3e38, -3e38, 1e-40, -1e-40, 2.0, -2.0, np.inf, -np.inf, np.nan], np.float32)
    got = run_conv(A, A.CONV_K3S1, [x], wt, b, act, cout)[0, 4, 4]
    with np.errstate(invalid="ignore", over="ignore"):
        want = b.copy() if act == 0 else (np.where(b > 0, b, np.float32(0.2) * b) if act == 1 else np.where(b > 0, b, np.float32(0.0)))
    want = want.astype(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    fin = ~np.isnan(want)
    # flush-to-zero of the 1e-40 denormal by the matrix pipeline's bias add is allowed: compare with denormals flushed
    flush = lambda a: np.where(np.abs(a) < np.float32(1.2e-38), np.float32(0) * a, a)
    np.testing.assert_array_equal(flush(got[fin]), flush(want[fin]))


RINGF_CASES = [
    ("CONV_K3S1", (2, 16, 32), [16], 64), ("CONV_K3S1", (1, 24, 64), [32, 16], 96), ("CONVT_K3S1", (2, 16, 32), [16, 16, 16, 16], 36),
    ("CONV_K3S2", (2, 32, 64), [16, 16], 64), ("CONV_K3S2", (1, 48, 64), [32], 72), ("CONVT_K4S2", (2, 16, 32), [32, 16], 64),
    ("CONVT_K4S2", (1, 8, 64), [16], 40),
]


@pytest.mark.parametrize("force", [23, 24, 27])
@pytest.mark.parametrize("kname,shape,src_c,cout", RINGF_CASES)
@pytest.mark.parametrize("act", [1, 2])
def test_fp32_ring_kernel_vs_oracle(hip, oracle, kname, shape, src_c, cout, act, force):
    """conv_ringf_kernel (csrc/conv_ring_f32.hip: persistent LDS-ring, exact fp32 MFMA) forced for small launches with both tile
    heights (PWS_OPT_EXPERIMENT 23 / 24; 27: the stride-2 kind's units of 32 output channels), against the C oracle and against conv_mfma_kernel (22): 3x3 s1, transposed 3x3 s1,
    3x3 s2 as parity planes, transposed 4x4 s2 as parity classes, virtual concats, cout ending inside a 64-channel block."""
    A = hip
    L = A.lib()
    if force == 27 and kname != "CONV_K3S2":
        pytest.skip("32-channel units exist for the stride-2 kind only")
    kind = getattr(A, kname)
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout, "rf")).encode()))
    cin = sum(src_c)
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    k = {"CONV_K3S1": 3, "CONV_K3S2": 3, "CONVT_K3S1": 3, "CONVT_K4S2": 4}[kname]
    is_t = kname.startswith("CONVT")
    wt = (rs.standard_normal((cin, cout, k, k) if is_t else (cout, cin, k, k)) / np.sqrt(cin * k * k / (4 if kname == "CONVT_K4S2" else 1))).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    s_, p_ = {"CONV_K3S1": (1, 1), "CONV_K3S2": (2, 1), "CONVT_K3S1": (1, 1), "CONVT_K4S2": (2, 1)}[kname]
    ref = (oracle.conv_transpose2d if is_t else oracle.conv2d)(x, wt, b, s_, p_, {1: oracle.ACT_LRELU, 2: oracle.ACT_RELU}[act])
    xs = nhwc(x)
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(np.ascontiguousarray(xs[..., c0:c0 + c]))
        c0 += c
    got = {}
    try:
        for exp in (force, 22):
            assert L.pws_set_option(100, exp) == 0
            L.pws_prof_enable(1)
            got[exp] = run_conv(A, kind, srcs, wt, b, act, cout, ws_mb=64)
            L.pws_prof_enable(0)
            names = [r[0] for r in A.prof_collect()]
            assert (names == ["conv_ringf_kernel"]) == (exp != 22), (exp, names)
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    assert not np.isnan(got[force]).any()
    np.testing.assert_allclose(nchw(got[force]), ref, rtol=0, atol=5e-5)
    np.testing.assert_allclose(got[force], got[22], rtol=0, atol=5e-5)


@pytest.mark.parametrize("kname,shape,src_c,cout", [
    ("CONV_K3S1", (4, 128, 128), [64, 64], 64),
    ("CONVT_K3S1", (5, 112, 130), [128], 96),     # ragged extent, cout not a multiple of 64
    ("CONV_K3S1", (16, 64, 64), [64, 32, 32], 128),
    ("CONV_K3S1", (8, 256, 256), [16], 64),       # few input channels: taken only because the map is huge
    ("CONVT_K4S2", (4, 64, 64), [128, 64], 64),   # transposed conv k4 s2: F(3x3,2x2) per output parity class
    ("CONVT_K4S2", (3, 50, 77), [64], 96),        # extents not multiples of 3 / of the 12x24 workgroup, ragged cout
    ("CONVT_K4S2", (8, 128, 128), [32, 32], 64),
])
def test_conv_winograd_vs_oracle(hip, oracle, kname, shape, src_c, cout):
    """3x3 stride-1 layers through the Winograd F(2x2,3x3) kernel and k4 s2 transposed convs through F(3x3,2x2)
    (taken when their workgroups fill the chip)."""
    A = hip
    kind = getattr(A, kname)
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout, "w")).encode()))
    cin = sum(src_c)
    is_t = kname.startswith("CONVT")
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    k = 4 if kname == "CONVT_K4S2" else 3
    wt = (rs.standard_normal((cin, cout, k, k) if is_t else (cout, cin, k, k)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    ref = (oracle.conv_transpose2d if is_t else oracle.conv2d)(x, wt, b, 2 if k == 4 else 1, 1, oracle.ACT_LRELU)
    xs = nhwc(x)
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(np.ascontiguousarray(xs[..., c0:c0 + c]))
        c0 += c
    direct = run_conv(A, kind, srcs, wt, b, 1, cout)
    got = run_conv(A, kind, srcs, wt, b, 1, cout, wino=True)
    assert not np.isnan(got).any()
    assert not np.array_equal(got, direct), "the Winograd kernel was not taken"
    np.testing.assert_allclose(nchw(got), ref, rtol=0, atol=5e-5)  # Winograd rounding: a few 1e-6 at O(1) outputs


@pytest.mark.parametrize("kname,shape,src_c,cout", [
    ("CONV_K3S1", (2, 32, 64), [32, 16], 64),     # virtual concat of two sources, 3 chunks, several units per image
    ("CONVT_K3S1", (3, 16, 32), [64], 32),        # one unit per image (all four borders inside the unit), flipped taps
    ("CONV_K3S1", (1, 48, 96), [16], 96),         # one chunk per unit: every barrier is a unit boundary
    ("CONV_K3S1", (8, 64, 64), [128], 128),       # >= 256 units: several units per workgroup (the product's regime)
    ("CONVT_K4S2", (2, 16, 32), [64], 32),        # transposed k4 s2 as F(2x2,2x2) per parity class: one tile block per image
    ("CONVT_K4S2", (1, 32, 64), [32, 16], 64),    # virtual concat, several tile blocks
    ("CONVT_K4S2", (8, 32, 32), [128], 128),      # 256 units
    ("CONV_K3S1", (4, 16, 16), [64], 64),         # 16-pixel-wide map: a unit = 16 x 16 pixels of two samples
    ("CONVT_K4S2", (2, 32, 16), [32, 32], 32),    # ... transposed kind, two row blocks, virtual concat
    ("CONV_K3S1", (2, 16, 32), [128], 64),        # K split over units (8 chunks -> 2 x 4), partial sums through the workspace
    ("CONVT_K4S2", (4, 16, 16), [256], 64),       # 16-wide + K split (16 chunks -> 4 x 4) + both class variants
    ("CONVT_K3S1", (6, 16, 16), [96, 32], 32),    # 16-wide, K split 2 x 4 across the source boundary of a virtual concat
])
def test_conv_winograd_ring_vs_oracle(hip, oracle, kname, shape, src_c, cout):
    """3x3 stride-1 layers (F(2x2,3x3)) and transposed k4 s2 layers (F(2x2,2x2) per parity class) through the persistent LDS-ring
    Winograd kernel (conv_wring.hip; PWS_OPT_EXPERIMENT 58 takes it whatever the number of units, 50 switches it off) against the C
    oracle and the kernels that run otherwise."""
    A = hip
    L = A.lib()
    kind = getattr(A, kname)
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout, "wr")).encode()))
    cin = sum(src_c)
    is_t = kname.startswith("CONVT")
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    k = 4 if kname == "CONVT_K4S2" else 3
    wt = (rs.standard_normal((cin, cout, k, k) if is_t else (cout, cin, k, k)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    ref = (oracle.conv_transpose2d if is_t else oracle.conv2d)(x, wt, b, 2 if k == 4 else 1, 1, oracle.ACT_LRELU)
    xs = nhwc(x)
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(np.ascontiguousarray(xs[..., c0:c0 + c]))
        c0 += c
    ring_name = "wino_ring_kernel<convT4,F(2x2,2x2)>" if k == 4 else "wino_ring_kernel<F(2x2,3x3)>"
    got = {}
    try:
        for force in (58, 59, 50) if k == 4 else (58, 50):   # 59: one parity class per unit (the variant for launches of few units)
            L.pws_set_option(100, force)
            L.pws_prof_enable(1)
            got[force] = run_conv(A, kind, srcs, wt, b, 1, cout, wino=True, ws_mb=64)
            L.pws_prof_enable(0)
            names = [r[0] for r in A.prof_collect()]
            assert (ring_name in names) == (force != 50), names
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    assert not np.isnan(got[58]).any()
    np.testing.assert_allclose(nchw(got[58]), ref, rtol=0, atol=5e-5)
    np.testing.assert_allclose(got[58], got[50], rtol=0, atol=5e-5)
    if 59 in got:
        np.testing.assert_allclose(nchw(got[59]), ref, rtol=0, atol=5e-5)


@pytest.mark.parametrize("kname,shape,src_c,cout", [
    ("CONV_K3S1", (8, 4, 4), [512], 512),          # M = 128: two M tiles per wave; 32 chunks -> one per workgroup
    ("CONVT_K3S1", (8, 2, 2), [64, 32], 80),       # M = 32 (2 x 2 wave grid), virtual concat, cout not a multiple of 64 / 16
    ("CONV_K3S2", (8, 2, 2), [512, 512], 512),     # 2x2 -> 1x1: M = 8, 64 chunks
    ("CONV_K3S2", (8, 8, 8), [128, 128], 96),      # 8x8 -> 4x4: M = 128
    ("CONV_K3S2", (3, 5, 7), [48], 32),            # ragged extent: 3x4 outputs per sample, M = 36 (padding pixels in the last tile)
    ("CONVT_K4S2", (8, 1, 1), [512], 512),         # 1x1 -> 2x2: M = 8 per parity class
    ("CONVT_K4S2", (8, 4, 4), [512, 512], 256),    # 4x4 -> 8x8: M = 128 per class, 8 chunks of 2x2 taps per workgroup
    ("CONVT_K4S2", (2, 3, 5), [32, 16], 20),       # ragged, cout % 16 != 0
])
def test_conv_skinny_vs_oracle(hip, oracle, kname, shape, src_c, cout):
    """The deep levels' one-shot kernel (conv_skinny.hip: at most 128 output pixels per parity class, all weights of a workgroup
    in one LDS-DMA burst, split-K through the workspace) against the C oracle and conv_mfma_kernel's small tiles."""
    A = hip
    L = A.lib()
    kind = getattr(A, kname)
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((kname, shape, cout, "sk")).encode()))
    cin = sum(src_c)
    is_t = kname.startswith("CONVT")
    k = 4 if kname == "CONVT_K4S2" else 3
    stride = 2 if kname in ("CONVT_K4S2", "CONV_K3S2") else 1
    x = rs.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rs.standard_normal((cin, cout, k, k) if is_t else (cout, cin, k, k)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    ref = (oracle.conv_transpose2d if is_t else oracle.conv2d)(x, wt, b, stride, 1, oracle.ACT_LRELU)
    xs = nhwc(x)
    srcs, c0 = [], 0
    for c in src_c:
        srcs.append(np.ascontiguousarray(xs[..., c0:c0 + c]))
        c0 += c
    got = {}
    try:
        for force in (0, 70):
            L.pws_set_option(100, force)
            L.pws_prof_enable(1)
            got[force] = run_conv(A, kind, srcs, wt, b, 1, cout, ws_mb=64)
            L.pws_prof_enable(0)
            names = [r[0] for r in A.prof_collect()]
            assert ("conv_skinny_kernel" in names) == (force == 0), names
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    assert not np.isnan(got[0]).any()
    np.testing.assert_allclose(nchw(got[0]), ref, rtol=0, atol=5e-5)
    np.testing.assert_allclose(got[0], got[70], rtol=0, atol=5e-5)


def test_conv_first_layer_nchw_31ch(hip, oracle):
    """transfer: k5 s1 p2 on the reference's NCHW 31-channel window (lib/networks_cascading.py:112,153)."""
    A = hip
    rs = np.random.RandomState(5)
    x = rs.standard_normal((2, 31, 37, 50)).astype(np.float32)
    wt = (rs.standard_normal((64, 31, 5, 5)) / 28).astype(np.float32)
    b = rs.standard_normal((64,)).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, 1, 2, oracle.ACT_LRELU)
    got = run_conv(A, A.CONV_K5S1, None, wt, b, 1, 64, nchw_src=x)
    np.testing.assert_allclose(nchw(got), ref, rtol=0, atol=5e-5)


FIRST_CASES = [   # (n, h, w), cin, cout, planes per sample in memory (None = dense), forced (PWS_OPT_EXPERIMENT 29: small launches)
    ((1, 256, 256), 31, 64, None, False),    # the generator's window: 256 units, the product dispatch
    ((2, 128, 256), 31, 64, 34, False),      # the driver's 34-plane item read in place
    ((2, 16, 32), 31, 64, None, True), ((1, 8, 64), 17, 64, None, True), ((3, 24, 32), 32, 20, None, True),
    ((1, 40, 96), 31, 128, 36, True), ((5, 8, 32), 20, 4, None, True),
    ((1, 8, 32), 31, 64, None, True),        # ONE unit: every image border inside the same halo
    ((5, 264, 32), 31, 64, None, False),     # one tile column: left and right border in every unit; 165 units on 256 CUs
    ((9, 64, 96), 32, 64, None, False),      # 216 units: workgroups with and without a next unit to prefetch
]


@pytest.mark.parametrize("shape,cin,cout,planes,forced", FIRST_CASES)
@pytest.mark.parametrize("act", [1, 2])
def test_first_layer_planar_kernel_vs_oracle(hip, oracle, shape, cin, cout, planes, forced, act):
    """conv_first_kernel (csrc/conv_first.hip: persistent, planar NCHW halo in LDS by LDS-DMA, exact fp32 MFMA) against the C oracle
    and against conv_mfma_kernel<k5s1, NCHW> (PWS_OPT_EXPERIMENT 25): image borders on every side, the zero padding channel,
    cout ending inside / spanning 64-channel blocks, samples read in place out of wider items."""
    A = hip
    L = A.lib()
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((shape, cin, cout, planes)).encode()))
    xfull = rs.standard_normal((n, planes or cin, h, w)).astype(np.float32)
    x = xfull[:, :cin]
    wt = (rs.standard_normal((cout, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    ref = oracle.conv2d(np.ascontiguousarray(x), wt, b, 1, 2, {1: oracle.ACT_LRELU, 2: oracle.ACT_RELU}[act])
    got = {}
    try:
        for exp in ((29 if forced else 0), 25):
            assert L.pws_set_option(100, exp) == 0
            L.pws_prof_enable(1)
            got[exp] = run_conv(A, A.CONV_K5S1, None, wt, b, act, cout, nchw_src=xfull, nchw_channels=cin if planes else None)
            L.pws_prof_enable(0)
            names = [r[0] for r in A.prof_collect()]
            assert names == (["conv_mfma_kernel<k5s1,16x16>"] if exp == 25 else ["conv_first_kernel"]), (exp, names)
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    new = got[29 if forced else 0]
    assert not np.isnan(new).any(), "kernel left output elements unwritten"
    np.testing.assert_allclose(nchw(new), ref, rtol=0, atol=5e-5)
    np.testing.assert_allclose(new, got[25], rtol=0, atol=5e-5)


WINO5_CASES = [   # (n, h, w), cin, planes per sample in memory (None = dense), forced (PWS_OPT_EXPERIMENT 30: launches of fewer than 2 units per CU)
    ((8, 256, 256), 31, None, False),     # BASELINE configs[1]'s batch: 4 096 units, the product dispatch
    ((2, 256, 256), 31, 34, False),       # the driver's 34-plane item read in place: 1 024 units
    ((1, 8, 16), 31, None, True),         # ONE unit: every image border inside the same halo, one k-step sequence
    ((1, 16, 48), 17, None, True), ((3, 24, 32), 32, None, True), ((2, 40, 96), 20, 36, True),
    ((5, 264, 16), 31, None, True),       # one unit column: left and right border in every unit; workgroups with 1 and 2 units
    ((1, 64, 64), 31, None, True),        # 32 units on 256 CUs: most workgroups idle
]


@pytest.mark.parametrize("shape,cin,planes,forced", WINO5_CASES)
@pytest.mark.parametrize("act", [1, 2])
def test_first_layer_winograd_kernel_vs_oracle(hip, oracle, shape, cin, planes, forced, act):
    """wino5_first_kernel (csrc/conv_first_wino.hip: F(2x2,5x5) in exact fp32, transformed input shared through LDS, 36 component accumulators
    per wave) against the C oracle and against conv_first_kernel (PWS_OPT_EXPERIMENT 26): image borders on every side, the zero padding
    channels, samples read in place out of wider items, launches of one unit and of several units per workgroup."""
    A = hip
    L = A.lib()
    n, h, w = shape
    rs = np.random.RandomState(zlib.crc32(repr((shape, cin, planes, "w5")).encode()))
    xfull = rs.standard_normal((n, planes or cin, h, w)).astype(np.float32)
    x = xfull[:, :cin]
    wt = (rs.standard_normal((64, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    b = rs.standard_normal((64,)).astype(np.float32)
    small = n * h * w <= 2 * 256 * 256
    ref = oracle.conv2d(np.ascontiguousarray(x), wt, b, 1, 2, {1: oracle.ACT_LRELU, 2: oracle.ACT_RELU}[act]) if small else None
    got = {}
    try:
        for exp in ((30 if forced else 0), 26):
            assert L.pws_set_option(100, exp) == 0
            L.pws_prof_enable(1)
            got[exp] = run_conv(A, A.CONV_K5S1, None, wt, b, act, 64, nchw_src=xfull, nchw_channels=cin if planes else None, wino=True)
            L.pws_prof_enable(0)
            names = [r[0] for r in A.prof_collect()]
            if exp != 26:
                assert names == ["wino5_first_kernel"], (exp, names)
            else:
                assert names[0] in ("conv_first_kernel", "conv_mfma_kernel<k5s1,16x16>"), names
    finally:
        L.pws_prof_enable(0)
        L.pws_set_option(100, 0)
    new = got[30 if forced else 0]
    assert not np.isnan(new).any(), "kernel left output elements unwritten"
    if ref is not None:
        np.testing.assert_allclose(nchw(new), ref, rtol=0, atol=5e-5)
    np.testing.assert_allclose(new, got[26], rtol=0, atol=5e-5)


def test_conv_empty_batch_and_bad_args(hip):
    A = hip
    args = A.PwsConvArgs()
    args.kind, args.n, args.h, args.w, args.nsrc, args.cout, args.out_ld = A.CONV_K3S1, 1, 4, 4, 0, 16, 16
    assert A.lib().pws_conv2d_fwd(ctypes.byref(args), None) == -22  # PWS_EINVAL
    assert b"nsrc" in A.lib().pws_last_error()


@pytest.mark.parametrize("tag", ["gs_small", "gs_mid", "gs_gray"])
@pytest.mark.parametrize("ac", [0, 1])
def test_grid_sample_golden(hip, ops_golden, tag, ac):
    """F.grid_sample fwd + bwd known-answer vectors from torch (OOB, exact-integer, half-integer coordinates)."""
    A, g = hip, ops_golden
    L, st = A.lib(), A.current_stream()
    img, grid, gout = dev(g[tag + "_img"]), dev(g[tag + "_grid"]), dev(g[tag + "_gout"])
    n, c, h, w = img.shape
    ho, wo = grid.shape[1], grid.shape[2]
    out = torch.empty((n, c, ho, wo), device="cuda")
    A.check(L.pws_grid_sample_fwd(A.ptr(img), A.ptr(grid), A.ptr(out), n, c, h, w, ho, wo, ac, st), "fwd")
    gi, gg = torch.empty_like(img), torch.empty_like(grid)
    A.check(L.pws_grid_sample_bwd(A.ptr(gout), A.ptr(img), A.ptr(grid), A.ptr(gi), A.ptr(gg), n, c, h, w, ho, wo, ac, st), "bwd")
    torch.cuda.synchronize()
    # coordinate conditioning, see tests/test_oracle_golden.py::test_grid_sample_fwd_bwd
    np.testing.assert_allclose(out.cpu().numpy(), g["%s_out_ac%d" % (tag, ac)], rtol=0, atol=2e-5)
    np.testing.assert_allclose(gi.cpu().numpy(), g["%s_ginput_ac%d" % (tag, ac)], rtol=0, atol=2e-5)
    np.testing.assert_allclose(gg.cpu().numpy(), g["%s_ggrid_ac%d" % (tag, ac)], rtol=0, atol=1e-3)


@pytest.mark.parametrize("shape", [(3, 3, 256, 256), (2, 1, 64, 100), (1, 3, 37, 53)])
def test_grid_sample_vs_oracle(hip, oracle, shape):
    A = hip
    n, c, h, w = shape
    img = synth.make_frames(n, c, h, w, seed=3)
    rs = np.random.RandomState(4)
    theta = (np.array([1, 0, 0, 0, 1, 0], np.float32) + 0.1 * rs.standard_normal((n, 6))).astype(np.float32)
    grid = oracle.affine_grid(theta, h, w) + 0.02 * rs.standard_normal((n, h, w, 2)).astype(np.float32)
    gout = rs.standard_normal((n, c, h, w)).astype(np.float32)
    ref = oracle.grid_sample_fwd(img, grid)
    rgi, rgg = oracle.grid_sample_bwd(gout, img, grid)
    L, st = A.lib(), A.current_stream()
    d_img, d_grid, d_gout = dev(img), dev(grid), dev(gout)
    out, gi, gg = torch.empty_like(d_img), torch.empty_like(d_img), torch.empty_like(d_grid)
    A.check(L.pws_grid_sample_fwd(A.ptr(d_img), A.ptr(d_grid), A.ptr(out), n, c, h, w, h, w, 0, st), "fwd")
    A.check(L.pws_grid_sample_bwd(A.ptr(d_gout), A.ptr(d_img), A.ptr(d_grid), A.ptr(gi), A.ptr(gg), n, c, h, w, h, w, 0, st), "bwd")
    torch.cuda.synchronize()
    # frames are 0..255; coordinate rounding is ~3e-5 px at W=256 and the zero padding makes a 0 -> ~200 step at the
    # image border, so border pixels differ by up to 255 * 3e-5 ~ 8e-3 (6e-5 on the [-1,1] scale; bound there 1e-3)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-2)
    np.testing.assert_allclose(gi.cpu().numpy(), rgi, rtol=0, atol=1e-4)
    np.testing.assert_allclose(gg.cpu().numpy(), rgg, rtol=1e-4, atol=0.5)  # terms O(C*W/2*255)
    # grad-only variants
    gg2 = torch.empty_like(d_grid)
    A.check(L.pws_grid_sample_bwd(A.ptr(d_gout), A.ptr(d_img), A.ptr(d_grid), None, A.ptr(gg2), n, c, h, w, h, w, 0, st), "bwd")
    torch.cuda.synchronize()
    # (the field-only call runs grid_sample_bwd_field_kernel where the shape allows: same terms, another summation order)
    np.testing.assert_allclose(gg2.cpu().numpy(), rgg, rtol=1e-4, atol=0.5)
    np.testing.assert_allclose(gg2.cpu().numpy(), gg.cpu().numpy(), rtol=1e-5, atol=2e-2)


@pytest.mark.parametrize("ac", [0, 1])
@pytest.mark.parametrize("shape", [(2, 3, 64, 96, 64, 96), (1, 1, 40, 52, 24, 36), (3, 3, 17, 33, 8, 10), (1, 3, 256, 256, 256, 256)])
def test_grid_sample_field_gradient_kernel(hip, oracle, shape, ac):
    """The 4-pixel-per-lane field-gradient kernel (frame = data, reference main_new.py:106-118): fields that leave the image on
    every side, exact-integer and half-integer coordinates, output size != input size, both coordinate conventions; against the C
    oracle and against the general backward kernel (PWS_OPT_EXPERIMENT 2)."""
    A = hip
    n, c, h, w, ho, wo = shape
    rs = np.random.RandomState(11 + ac)
    img = synth.make_frames(n, c, h, w, seed=8)
    theta = (np.array([1.2, 0.1, 0, -0.1, 1.2, 0], np.float32) + 0.05 * rs.standard_normal((n, 6))).astype(np.float32)
    grid = oracle.affine_grid(theta, ho, wo, align_corners=bool(ac)) + 0.03 * rs.standard_normal((n, ho, wo, 2)).astype(np.float32)
    grid[:, 0, :4] = np.array([[-1, -1], [1, 1], [0, 0], [1.0 - 1.0 / w, -1.0 + 1.0 / h]], np.float32)   # corners / centre / pixel centres
    grid[:, 1, :2] = np.array([[-1.5, 0.3], [0.2, 1.7]], np.float32)                                    # fully outside
    gout = rs.standard_normal((n, c, ho, wo)).astype(np.float32)
    _, rgg = oracle.grid_sample_bwd(gout, img, grid, align_corners=bool(ac), want_input=False)
    L, st = A.lib(), A.current_stream()
    d_img, d_grid, d_gout = dev(img), dev(grid), dev(gout)
    got = {}
    for exp in (0, 2):
        assert L.pws_set_option(100, exp) == 0
        gg = torch.full_like(d_grid, float("nan"))
        L.pws_prof_enable(1)
        A.check(L.pws_grid_sample_bwd(A.ptr(d_gout), A.ptr(d_img), A.ptr(d_grid), None, A.ptr(gg), n, c, h, w, ho, wo, ac, st), "bwd")
        L.pws_prof_enable(0)
        assert [r[0] for r in A.prof_collect()] == ["grid_sample_bwd_kernel"]
        torch.cuda.synchronize()
        got[exp] = gg.cpu().numpy()
    L.pws_set_option(100, 0)
    assert not np.isnan(got[0]).any()
    scale = np.abs(rgg).max()
    assert np.abs(got[0] - rgg).max() < 2e-5 * scale + 1e-3, np.abs(got[0] - rgg).max() / scale
    assert np.abs(got[0] - got[2]).max() < 2e-5 * scale + 1e-3


def test_grid_sample_linearity_full_size(hip):
    """Size-independent property at the bench size: the warp is linear in the frame and, for the identity field
    at pixel centres, the identity map."""
    A = hip
    L, st = A.lib(), A.current_stream()
    n, c, h, w = 8, 3, 256, 256
    a, b = torch.rand((n, c, h, w), device="cuda"), torch.rand((n, c, h, w), device="cuda")
    theta = torch.tensor([1, 0, 0, 0, 1, 0], device="cuda", dtype=torch.float32).repeat(n, 1)
    grid = torch.empty((n, h, w, 2), device="cuda")
    A.check(L.pws_affine_grid(A.ptr(theta), A.ptr(grid), n, h, w, 0, st), "affine")
    oa, ob, oab = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    A.check(L.pws_grid_sample_fwd(A.ptr(a), A.ptr(grid), A.ptr(oa), n, c, h, w, h, w, 0, st), "fwd")
    assert (oa - a).abs().max().item() < 1e-4  # identity field reproduces the frame
    g2 = grid + 0.05 * torch.randn_like(grid)
    ab = (2 * a + 3 * b).contiguous()
    for src, dst in ((a, oa), (b, ob), (ab, oab)):
        A.check(L.pws_grid_sample_fwd(A.ptr(src), A.ptr(g2), A.ptr(dst), n, c, h, w, h, w, 0, st), "fwd")
    assert (oab - (2 * oa + 3 * ob)).abs().max().item() < 1e-5


def test_affine_grid_golden(hip, ops_golden):
    A, g = hip, ops_golden
    theta = dev(g["ag_theta"])
    for ac in (0, 1):
        out = torch.empty((3, 12, 20, 2), device="cuda")
        A.check(A.lib().pws_affine_grid(A.ptr(theta), A.ptr(out), 3, 12, 20, ac, A.current_stream()), "affine")
        np.testing.assert_allclose(out.cpu().numpy(), g["ag_out_ac%d" % ac], rtol=0, atol=1e-6)


def test_upsample_and_fused_720p_golden(hip, ops_golden):
    """main_new.py:706-716: UpsamplingBilinear2d(720,1280) of the field, then grid_sample; separate and fused."""
    A, g = hip, ops_golden
    L, st = A.lib(), A.current_stream()
    field = dev(g["up_field"])  # 1,256,256,2
    f_nchw = field.permute(0, 3, 1, 2).contiguous()
    up = torch.empty((1, 2, 720, 1280), device="cuda")
    A.check(L.pws_upsample_bilinear_ac(A.ptr(f_nchw), A.ptr(up), 1, 2, 256, 256, 720, 1280, st), "upsample")
    up_nhwc = up.permute(0, 2, 3, 1).contiguous()
    np.testing.assert_allclose(up_nhwc.cpu().numpy()[:, ::9, ::16], g["up_out_sub"], rtol=0, atol=2e-6)
    frame = dev(synth.make_frames(1, 3, 720, 1280, seed=99))
    w1, w2 = torch.empty_like(frame), torch.empty_like(frame)
    A.check(L.pws_grid_sample_fwd(A.ptr(frame), A.ptr(up_nhwc), A.ptr(w1), 1, 3, 720, 1280, 720, 1280, 0, st), "fwd")
    A.check(L.pws_upsample_grid_sample_fwd(A.ptr(frame), A.ptr(field), A.ptr(w2), 1, 3, 720, 1280, 256, 256, 0, st), "fused")
    torch.cuda.synchronize()
    # 0..255 frames.  One ulp of the field at |g|~1 is 1.2e-7 = 7.6e-5 px at W=1280; interior pixels (smooth image,
    # |grad| <~ 3/px) therefore agree to ~1e-3, but a pixel whose taps straddle the zero padding sees a 0 -> ~200 step
    # and moves by 200 * (a few ulp) ~ 0.05.  Bound: north_star's 1e-3 on the [-1,1] scale (0.1275 here) everywhere,
    # and 1e-2 (8e-5 scaled) for all but a handful of border samples.
    for wgot in (w1, w2):
        d = np.abs(wgot.cpu().numpy()[:, :, ::9, ::16] - g["up_warp_sub"])
        assert d.max() < 0.1275, d.max()
        assert (d > 1e-2).mean() < 2e-3
    assert (w1 - w2).abs().max().item() < 0.1275


def test_adam_golden_and_ragged(hip, oracle, ops_golden):
    A, g = hip, ops_golden
    p = dev(g["adam_p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(3):
        gr = dev(g["adam_grads"][step])
        A.check(A.lib().pws_adam_step(A.ptr(p), A.ptr(gr), A.ptr(m), A.ptr(v), p.numel(), 1e-2, 0.5, 0.999, 1e-8, step + 1,
                                      A.current_stream()), "adam")
        np.testing.assert_allclose(p.cpu().numpy(), g["adam_p%d" % (step + 1)], rtol=0, atol=2e-6)
    # large ragged buffer vs the oracle
    rs = np.random.RandomState(0)
    n = 1_000_003
    hp, hg = rs.standard_normal(n).astype(np.float32), rs.standard_normal(n).astype(np.float32)
    hm, hv = np.zeros(n, np.float32), np.zeros(n, np.float32)
    dp, dg, dm, dv = dev(hp), dev(hg), dev(hm), dev(hv)
    for step in (1, 2):
        oracle.adam_step(hp, hg, hm, hv, 1e-3, 0.5, 0.999, 1e-8, step)
        A.check(A.lib().pws_adam_step(A.ptr(dp), A.ptr(dg), A.ptr(dm), A.ptr(dv), n, 1e-3, 0.5, 0.999, 1e-8, step,
                                      A.current_stream()), "adam")
    np.testing.assert_allclose(dp.cpu().numpy(), hp, rtol=0, atol=2e-6)


def test_heads_vs_oracle(hip, oracle):
    """theta head (flatten+linear, LeakyReLU on theta) and field head (out conv, tanh, tanh, + affine)."""
    A = hip
    L, st = A.lib(), A.current_stream()
    rs = np.random.RandomState(9)
    n, c, hidden = 3, 64, 128
    x = rs.standard_normal((n, c, 2, 2)).astype(np.float32)
    wf = (rs.standard_normal((hidden, c, 2, 2)) / 16).astype(np.float32)
    bf = rs.standard_normal(hidden).astype(np.float32)
    wl = (rs.standard_normal((6, hidden, 1, 1)) / 11).astype(np.float32)
    bl = rs.standard_normal(6).astype(np.float32)
    f = oracle.conv2d(x, wf, bf, 1, 0, oracle.ACT_LRELU)
    ref_theta = oracle.conv2d(f, wl, bl, 1, 0, oracle.ACT_LRELU).reshape(n, 6)
    pf = torch.empty(L.pws_packed_weight_floats(A.CONV_K2S1P0, c, hidden), device="cuda")
    pl = torch.empty(L.pws_packed_weight_floats(A.CONV_K1, hidden, 6), device="cuda")
    d_wf, d_wl = dev(wf), dev(wl)
    A.check(L.pws_pack_conv_weight(A.ptr(d_wf), A.ptr(pf), A.CONV_K2S1P0, c, hidden, st), "pack")
    A.check(L.pws_pack_conv_weight(A.ptr(d_wl), A.ptr(pl), A.CONV_K1, hidden, 6, st), "pack")
    d_x, d_bf, d_bl = dev(nhwc(x)), dev(bf), dev(bl)
    theta = torch.empty((n, 6), device="cuda")
    tws = torch.empty(L.pws_theta_head_ws_floats(n, c, hidden), device="cuda")
    A.check(L.pws_theta_head_fwd(A.ptr(d_x), n, c, hidden, A.ptr(pf), A.ptr(d_bf), A.ptr(pl), A.ptr(d_bl), A.ptr(tws), A.ptr(theta),
                                 st), "theta")
    np.testing.assert_allclose(theta.cpu().numpy(), ref_theta, rtol=0, atol=2e-5)

    h, w, c = 37, 45, 64
    x = rs.standard_normal((n, c, h, w)).astype(np.float32)
    wo = (rs.standard_normal((2, c, 3, 3)) / 12).astype(np.float32)
    bo = rs.standard_normal(2).astype(np.float32)
    r = np.tanh(oracle.conv2d(x, wo, bo, 1, 1, oracle.ACT_TANH))
    ref_res = nhwc(r)
    ref_grid = ref_res + oracle.affine_grid(ref_theta, h, w)
    po = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1_OUT, c, 2), device="cuda")
    d_wo, d_bo, d_x = dev(wo), dev(bo), dev(nhwc(x))
    A.check(L.pws_pack_conv_weight(A.ptr(d_wo), A.ptr(po), A.CONV_K3S1_OUT, c, 2, st), "pack")
    res, grid = torch.empty((n, h, w, 2), device="cuda"), torch.empty((n, h, w, 2), device="cuda")
    A.check(L.pws_field_head_fwd(A.ptr(d_x), c, n, h, w, c, A.ptr(po), A.ptr(d_bo), A.ptr(theta), 0, A.ptr(res), A.ptr(grid),
                                 st), "field")
    np.testing.assert_allclose(res.cpu().numpy(), ref_res, rtol=0, atol=2e-5)
    np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=0, atol=5e-5)


def test_adam_multi_tensor_matches_single(hip):
    """pws_adam_step_multi (all tensors in one launch per 48) == pws_adam_step per tensor, bit for bit, over ragged sizes
    (aligned float4 bodies, scalar tails, > 48 tensors so that the table is split)."""
    A = hip
    L, st = A.lib(), A.current_stream()
    rs = np.random.RandomState(3)
    sizes = [1, 2, 6, 64, 4097, 5000, 100003, 256] + [rs.randint(1, 3000) for _ in range(60)]
    ps = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda() for s in sizes]
    gs = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda() for s in sizes]
    ms = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda() * 0.1 for s in sizes]
    vs = [torch.from_numpy(rs.uniform(0, 1, s).astype(np.float32)).cuda() for s in sizes]
    ref = [[t.clone() for t in grp] for grp in (ps, ms, vs)]
    for p, g, m, v in zip(ref[0], gs, ref[1], ref[2]):
        A.check(L.pws_adam_step(A.ptr(p), A.ptr(g), A.ptr(m), A.ptr(v), p.numel(), 1e-3, 0.5, 0.999, 1e-8, 3, st), "adam")
    n = len(sizes)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    counts = (ctypes.c_size_t * n)(*sizes)
    A.check(L.pws_adam_step_multi(arr(ps), arr(gs), arr(ms), arr(vs), counts, n, 1e-3, 0.5, 0.999, 1e-8, 3, st), "adam multi")
    for got, want in zip(ps + ms + vs, ref[0] + ref[1] + ref[2]):
        assert torch.equal(got, want)

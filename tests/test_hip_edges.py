"""GPU edge cases of the C ABI: empty batches, ragged / odd batch sizes, non-finite fields, error codes."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from pwstablenet_amd import synth  # noqa: E402


def test_empty_batches_are_noops(hip):
    A = hip
    L, st = A.lib(), A.current_stream()
    assert L.pws_grid_sample_fwd(None, None, None, 0, 3, 8, 8, 8, 8, 0, st) == 0
    assert L.pws_grid_sample_bwd(None, None, None, None, None, 0, 3, 8, 8, 8, 8, 0, st) == 0
    assert L.pws_upsample_grid_sample_fwd(None, None, None, 0, 3, 8, 8, 4, 4, 0, st) == 0
    assert L.pws_affine_grid(None, None, 0, 4, 4, 0, st) == 0
    assert L.pws_adam_step(None, None, None, None, 0, 1e-3, 0.5, 0.999, 1e-8, 1, st) == 0
    assert L.pws_netg_forward(None, None, 0, 31, 64, 0, 0, None, 0, None, None, None, st) == 0
    args = A.PwsConvArgs()
    args.kind, args.n, args.h, args.w, args.nsrc, args.cout, args.out_ld = A.CONV_K3S1, 0, 4, 4, 1, 16, 16
    dummy = torch.zeros(16, device="cuda")
    args.out, args.w_packed = dummy.data_ptr(), dummy.data_ptr()
    assert L.pws_conv2d_fwd(ctypes.byref(args), st) == 0


def test_error_codes_and_messages(hip):
    A = hip
    L, st = A.lib(), A.current_stream()
    assert L.pws_adam_step(None, None, None, None, 8, 1e-3, 0.5, 0.999, 1e-8, 0, st) == -22 and b"step" in L.pws_last_error()
    assert L.pws_grid_sample_fwd(None, None, None, 1, 0, 8, 8, 8, 8, 0, st) == -22
    assert L.pws_set_option(12345, 1) == -22
    # workspace too small -> PWS_ENOMEM, nothing launched
    packed = torch.zeros(L.pws_netg_packed_floats(31, 16), device="cuda")
    x = torch.zeros((1, 31, 256, 256), device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, 256, 256, 2), device="cuda")
    p = (ws.data_ptr() + 255) // 256 * 256
    rc = L.pws_netg_forward(A.ptr(packed), A.ptr(x), 1, 31, 16, 0, 0, ctypes.c_void_p(p), 1 << 19, A.ptr(out), None, None, st)
    assert rc == -12 and b"workspace" in L.pws_last_error()
    assert L.pws_netg_forward(A.ptr(packed), A.ptr(x), 1, 31, 24, 0, 0, ctypes.c_void_p(p), 1 << 19, A.ptr(out), None, None, st) == -22


def test_non_finite_and_far_fields(hip):
    """Coordinates far outside, +-inf and NaN: out-of-range taps contribute zero (NaN propagates as in ATen)."""
    from pwstablenet_amd import functional as PF
    img = torch.rand((1, 3, 16, 16), device="cuda") + 1.0
    grid = torch.zeros((1, 16, 16, 2), device="cuda")
    grid[0, 0, 0] = torch.tensor([1e30, 0.0])
    grid[0, 0, 1] = torch.tensor([-1e30, 5.0])
    grid[0, 0, 2] = torch.tensor([float("inf"), 0.0])
    grid[0, 0, 3] = torch.tensor([0.0, float("-inf")])
    grid[0, 0, 4] = torch.tensor([2.0 + 3.0 / 16, 0.0])
    out = PF.grid_sample(img, grid)
    assert torch.equal(out[0, :, 0, :5], torch.zeros((3, 5), device="cuda"))
    assert torch.isfinite(out).all()
    # ATen's CPU kernel multiplies a NaN weight (inf - inf) into the masked-out tap and returns NaN at the +-inf pixels;
    # ATen's GPU kernel skips out-of-range taps and returns 0 there, as this kernel does.  Compare the finite pixels.
    ref = torch.nn.functional.grid_sample(img.cpu(), grid.cpu(), align_corners=False)
    fin = torch.isfinite(ref)
    assert (out.cpu() - ref)[fin].abs().max().item() < 1e-5 and int((~fin).sum()) == 6
    grid[0, 1, 0] = float("nan")
    assert torch.isnan(PF.grid_sample(img, grid)[0, :, 1, 0]).all()


@pytest.mark.parametrize("n", [1, 3, 9])
def test_netg_odd_batches_match_oracle(hip, oracle, n):
    from pwstablenet_amd.lib.networks_cascading import define_G
    ngf = 16
    weights = synth.make_weights("W1", seed=123, ngf=ngf)
    net = define_G(31, 2, ngf, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in weights})
    net = net.cuda()
    xw = synth.make_window(n, 31, 256, seed=40 + n)
    with torch.no_grad():
        got = net(torch.from_numpy(xw).cuda(), False).cpu().numpy()
    ref = oracle.netg_forward([v for _, v in weights], xw, is_training=False, ngf=ngf)["grids"][0]
    assert np.abs(got - ref).max() < 5e-4


@pytest.mark.parametrize("shape,n", [((64, 64), 3), ((256, 256), 2), ((37, 52), 5)])
def test_grid_sample_row_window_shuffle_variant_is_bit_identical(hip, shape, n):
    """north_star's "wavefront shuffles for the bilinear gather": the row-window + wave-shuffle variant of grid_sample forward
    (PWS_OPT_EXPERIMENT 5; measured in DESIGN.md, not the product default) must give the product kernel's values bit for bit -- on a
    translation (every lane takes the window), on a rotated / scaled field (mixed waves), on a field that leaves the frame, and on a
    group count that is not a multiple of the workgroup."""
    import torch.nn.functional as F
    from pwstablenet_amd import functional as PF
    h, w = shape
    rs = np.random.RandomState(h + n)
    img = torch.from_numpy(rs.uniform(0, 255, (n, 3, h, w)).astype(np.float32)).cuda()
    eye = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])
    thetas = {"translation": eye + torch.tensor([[0, 0, 0.031], [0, 0, -0.017]]),
              "rotation+scale": torch.tensor([[1.04, 0.06, 0.01], [-0.05, 0.97, 0.02]]),
              "leaves the frame": torch.tensor([[1.3, 0.0, 0.4], [0.0, 1.2, -0.5]])}
    for name, th in thetas.items():
        grid = F.affine_grid(th.unsqueeze(0).repeat(n, 1, 1), (n, 3, h, w), align_corners=False).cuda()
        want = PF.grid_sample(img, grid)
        hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 5)
        try:
            got = PF.grid_sample(img, grid)
        finally:
            hip.lib().pws_set_option(hip.OPT_EXPERIMENT, 0)
        assert torch.equal(got, want), name

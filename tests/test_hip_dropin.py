"""The unchanged-driver reach: with `dropin/` on the path, the torch entry points the reference's driver calls
directly -- `torch.nn.functional.grid_sample` / `affine_grid`, `torch.nn.UpsamplingBilinear2d` (main_new.py:106-118,195-197,
708,716) -- run the HIP kernels for device tensors and pass CPU tensors through to torch.  The loop lives in
tests/dropin_driver_loop.py (a fresh child process, as a user would start the driver); here its results are compared with the
torch-CPU restatement of the same calls, and the kernels' own launch records (pws_prof_*) prove which code ran.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from pwstablenet_amd import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_driver_shaped_loop_through_dropin_reaches_the_hip_kernels(hip):
    import torch.nn.functional as F
    from oracle import torch_ref
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_driver_loop.py")], capture_output=True, text=True,
                       env=env, cwd="/tmp", timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DROPIN_JSON ")][-1]
    got = json.loads(line[len("DROPIN_JSON "):])

    # which code ran: 3+1+3 warps + 3 temporal warps forward, the same ten backward; one affine_grid per stage; video loop: one each
    assert got["installed"]
    assert got["train_kernels"] == {"grid_sample_fwd_kernel": 10, "grid_sample_bwd_kernel": 10, "affine_grid_kernel": 3}
    assert got["video_kernels"] == {"grid_sample_fwd_kernel": 1, "upsample_bilinear_ac_kernel": 1}
    assert got["video_is_module"]
    assert got["cpu_devices"] == ["cpu", "cpu", "cpu"] and got["cpu_passed"] == [1, 1, 1]
    assert got["routed"] == {"grid_sample": 11, "affine_grid": 3, "upsample": 1}

    # the same calls on the CPU with torch's own functions (the arithmetic the reference executes)
    ngf, n, period = 16, 2, 30
    torch.set_num_threads(8)   # (not os.cpu_count(): the box reports every core of the host, the cgroup grants a few -- oversubscribed oneDNN crawls)
    params = [torch.from_numpy(v).clone().requires_grad_(True) for _, v in synth.make_weights("W1", seed=123, ngf=ngf)]
    rs = np.random.RandomState(7)
    u1 = torch.from_numpy(np.concatenate([synth.make_window(n, 31, 256, seed=5), synth.make_frames(n, 3, 256, 256, seed=6) / 127.5 - 1], 1).astype(np.float32))
    u2 = torch.from_numpy(np.concatenate([synth.make_window(n, 31, 256, seed=8), synth.make_frames(n, 3, 256, 256, seed=9) / 127.5 - 1], 1).astype(np.float32))
    theta_adj = torch.from_numpy((np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (n, 1)) + 0.02 * rs.randn(n, 6)).astype(np.float32))
    gs = lambda a, b: F.grid_sample(a, b, mode="bilinear", padding_mode="zeros", align_corners=False)  # noqa: E731
    grid1, _ = torch_ref.netg_forward(params, u1[:, :period + 1], True)
    fake1 = [gs((u1[:, period + 1:period + 4] + 1) * 127.5, grid1[nl]) / 127.5 - 1 for nl in range(3)]
    fake1_gray = gs((u1[:, period // 2:period // 2 + 1] + 1) * 127.5, grid1[2]) / 127.5 - 1
    grid2, _ = torch_ref.netg_forward(params, u2[:, :period + 1], True)
    fake2 = [gs((u2[:, period + 1:period + 4] + 1) * 127.5, grid2[nl]) / 127.5 - 1 for nl in range(3)]
    loss = 0
    for nl in range(3):
        grid = F.affine_grid(theta_adj.view(-1, 2, 3), fake1[nl].size(), align_corners=False)
        loss = loss + torch.mean(torch.abs(gs(fake2[nl], grid) - fake1[nl]))
    loss = loss + torch.mean(torch.abs(fake1_gray))
    loss.backward()
    assert abs(got["train_loss"] - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    np.testing.assert_allclose(np.array(got["train_fake1_2"]), fake1[2].detach().numpy()[:, :, ::16, ::16].ravel(), rtol=0, atol=1e-3)
    i_up1 = [k for k, _ in synth.make_weights("W1", seed=123, ngf=ngf)].index("up1.mpconv.0.weight")
    ref_g = params[i_up1].grad.numpy().astype(np.float64).ravel()[::97]
    e = np.linalg.norm(np.array(got["train_grad_up1"]) - ref_g) / np.linalg.norm(ref_g)
    print("dropin loop: loss %.6f (cpu %.6f), up1 weight-gradient rel. L2 error %.3g" % (got["train_loss"], float(loss), e))
    assert e < 1e-2

    with torch.no_grad():
        pr = [p.detach() for p in params]
        field = torch_ref.netg_forward(pr, u1[:1, :period + 1], False)
        now = torch.from_numpy(synth.make_frames(1, 3, 720, 1280, seed=11))
        gr = torch.nn.UpsamplingBilinear2d(size=(720, 1280))(field.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        ref_s = gs(now, gr)[0].numpy()
    ferr = np.abs(np.array(got["video_field"]) - field.numpy()[:, ::8, ::8].ravel()).max()
    serr = np.abs(np.array(got["video_samples"]) - ref_s[:, ::24, ::40].ravel()).max()
    print("dropin video loop: field err %.3g, warped 720p frame err %.3g (0..255 scale)" % (ferr, serr))
    assert ferr < 5e-4 and serr < 1e-3 * 127.5


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((2, 2, 256, 256), (720, 1280)), ((1, 3, 5, 7), (11, 9)), ((2, 1, 4, 4), (4, 4)),
                                         ((1, 2, 6, 5), (1, 1)), ((1, 1, 1, 3), (5, 8)), ((1, 2, 9, 8), (3, 4))])
def test_upsample_bilinear_ac_backward_is_the_adjoint(hip, shape, size):
    """pws_upsample_bilinear_ac_bwd against torch-CPU autograd of UpsamplingBilinear2d (main_new.py:708), incl. down-scaling,
    1-pixel outputs and 1-pixel inputs."""
    from pwstablenet_amd import functional as PF
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal(shape[:2] + size).astype(np.float32))
    xc = x.clone().requires_grad_(True)
    yc = torch.nn.functional.interpolate(xc, size=size, mode="bilinear", align_corners=True)
    yc.backward(g)
    xd = x.cuda().requires_grad_(True)
    yd = PF.upsample_bilinear2d(xd, size)
    yd.backward(g.cuda())
    assert np.abs(yd.detach().cpu().numpy() - yc.detach().numpy()).max() < 2e-6   # same source-index arithmetic as torch (rounded products)
    err = np.abs(xd.grad.cpu().numpy() - xc.grad.numpy()).max()
    assert err < 1e-5 * max(1.0, float(xc.grad.abs().max())), err


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,ac", [(3, 256, 256, False), (2, 5, 7, True), (1, 1, 1, False), (2, 33, 65, False)])
def test_affine_grid_backward_vs_torch(hip, n, h, w, ac):
    """pws_affine_grid_bwd against torch-CPU autograd of F.affine_grid (lib/networks_cascading.py:164; main_new.py:195)."""
    import torch.nn.functional as F
    from pwstablenet_amd import functional as PF
    rs = np.random.RandomState(4)
    th = torch.from_numpy(rs.standard_normal((n, 2, 3)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((n, h, w, 2)).astype(np.float32))
    tc = th.clone().requires_grad_(True)
    F.affine_grid(tc, (n, 3, h, w), align_corners=ac).backward(g)
    td = th.cuda().requires_grad_(True)
    out = PF.affine_grid(td, (n, 3, h, w), align_corners=ac)
    out.backward(g.cuda())
    assert np.abs(out.detach().cpu().numpy() - F.affine_grid(th, (n, 3, h, w), align_corners=ac).numpy()).max() < 2e-6
    scale = max(1.0, float(tc.grad.abs().max()))
    assert np.abs(td.grad.cpu().numpy() - tc.grad.numpy()).max() < 1e-4 * scale


def test_routing_passes_cpu_tensors_through_and_uninstalls():
    """No GPU needed: installed routing leaves CPU calls to torch (bit-identical results) and uninstall() restores torch's own
    attributes."""
    import torch.nn.functional as F
    from pwstablenet_amd import routing
    orig = (F.grid_sample, F.affine_grid, torch.nn.UpsamplingBilinear2d)
    img, grid = torch.rand(1, 3, 9, 11), torch.rand(1, 5, 7, 2) * 2 - 1
    want = (F.grid_sample(img, grid, align_corners=False), torch.nn.UpsamplingBilinear2d(size=(6, 8))(img),
            F.affine_grid(torch.eye(2, 3).unsqueeze(0), (1, 3, 4, 5), align_corners=False))
    routing.install()
    try:
        routing.install()   # idempotent
        assert F.grid_sample is not orig[0] and torch.nn.UpsamplingBilinear2d is not orig[2]
        m = torch.nn.UpsamplingBilinear2d(size=(6, 8))
        assert isinstance(m, orig[2])   # still an instance of torch's class: isinstance checks elsewhere keep working
        got = (F.grid_sample(img, grid, align_corners=False), m(img),
               F.affine_grid(torch.eye(2, 3).unsqueeze(0), (1, 3, 4, 5), align_corners=False))
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        # modes the kernels do not implement stay with torch as well
        assert torch.equal(F.grid_sample(img, grid, mode="nearest", align_corners=False),
                           orig[0](img, grid, mode="nearest", align_corners=False))
    finally:
        routing.uninstall()
    assert (F.grid_sample, F.affine_grid, torch.nn.UpsamplingBilinear2d) == orig

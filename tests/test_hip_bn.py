"""Training-mode BatchNorm2d (+ activation) kernels and the use_BN=True training path of the generator against torch's own
BatchNorm on the CPU and against goldens produced by the reference built with --use_BN 1 in train() mode."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pwstablenet_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACTS = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu}


@pytest.mark.parametrize("pixels,c,act", [(2 * 16 * 16, 64, 1), (3 * 9 * 7, 96, 2), (4, 512, 1), (5, 6, 1), (2 * 64 * 64, 2, 0),
                                          (70000, 16, 2)])
def test_bn_train_fwd_bwd_vs_torch(hip, pixels, c, act):
    L, st = hip.lib(), hip.current_stream
    rs = np.random.RandomState(pixels + c)
    z = torch.from_numpy((rs.standard_normal((pixels, c)) * rs.uniform(0.5, 2, c) + rs.uniform(-1, 1, c)).astype(np.float32))
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
    beta = torch.from_numpy((rs.standard_normal(c) * 0.2).astype(np.float32))
    rm0 = torch.from_numpy((rs.standard_normal(c) * 0.1).astype(np.float32))
    rv0 = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
    gy = torch.from_numpy(rs.standard_normal((pixels, c)).astype(np.float32))
    # torch: (pixels, c) rows are the N*H*W positions of an NCHW tensor with that many "pixels"
    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = ACTS[act](F.batch_norm(zr.t().reshape(1, c, pixels), rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5))
    yr.backward(gy.t().reshape(1, c, pixels))
    ws = torch.empty(L.pws_bn_ws_bytes(c), device="cuda", dtype=torch.uint8)
    d_z, d_g, d_b = z.cuda(), gamma.cuda(), beta.cuda()
    y = torch.empty_like(d_z)
    stats = torch.empty(2 * c, device="cuda")
    d_rm, d_rv = rm0.cuda(), rv0.cuda()
    hip.check(L.pws_bn_train_fwd(hip.ptr(d_z), pixels, c, hip.ptr(d_g), hip.ptr(d_b), act, hip.ptr(y), hip.ptr(stats), hip.ptr(d_rm),
                                 hip.ptr(d_rv), 0.1, 1e-5, 1, hip.ptr(ws), ws.numel(), st()), "bn fwd")
    want = yr.detach().reshape(c, pixels).t().numpy()
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=2e-5)
    np.testing.assert_allclose(d_rm.cpu().numpy(), rm.numpy(), atol=1e-6)
    np.testing.assert_allclose(d_rv.cpu().numpy(), rv.numpy(), rtol=1e-5, atol=1e-6)
    dy = gy.cuda().clone()
    dg, db = torch.full((c,), 0.5, device="cuda"), torch.full((c,), -0.25, device="cuda")      # accumulated onto
    hip.check(L.pws_bn_train_bwd(hip.ptr(dy), hip.ptr(y), hip.ptr(d_z), hip.ptr(stats), hip.ptr(d_g), act, pixels, c, hip.ptr(dg),
                                 hip.ptr(db), hip.ptr(ws), ws.numel(), st()), "bn bwd")
    scale = np.abs(zr.grad.numpy()).max()
    np.testing.assert_allclose(dy.cpu().numpy(), zr.grad.numpy(), atol=2e-4 * max(scale, 1e-3))
    np.testing.assert_allclose(dg.cpu().numpy() - 0.5, gr.grad.numpy(), rtol=2e-4, atol=2e-4 * np.abs(gr.grad.numpy()).max())
    np.testing.assert_allclose(db.cpu().numpy() + 0.25, br.grad.numpy(), rtol=2e-4, atol=2e-4 * np.abs(br.grad.numpy()).max())


def test_bn_refuses_one_value_per_channel_and_repeats_running_update(hip):
    L, st = hip.lib(), hip.current_stream
    c = 8
    ws = torch.empty(L.pws_bn_ws_bytes(c), device="cuda", dtype=torch.uint8)
    z = torch.randn((1, c), device="cuda")
    t = torch.ones(2 * c, device="cuda")
    args = (hip.ptr(t), hip.ptr(t), 0, hip.ptr(z.clone()), hip.ptr(t.clone()), None, None, 0.1, 1e-5, 1, hip.ptr(ws), ws.numel(), st())
    assert L.pws_bn_train_fwd(hip.ptr(z), 1, c, *args) == -22 and b"more than 1 value" in L.pws_last_error()
    z = torch.randn((40, c), device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y, stats = torch.empty_like(z), torch.empty(2 * c, device="cuda")
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    hip.check(L.pws_bn_train_fwd(hip.ptr(z), 40, c, hip.ptr(g), hip.ptr(b), 0, hip.ptr(y), hip.ptr(stats), hip.ptr(rm), hip.ptr(rv), 0.1,
                                 1e-5, 2, hip.ptr(ws), ws.numel(), st()), "bn fwd")
    m, v = z.mean(0), z.var(0, unbiased=True)
    np.testing.assert_allclose(rm.cpu().numpy(), (0.19 * m).cpu().numpy(), atol=1e-6)            # two updates: 1 - 0.9^2
    np.testing.assert_allclose(rv.cpu().numpy(), (0.81 + 0.19 * v).cpu().numpy(), rtol=1e-5)

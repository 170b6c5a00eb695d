"""GPU parity of the whole generator through the drop-in boundary (define_G / netG(x) / netG(x, False)),
against the golden vectors generated from the reference and against the CPU oracle run on the box.

Tolerance: warp-field max-abs error in normalised coordinates.  fp32 everywhere; the HIP path reorders the
K-sums of ~75 chained convolutions, measured ~1e-5 (W1) / ~1e-4 (W2, saturating activations); bound 5e-4 field,
and 1e-3 on warped frames scaled to [-1,1] (north_star).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from pwstablenet_amd import synth  # noqa: E402

FIELD_TOL = 5e-4
WARP_TOL_255 = 1e-3 * 127.5


def make_net(kind, ngf):
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, ngf, "normal", 0.02)
    sd = {"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(kind, seed=123, ngf=ngf)}
    net.load_state_dict(sd, strict=True)
    return net.cuda()


@pytest.mark.parametrize("tag,kind,ngf,n", [("W1_g16", "W1", 16, 2), ("W2_g16", "W2", 16, 1),
                                            ("W1_g64", "W1", 64, 2), ("W2_g64", "W2", 64, 2)])
def test_netg_vs_reference_goldens(hip, netg_golden, tag, kind, ngf, n):
    from pwstablenet_amd import functional as PF
    g = netg_golden
    net = make_net(kind, ngf)
    x = torch.from_numpy(synth.make_window(n, 31, 256, seed=123)).cuda()
    frames = torch.from_numpy(synth.make_frames(n, 3, 256, 256, seed=321)).cuda()
    with torch.no_grad():
        grids, resid = net(x)
        g_inf = net(x, False)
    assert isinstance(grids, list) and len(grids) == 3 and len(resid) == 3
    assert tuple(g_inf.shape) == (n, 256, 256, 2)
    assert torch.equal(g_inf, grids[2]), "netG(x, False) must equal netG(x)[0][2] (reference :237)"
    for k in range(3):
        gk = grids[k].cpu().numpy()
        err = np.abs(gk[:, ::4, ::4] - g["%s_grid%d_sub" % (tag, k)]).max()
        print("%s stage %d warp-field max-abs err vs reference: %.3g" % (tag, k, err))
        assert err < FIELD_TOL
        np.testing.assert_allclose(np.abs(resid[k].cpu().numpy().astype(np.float64)).sum(),
                                   g["%s_resid%d_csum" % (tag, k)][1], rtol=1e-4)
        with torch.no_grad():
            warped = PF.grid_sample(frames, grids[k])
        werr = np.abs(warped.cpu().numpy()[:, :, ::4, ::4] - g["%s_warp%d_sub" % (tag, k)]).max()
        print("%s stage %d warped-frame max-abs err (0..255 scale): %.3g" % (tag, k, werr))
        assert werr < WARP_TOL_255
    if ngf <= 16:
        assert np.abs(g_inf.cpu().numpy() - g[tag + "_grid2_full"]).max() < FIELD_TOL
    np.testing.assert_allclose(net.module.last_thetas[2].cpu().numpy().shape, (n, 6))


def test_netg_vs_oracle_on_box(hip, oracle):
    """Same seeded inputs through the HIP path and the C oracle (ngf=16 keeps the oracle at ~0.3 s)."""
    kind, ngf, n = "W2", 16, 3
    net = make_net(kind, ngf)
    xw = synth.noise_window(n, 31, 256, seed=7)
    ref = oracle.netg_forward([v for _, v in synth.make_weights(kind, seed=123, ngf=ngf)], xw, True, ngf=ngf)
    with torch.no_grad():
        grids, resid = net(torch.from_numpy(xw).cuda())
    for k in range(3):
        assert np.abs(grids[k].cpu().numpy() - ref["grids"][k]).max() < FIELD_TOL
        assert np.abs(resid[k].cpu().numpy() - ref["resid"][k]).max() < FIELD_TOL
    np.testing.assert_allclose(net.module.last_thetas.cpu().numpy(), ref["thetas"], rtol=0, atol=1e-4)


def test_netg_batch_invariance_and_repack(hip):
    """Frames are independent units (no cross-sample op): a batch of 5 equals five batches of 1 up to the fp32
    summation order (tile shape and K split are chosen per batch size), and repeated calls are bit-identical
    (no atomics on the forward path).  An in-place weight update is picked up (re-pack on version change)."""
    net = make_net("W1", 16)
    x = torch.from_numpy(synth.make_window(5, 31, 256, seed=11)).cuda()
    with torch.no_grad():
        full = net(x, False)
        assert torch.equal(net(x, False), full), "forward must be deterministic run to run"
        for i in range(5):
            assert (net(x[i:i + 1], False)[0] - full[i]).abs().max().item() < 2e-5
        before = full.clone()
        net.module.out.mpconv[0].bias.add_(0.25)
        after = net(x, False)
    assert (after - before).abs().max().item() > 1e-3


def test_inference_with_grad_mode_on_and_errors(hip):
    """The reference's video loop calls netG(images, False) without no_grad (main_new.py:697)."""
    net = make_net("W1", 16)
    x = torch.from_numpy(synth.make_window(1, 31, 256, seed=1)).cuda()
    out = net(x, False)
    assert tuple(out.shape) == (1, 256, 256, 2)
    with pytest.raises(RuntimeError):
        net(x.cpu(), False)
    with pytest.raises(RuntimeError):
        net(x[:, :, :128, :128], False)


def test_use_bn_eval_inference_vs_reference_golden(hip):
    """use_BN=True in eval(): BatchNorm running statistics folded into the convs; stage-3 field against the reference built
    with --use_BN 1 (tests/golden/make_golden_bn.py, N=1, ngf=16).  Tolerance 2e-5: fp32 re-association of the fold."""
    import os
    from pwstablenet_amd.lib.networks_cascading import SingleDeviceParallel, UnetGenerator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netg_bn.npz"))
    net = SingleDeviceParallel(UnetGenerator(31, 2, 16, use_BN=True))
    sd = {"module." + k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_weights("W1", seed=123, ngf=16)}
    sd.update({"module." + k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_bn_state(seed=321, ngf=16)})
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    x = torch.from_numpy(synth.make_window(1, 31, 256, seed=123)).cuda()
    with torch.no_grad():
        field = net(x, False).cpu().numpy()
    err = np.abs(field - g["field"]).max()
    assert err < 2e-5, err
    # changing a running statistic re-folds (the packed-weight cache follows the BatchNorm buffers)
    with torch.no_grad():
        net.module.up_bottom1.mpconv[1].running_var.mul_(4.0)
        field2 = net(x, False).cpu().numpy()
    assert np.abs(field2 - field).max() > 1e-4
    with pytest.raises(ValueError):      # train() mode runs BatchNorm with batch statistics (tests/test_hip_bn.py); batch 1 is refused as by torch
        net.train()(x)

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


@pytest.mark.parametrize("ngf", [32, 16])
def test_pack_for_one_math_mode(hip, ngf):
    """pws_netg_pack_weights_for (round 4): math = -1 is pws_netg_pack_weights bit for bit; PWS_MATH_BF16 / PWS_MATH_FP32 write a subset
    of the same values and leave the rest of the buffer alone (checked on a NaN-filled buffer: every written float equals the full
    pack's) -- and a forward of that mode on such a buffer equals the forward on the full pack bit for bit, while the generator
    re-packs when set_math changes the mode.  ngf 16: the shallow layers (16 / 48 input channels) have no bf16 weights and run fp32
    Winograd in bf16 mode too -- their Winograd copies must be kept."""
    import ctypes
    A = hip
    L, st = A.lib(), A.current_stream()
    net = make_net("W2", ngf)
    g = net.module
    params = g._effective_params()
    ptrs = (ctypes.c_void_p * len(params))(*[p_.data_ptr() for p_ in params])
    nfl = L.pws_netg_packed_floats(31, ngf)
    full = torch.full((nfl,), float("nan"), device="cuda")
    A.check(L.pws_netg_pack_weights(ptrs, A.ptr(full), 31, ngf, st), "pack")
    written = {}
    for math in (-1, A.MATH_BF16, A.MATH_FP32):
        buf = torch.full((nfl,), float("nan"), device="cuda")
        A.check(L.pws_netg_pack_weights_for(ptrs, A.ptr(buf), 31, ngf, math, st), "pack_for")
        # (bf16 pairs read as fp32 may BE NaN patterns: compare bit patterns against the fill)
        w = buf.view(torch.int32) != torch.full((1,), float("nan"), device="cuda").view(torch.int32)
        assert torch.equal(buf.view(torch.int32)[w], full.view(torch.int32)[w]), math
        written[math] = int(w.sum())
    full_w = int((full.view(torch.int32) != torch.full((1,), float("nan"), device="cuda").view(torch.int32)).sum())
    assert written[-1] == full_w
    assert written[A.MATH_FP32] < full_w    # the bf16 copies are left out
    assert written[A.MATH_BF16] < full_w    # the Winograd copies of the layers that run on bf16 weights are left out
    x = torch.from_numpy(synth.make_window(2, 31, 256, seed=3)).cuda()
    outs = {}
    with torch.no_grad():
        for math in ("fp32", "bf16", "fp32"):
            g.set_math(math)
            out = net(x, False).clone()
            assert g._packed_key[0] == math
            # the same forward on a full pack of the same weights
            keep_key = g._packed_key
            A.check(L.pws_netg_pack_weights(ptrs, A.ptr(g._packed), 31, ngf, st), "pack")
            assert torch.equal(net(x, False), out), math
            assert g._packed_key == keep_key
            outs.setdefault(math, out)
            assert torch.equal(outs[math], out)   # fp32 again after bf16: re-packed, same result


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


@pytest.mark.parametrize("math", ["fp32", "bf16"])
def test_lockstep_schedule_equals_the_per_stage_schedule(hip, math):
    """Round 4: stages 2 and 3 run in lockstep -- the 23 layers they share go out as ONE launch of batch 2n on sample-group views of
    shared buffers (csrc/netg.cpp forward_lockstep; reference lib/networks_cascading.py:178-219 runs them one after the other).
    PWS_OPT_EXPERIMENT 15 keeps the per-stage schedule: same layers on the same inputs, so the six training fields and the inference
    field agree to rounding (a launch of twice the batch may split K differently: fp32 summation order), and so does one backward
    (the tape holds one op per stage in either schedule)."""
    L = hip.lib()
    x = torch.from_numpy(synth.make_window(3, 31, 256, seed=77)).cuda()   # odd batch: group offsets that are no multiple of anything
    up = [torch.from_numpy(np.random.RandomState(5 + k).standard_normal((3, 256, 256, 2)).astype(np.float32)).cuda() for k in range(3)]
    out = {}
    try:
        for e in (15, 0):
            L.pws_set_option(100, e)
            net = make_net("W2", 32)          # a fresh generator per schedule: arenas are sized by the schedule in force
            net.module.set_math(math)
            net.module.deterministic = True   # (no atomics noise in the gradients: what is left is the K-split / summation order)
            with torch.no_grad():
                inf = net(x, False).clone()
            grids, resid = net(x)
            assert torch.equal(inf, grids[2].detach())
            loss = sum((g * u).sum() for g, u in zip(grids, up)) + (resid[2] ** 2).sum()
            loss.backward()
            out[e] = ([g.detach().clone() for g in grids] + [r.detach().clone() for r in resid], [p.grad.clone() for p in net.parameters()])
    finally:
        L.pws_set_option(100, 0)
    ftol = 2e-5 if math == "fp32" else 3e-2
    for a, b in zip(out[15][0], out[0][0]):
        assert float((a - b).abs().max()) < ftol, float((a - b).abs().max())
    if math == "bf16":
        # bf16 storage: a 1e-2 difference in the forward flips LeakyReLU masks of the theta heads' six values per sample (slope 1 <-> 0.2), so
        # single tensors of the heads move by tens of per cent between equally valid evaluations; judged over ALL 48.5 M... (12 M here) entries
        fa = torch.cat([t.double().reshape(-1) for t in out[15][1]])
        fb = torch.cat([t.double().reshape(-1) for t in out[0][1]])
        assert float((fa @ fb) / (fa.norm() * fb.norm())) > 0.999
        return
    for i, (a, b) in enumerate(zip(out[15][1], out[0][1])):
        scale = float(a.abs().max()) + 1e-20
        # (measured 2.0e-4 .. 5.2e-4 of a tensor's maximum on deep layers: the loss weights the fields with white noise and W2 saturates
        #  the tanh heads, so these gradients are strongly cancelling sums, and a 1e-6 difference in the forward moves LeakyReLU / ReLU
        #  masks of activations near zero; tests/test_hip_ddp.py sees 1.5e-4 between one launch of 32 items and two of 16)
        assert float((a - b).abs().max()) / scale < 2e-3, (i, float((a - b).abs().max()) / scale)
        cos = float((a.double().reshape(-1) @ b.double().reshape(-1)) / (a.double().norm() * b.double().norm() + 1e-300))
        assert cos > 0.999999, (i, cos)


@pytest.mark.parametrize("math", ["fp32", "bf16"])
def test_prune_dead_gives_the_same_field_bit_for_bit(hip, math):
    """PWS_NETG_PRUNE_DEAD (UnetGenerator.prune_dead): the inference forward without stage 1's up2 -- its output x122 (reference
    lib/networks_cascading.py:171) is read only under `if is_training` (:173, :196) -- returns the SAME field, eagerly and from a replayed graph,
    with one launch less; the training forward ignores the flag (it needs x122)."""
    from pwstablenet_amd import synth
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W2", seed=123, ngf=64)})
    net = net.cuda()
    net.module.set_math(math)
    x = torch.from_numpy(synth.make_window(4, 31, 256, seed=17)).cuda()
    L = hip.lib()

    def run(prune, graph):
        net.module.prune_dead = prune
        net.module.enable_graph(graph)
        with torch.no_grad():
            if graph:
                net(x, False)
            L.pws_prof_enable(0 if graph else 1)
            f = net(x, False).clone()
            L.pws_prof_enable(0)
        return f, (0 if graph else len(hip.prof_collect()))
    try:
        f0, n0 = run(False, False)
        f1, n1 = run(True, False)
        g1, _ = run(True, True)
        assert torch.equal(f0, f1) and torch.equal(f0, g1)
        assert n1 == n0 - 1, (n0, n1)   # exactly the one launch of up2 (profiler scopes: one per conv launch)
        net.module.enable_graph(False)
        with torch.no_grad():
            a = net(x)
            net.module.prune_dead = False
            b = net(x)
        assert all(torch.equal(p, q) for p, q in zip(a[0] + a[1], b[0] + b[1]))
    finally:
        net.module.prune_dead = False
        net.module.enable_graph(False)
        net.module.set_math("fp32")

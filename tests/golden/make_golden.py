#!/usr/bin/env python3
"""Generates the golden vectors in this directory by IMPORTING THE REFERENCE (read-only) on CPU.

Run in the build container only:  python tests/golden/make_golden.py
(the reference checkout lives at /root/reference and never travels to the GPU box; nothing under
tests/, bench.py or smoke() reads it at run time -- they read the .npz/.json files written here.)

The reference ships no tests or fixtures, so these vectors are what pins the
oracle: outputs of the reference's own ``lib/networks_cascading.py`` (UnetGenerator via define_G) and
of the torch built-ins its drivers call (F.grid_sample, F.affine_grid, UpsamplingBilinear2d, Adam) on
deterministic inputs / weights that ``pwstablenet_amd.synth`` regenerates bit-identically everywhere.
Only inputs' seeds and the reference's *outputs* are stored; no reference source is copied.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PWS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pwstablenet_amd import spec, synth  # noqa: E402


def import_reference():
    argv, sys.argv = sys.argv, ["x"]  # lib/cfg.py parses argv at import time (lib/cfg.py:43)
    sys.path.insert(0, REF)
    try:
        import lib.cfg as rcfg  # noqa: F401
        import lib.networks_cascading as rnet
    finally:
        sys.argv = argv
        sys.path.remove(REF)
    return rcfg, rnet


def csum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), np.abs(a).max()], dtype=np.float64)


def sample_idx(n, k=32, seed=7):
    return np.random.RandomState(seed).randint(0, n, k)


def net_goldens(rnet, out, tag, kind, ngf, n):
    torch.manual_seed(0)
    net = rnet.define_G(31, 2, ngf, "normal", 0.02)
    weights = synth.make_weights(kind, seed=123, ngf=ngf)
    sd = {"module." + k: torch.from_numpy(v) for k, v in weights}
    net.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth.make_window(n, 31, 256, seed=123))
    frames = torch.from_numpy(synth.make_frames(n, 3, 256, 256, seed=321))

    # hooks for intermediate activations named as in the reference's forward()
    m = net.module
    acts = {}
    with torch.no_grad():
        grids, resid = net(x)  # is_training=True
        g_inf = net(x, False)
        assert torch.equal(g_inf, grids[2])
        # theta per stage recomputed through the module's own blocks
        x11 = m.transfer(x)
        acts["x11"] = x11
        xs = [x11]
        for i in range(1, 8):
            xs.append(getattr(m, "down%d" % i)(xs[-1]))
        acts["x14"], acts["x18"] = xs[3], xs[7]
        theta1 = m.linear(m.flatten(xs[7])).view(-1, 6)
        warped = [F.grid_sample(frames, g, align_corners=False) for g in grids]
    for k in range(3):
        g = grids[k].numpy()
        r = resid[k].numpy()
        out["%s_grid%d_sub" % (tag, k)] = g[:, ::4, ::4, :].copy()
        out["%s_grid%d_csum" % (tag, k)] = csum(g)
        out["%s_resid%d_csum" % (tag, k)] = csum(r)
        out["%s_warp%d_sub" % (tag, k)] = warped[k].numpy()[:, :, ::4, ::4].copy()
        out["%s_warp%d_csum" % (tag, k)] = csum(warped[k].numpy())
        # affine part = grid - residual ; recover theta by solving on 3 pixels is overkill: store corners
        aff = g - r
        out["%s_affine%d_corners" % (tag, k)] = aff[:, [0, 0, -1, -1], [0, -1, 0, -1], :].copy()
    if ngf <= 16:
        out["%s_grid2_full" % tag] = grids[2].numpy().copy()
    out["%s_theta1" % tag] = theta1.numpy().copy()
    for name in ("x11", "x14", "x18"):
        a = acts[name].numpy()
        out["%s_act_%s_csum" % (tag, name)] = csum(a)
        flat = a.reshape(-1)
        out["%s_act_%s_samples" % (tag, name)] = flat[sample_idx(flat.size)].copy()

    # one training-step golden: L1 loss of the warped frame against a shifted target, summed over the
    # three stages (the L1 pixel term of lib/utils.py:339-343 / main_new.py:184-190), backward, grads.
    net.zero_grad()
    target = torch.roll(frames, shifts=(2, -3), dims=(2, 3))
    grids, resid = net(x)
    loss = sum(F.l1_loss(F.grid_sample(frames, g, align_corners=False) / 127.5 - 1, target / 127.5 - 1)
               for g in grids)
    loss.backward()
    out["%s_loss" % tag] = np.array([loss.item()], dtype=np.float64)
    names = ["transfer.mpconv.0.weight", "down4.mpconv.0.weight", "up3.mpconv.0.weight", "out.mpconv.0.weight",
             "down_bottom1.conv_same.0.weight", "down_bottom5.mpconv.0.weight", "up_bottom6.mpconv.0.weight",
             "up_bottom2.conv_same.0.weight", "flatten.mpconv.0.weight", "linear.mpconv.0.bias",
             "out.mpconv.0.bias", "up_bottom1.mpconv.0.bias"]
    named = dict(m.named_parameters())
    for nm in names:
        gnp = named[nm].grad.numpy()
        out["%s_grad_%s_csum" % (tag, nm)] = csum(gnp)
        flat = gnp.reshape(-1)
        out["%s_grad_%s_samples" % (tag, nm)] = flat[sample_idx(flat.size, 16)].copy()
    return net


def op_goldens(out):
    rs = np.random.RandomState(11)
    # --- grid_sample fwd/bwd known-answer cases (main_new.py:106-118,197,716) ---
    for tag, (n, c, h, w, ho, wo) in {"gs_small": (2, 3, 5, 7, 5, 7), "gs_mid": (2, 3, 33, 65, 17, 40),
                                      "gs_gray": (1, 1, 9, 9, 9, 9)}.items():
        img = rs.standard_normal((n, c, h, w)).astype(np.float32)
        grid = rs.uniform(-1.3, 1.3, (n, ho, wo, 2)).astype(np.float32)
        # exact-integer and half-integer source coordinates and exact borders
        grid[0, 0, 0] = [-1.0, -1.0]
        grid[0, 0, 1] = [1.0, 1.0]
        grid[0, 0, 2] = [(2 * 2 + 1) / w - 1, (2 * 1 + 1) / h - 1]  # pixel centre (2,1)
        grid[0, 0, 3] = [(2 * 2.5 + 1) / w - 1, (2 * 1.5 + 1) / h - 1]  # half-way
        grid[0, 0, 4] = [-1.0 - 1.0 / w, 0.0]  # exactly half a pixel outside
        gout = rs.standard_normal((n, c, ho, wo)).astype(np.float32)
        out[tag + "_img"], out[tag + "_grid"], out[tag + "_gout"] = img, grid, gout
        for ac in (False, True):
            ti = torch.from_numpy(img).requires_grad_(True)
            tg = torch.from_numpy(grid).requires_grad_(True)
            o = F.grid_sample(ti, tg, mode="bilinear", padding_mode="zeros", align_corners=ac)
            o.backward(torch.from_numpy(gout))
            sfx = "_ac%d" % int(ac)
            out[tag + "_out" + sfx] = o.detach().numpy().copy()
            out[tag + "_ginput" + sfx] = ti.grad.numpy().copy()
            out[tag + "_ggrid" + sfx] = tg.grad.numpy().copy()
    # --- affine_grid (lib/networks_cascading.py:164) ---
    theta = rs.standard_normal((3, 2, 3)).astype(np.float32)
    out["ag_theta"] = theta
    for ac in (False, True):
        out["ag_out_ac%d" % int(ac)] = F.affine_grid(torch.from_numpy(theta), torch.Size((3, 3, 12, 20)),
                                                     align_corners=ac).numpy().copy()
    out["ag_out256_ac0_sub"] = F.affine_grid(torch.from_numpy(theta), torch.Size((3, 3, 256, 256)),
                                            align_corners=False).numpy()[:, ::16, ::16].copy()
    # --- UpsamplingBilinear2d(size=(720,1280)) of a 256x256 field (main_new.py:706-710) + warp (:716) ---
    field = (F.affine_grid(torch.from_numpy(theta[:1] * 0.05 + np.array([[[1, 0, 0], [0, 1, 0]]], np.float32)),
                           torch.Size((1, 3, 256, 256)), align_corners=False)
             + 0.05 * torch.from_numpy(rs.standard_normal((1, 256, 256, 2)).astype(np.float32)))
    out["up_field"] = field.numpy().copy()
    up = torch.nn.UpsamplingBilinear2d(size=(720, 1280))(field.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    out["up_out_sub"] = up.numpy()[:, ::9, ::16].copy()
    out["up_out_csum"] = csum(up.numpy())
    frame = torch.from_numpy(synth.make_frames(1, 3, 720, 1280, seed=99))
    warped = F.grid_sample(frame, up.contiguous(), align_corners=False)
    out["up_warp_sub"] = warped.numpy()[:, :, ::9, ::16].copy()
    out["up_warp_csum"] = csum(warped.numpy())
    # --- conv / conv_transpose small cases covering every (k,s,p,act) the generator uses ---
    cases = [("c5s1", "conv", 7, 5, 5, 1, 2, "lrelu", 12, 10), ("c3s2", "conv", 6, 9, 3, 2, 1, "lrelu", 12, 10),
             ("c3s1", "conv", 6, 6, 3, 1, 1, "lrelu", 9, 11), ("c2s1", "conv", 8, 12, 2, 1, 0, "lrelu", 2, 2),
             ("c1s1", "conv", 12, 6, 1, 1, 0, "lrelu", 1, 1), ("c3s1t", "conv", 6, 2, 3, 1, 1, "tanh", 9, 11),
             ("t4s2", "convT", 6, 5, 4, 2, 1, "relu", 5, 6), ("t3s1", "convT", 6, 6, 3, 1, 1, "relu", 7, 5)]
    acts = {"lrelu": lambda t: F.leaky_relu(t, 0.2), "relu": F.relu, "tanh": torch.tanh}
    for tag, kind, ci, co, k, s, p, act, h, w in cases:
        x = rs.standard_normal((2, ci, h, w)).astype(np.float32)
        wshape = (co, ci, k, k) if kind == "conv" else (ci, co, k, k)
        wt = (rs.standard_normal(wshape) * 0.3).astype(np.float32)
        b = rs.standard_normal((co,)).astype(np.float32)
        fn = F.conv2d if kind == "conv" else F.conv_transpose2d
        y = acts[act](fn(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride=s, padding=p))
        out["conv_%s_x" % tag], out["conv_%s_w" % tag], out["conv_%s_b" % tag] = x, wt, b
        out["conv_%s_y" % tag] = y.numpy().copy()
    # --- Adam: 3 steps on 3 small tensors, betas=(0.5,0.999) (main_new.py:63) ---
    ps = [torch.from_numpy(rs.standard_normal(s_).astype(np.float32)).requires_grad_(True)
          for s_ in ((5, 3), (17,), (2, 3, 4))]
    out["adam_p0"] = np.concatenate([p.detach().numpy().reshape(-1) for p in ps])
    opt = torch.optim.Adam(ps, lr=1e-2, betas=(0.5, 0.999))
    gs = []
    for step in range(3):
        g = [rs.standard_normal(tuple(p.shape)).astype(np.float32) for p in ps]
        gs.append(np.concatenate([a.reshape(-1) for a in g]))
        for p, a in zip(ps, g):
            p.grad = torch.from_numpy(a)
        opt.step()
        out["adam_p%d" % (step + 1)] = np.concatenate([p.detach().numpy().reshape(-1) for p in ps])
    out["adam_grads"] = np.stack(gs)


def main():
    torch.set_num_threads(8)
    rcfg, rnet = import_reference()
    meta = {"torch": torch.__version__, "numpy": np.__version__,
            "reference_files": ["lib/cfg.py", "lib/networks_cascading.py"],
            "align_corners_default": False}
    # state-dict keys / shapes of the reference generator
    net = rnet.define_G(31, 2, 64, "normal", 0.02)
    sd = net.state_dict()
    meta["state_dict"] = [[k, list(v.shape)] for k, v in sd.items()]
    mine = [("module." + k, list(s)) for k, s in spec.param_specs(31, 2, 64)]
    assert [list(t) for t in mine] == meta["state_dict"], "spec.param_specs disagrees with the reference"
    meta["num_params"] = int(sum(v.numel() for v in sd.values()))
    # config surface (lib/cfg.py:7-39): flag names, defaults and types
    meta["cfg_defaults"] = {k: [type(v).__name__, v] for k, v in sorted(vars(rcfg.opt).items())}
    meta["cfg_constants"] = {"period": rcfg.period, "index_sample": rcfg.index_sample.tolist(),
                             "index_sample_discriminator": rcfg.index_sample_discriminator.tolist(),
                             "train_files": rcfg.train_files, "val_files": rcfg.val_files,
                             "test_files": rcfg.test_files}
    with open(os.path.join(HERE, "reference_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)

    ops = {}
    op_goldens(ops)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    nets = {}
    net_goldens(rnet, nets, "W1_g64", "W1", 64, 2)
    net_goldens(rnet, nets, "W2_g64", "W2", 64, 2)
    net_goldens(rnet, nets, "W1_g16", "W1", 16, 2)
    net_goldens(rnet, nets, "W2_g16", "W2", 16, 1)
    np.savez_compressed(os.path.join(HERE, "netg.npz"), **nets)
    for k in sorted(nets):
        if k.endswith("csum") and ("grid" in k or "resid" in k):
            print(k, nets[k])
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()

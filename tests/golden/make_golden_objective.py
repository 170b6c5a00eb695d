#!/usr/bin/env python3
"""Golden vectors for the training objective around the hot path, produced by CALLING THE
REFERENCE's own functions on CPU.  Run in the build container only:  python tests/golden/make_golden_objective.py

What is the reference's and what is this script's:
  * reference code executed: ``lib/utils.py`` ``pre_propossing`` (:246), ``loss_calulate`` (:339), ``loss_pixel1`` (:405),
    ``generate_affine_matrix`` (:427) and ``lib/networks_cascading.py`` ``define_G`` / ``UnetGenerator.forward``;
  * ``main_new.py`` cannot be imported (visdom, module-level CUDA check :26-27), so the *composition* of a generator
    step (main_new.py:101-118 warps, :184-212 sums, no GAN, VGG term omitted: needs torchvision weights) is written out
    below with the reference's functions as the parts; the build's restatement of the same composition is
    oracle/objective_ref.py and tests compare it with the numbers stored here;
  * ``lib/utils.py`` has top-level ``import cv2`` / ``from torchvision.models.vgg import vgg16`` (both absent in this image)
    and ``loss_pixel1`` allocates with ``.cuda()``.  None of the four functions above touches cv2 / torchvision, so empty
    placeholder modules satisfy the import statements, and ``Tensor.cuda`` is made the identity for the duration of the
    run (device placement only; every arithmetic instruction executed is the reference's).
Only seeds and reference OUTPUTS are stored (tests/golden/objective.npz); no reference source is copied.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PWS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402

N = 2          # item pairs per step (opt.batchSize)
NGF = 64


def import_reference():
    for name in ("cv2", "torchvision", "torchvision.models", "torchvision.models.vgg"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.models.vgg"].vgg16 = None  # the name the import statement binds; never called
    argv, sys.argv = sys.argv, ["x", "--batchSize", str(N)]  # lib/cfg.py parses argv at import time
    sys.path.insert(0, REF)
    try:
        import lib.cfg as rcfg
        import lib.networks_cascading as rnet
        import lib.utils as rutils
    finally:
        sys.argv = argv
        sys.path.remove(REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    return rcfg, rnet, rutils


def csum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), np.abs(a).max()], dtype=np.float64)


def sample_idx(n, k=32, seed=7):
    return np.random.RandomState(seed).randint(0, n, k)


def smooth_field(n, size, seed, amp):
    """identity field + a smooth perturbation (what a trained generator emits), float32 (n,size,size,2)."""
    rs = np.random.RandomState(seed)
    theta = np.array([[1, 0, 0], [0, 1, 0]], np.float32)[None] + rs.normal(0, 0.02, (n, 2, 3)).astype(np.float32)
    base = F.affine_grid(torch.from_numpy(theta), torch.Size((n, 3, size, size)), align_corners=False).numpy()
    yy, xx = np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size), indexing="ij")
    pert = np.zeros((n, size, size, 2), np.float32)
    for i in range(n):
        for c in range(2):
            for _ in range(3):
                fx, fy, ph = rs.uniform(0.5, 4, 2).tolist() + [rs.uniform(0, 6.28)]
                pert[i, :, :, c] += (amp / 3 * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)).astype(np.float32)
    pert += rs.normal(0, amp * 0.02, pert.shape).astype(np.float32)
    return (base + pert).astype(np.float32), pert.astype(np.float32)


def component_goldens(rcfg, rutils, out):
    opt = rcfg.opt
    size, nf = opt.input_size, opt.number_feature
    images1, features1, _, _, _, _, adjacent = synth.make_train_batch(N, seed=11, size=size, number_feature=nf)
    # pre_propossing (lib/utils.py:246)
    st, un, fs, fu = rutils.pre_propossing(torch.from_numpy(images1), torch.from_numpy(features1).float())
    out["pre_stable_csum"], out["pre_unstable_csum"] = csum(st.numpy()), csum(un.numpy())
    out["pre_unstable_samples"] = un.numpy().reshape(-1)[sample_idx(un.numel())].copy()
    out["pre_fs"], out["pre_fu"] = fs.numpy().copy(), fu.numpy().copy()
    # loss_calulate (lib/utils.py:339) on a smooth field, values + gradients wrt field and warped frame
    grid_np, resid_np = smooth_field(N, size, 21, 0.05)
    grid = torch.from_numpy(grid_np).requires_grad_(True)
    rgb = (un[:, rcfg.period + 1:rcfg.period + 4] + 1) * 127.5
    fake = (F.grid_sample(rgb, grid, align_corners=False) / 127.5 - 1)
    fake.retain_grad()
    mse, delta, feat = rutils.loss_calulate(grid, fs, fu, fake, st, N)
    (mse + feat).backward()
    out["lc_values"] = np.array([mse.item(), delta.item(), feat.item()], np.float64)
    out["lc_ggrid_csum"] = csum(grid.grad.numpy())
    out["lc_ggrid_sub"] = grid.grad.numpy()[:, ::8, ::8].copy()
    out["lc_gfake_csum"] = csum(fake.grad.numpy())
    out["lc_fake_csum"] = csum(fake.detach().numpy())
    out["lc_fake_sub"] = fake.detach().numpy()[:, :, ::8, ::8].copy()
    # loss_pixel1 + generate_affine_matrix (lib/utils.py:405,427; main_new.py:78-80,203)
    A = torch.from_numpy(rutils.generate_affine_matrix(opt.block, opt.block)).float().unsqueeze(0).repeat(N, 1, 1, 1, 1).to(torch.float64)
    eye = torch.tensor([[1, 0, 0], [0, 1, 0]], dtype=torch.float).unsqueeze(0).expand(N, 2, 3)
    grid_eye = F.affine_grid(eye, torch.Size((N, 3, size, size)), align_corners=False)
    out["shape_basis_block"] = rutils.generate_affine_matrix(opt.block, opt.block)[:opt.block, :opt.block, 0, :].copy()
    resid = torch.from_numpy(resid_np).requires_grad_(True)
    lp = rutils.loss_pixel1(resid, grid_eye, A, opt.block, opt.block, size)
    lp.backward()
    out["shape_value"] = np.array([lp.item()], np.float64)
    out["shape_gresid_csum"] = csum(resid.grad.numpy())
    out["shape_gresid_sub"] = resid.grad.numpy()[:, ::8, ::8].copy()
    # a field that IS block-wise bilinear has zero loss: not a golden, a sanity check of the reference's meaning
    return A, grid_eye


def step_golden(rcfg, rnet, rutils, A, grid_eye, out, kind):
    """One generator step of train() (main_new.py:84-214) on a synthetic batch of N item pairs."""
    opt, period = rcfg.opt, rcfg.period
    size = opt.input_size
    torch.manual_seed(0)
    net = rnet.define_G(31, 2, NGF, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(kind, seed=123, ngf=NGF)})
    batch = synth.make_train_batch(N, seed=31, size=size, number_feature=opt.number_feature)
    images1, features1, affine1, images2, features2, affine2, adjacent = [torch.from_numpy(b) for b in batch]
    features1, features2, adjacent = features1.float(), features2.float(), adjacent.float()
    st1, un1, fs1, fu1 = rutils.pre_propossing(images1, features1)
    st2, un2, fs2, fu2 = rutils.pre_propossing(images2, features2)
    fake1, fake2 = [], []
    grid1, res1 = net(un1[:, 0:period + 1])
    for nl in range(opt.num_layer):
        t = F.grid_sample((un1[:, period + 1:period + 4] + 1) * 127.5, grid1[nl], align_corners=False)
        fake1.append(t / 127.5 - 1)
    grid2, res2 = net(un2[:, 0:period + 1])
    for nl in range(opt.num_layer):
        t = F.grid_sample((un2[:, period + 1:period + 4] + 1) * 127.5, grid2[nl], align_corners=False)
        fake2.append(t / 127.5 - 1)
    loss_mse = loss_feature = loss_delta = loss_g2 = 0
    for nl in range(opt.num_layer):
        m1, d1, f1 = rutils.loss_calulate(grid1[nl], fs1, fu1, fake1[nl], st1, opt.batchSize)
        m2, d2, f2 = rutils.loss_calulate(grid2[nl], fs2, fu2, fake2[nl], st2, opt.batchSize)
        loss_mse += m1 + m2
        loss_feature += f1 + f2
        loss_delta += d1 + d2
        adjacent = adjacent.view(-1, 2, 3)
        g = F.affine_grid(adjacent, fake1[nl].size(), align_corners=False)
        o21 = F.grid_sample(fake2[nl], g, align_corners=False)
        loss_g2 += torch.mean(torch.abs(o21 - fake1[nl]))
        if opt.shapeloss:
            loss_pixel = rutils.loss_pixel1(res1[nl], grid_eye, A, opt.block, opt.block, size) * opt.shapeloss_weight + \
                rutils.loss_pixel1(res2[nl], grid_eye, A, opt.block, opt.block, size) * opt.shapeloss_weight
    loss_g1 = loss_feature + loss_mse + loss_pixel
    loss_g = loss_g1 + loss_g2 * opt.lamd
    net.zero_grad()
    loss_g.backward()
    tag = "step_%s" % kind
    out[tag + "_losses"] = np.array([loss_g.item(), loss_mse.item(), loss_feature.item(), loss_delta.item(), loss_g2.item(),
                                     loss_pixel.item()], np.float64)
    out[tag + "_fake1_2_csum"] = csum(fake1[2].detach().numpy())
    out[tag + "_fake2_0_csum"] = csum(fake2[0].detach().numpy())
    named = dict(net.module.named_parameters())
    for nm in ["transfer.mpconv.0.weight", "down4.mpconv.0.weight", "up3.mpconv.0.weight", "out.mpconv.0.weight",
               "down_bottom5.mpconv.0.weight", "up_bottom2.conv_same.0.weight", "flatten.mpconv.0.weight",
               "linear.mpconv.0.bias", "out.mpconv.0.bias"]:
        gnp = named[nm].grad.numpy()
        out["%s_grad_%s_csum" % (tag, nm)] = csum(gnp)
        out["%s_grad_%s_samples" % (tag, nm)] = gnp.reshape(-1)[sample_idx(gnp.size, 16)].copy()
    print(tag, out[tag + "_losses"])


def main():
    torch.set_num_threads(8)
    rcfg, rnet, rutils = import_reference()
    assert rcfg.opt.batchSize == N and rcfg.opt.shapeloss and not rcfg.opt.use_gan
    out = {"cfg": np.array([N, rcfg.opt.input_size, rcfg.opt.number_feature, rcfg.opt.block, rcfg.opt.lamd, rcfg.opt.num_layer,
                            rcfg.opt.shapeloss_weight], np.float64)}
    A, grid_eye = component_goldens(rcfg, rutils, out)
    step_golden(rcfg, rnet, rutils, A, grid_eye, out, "W1")
    np.savez_compressed(os.path.join(HERE, "objective.npz"), **out)
    for k in ("lc_values", "shape_value"):
        print(k, out[k])
    print("wrote objective.npz (%d arrays)" % len(out))


if __name__ == "__main__":
    main()

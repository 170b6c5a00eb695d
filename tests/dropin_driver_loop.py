"""A freshly written driver loop with the CALL SHAPES of the reference's main_new.py, run through dropin/ exactly as a user
would run the unchanged driver (PYTHONPATH = repo root + dropin/).  Test infrastructure: started as a child process by
tests/test_hip_dropin.py, prints one JSON line.

Call shapes mirrored (reference main_new.py): :5 `from lib.networks_cascading import define_G, ...`; :9 `import
torch.nn.functional as functional`; :31,41 define_G(...).cuda(); :101 `grid1, affine1 = netG(x[:, 0:period+1])`; :106
`functional.grid_sample((x[:, period+1:period+4] + 1) * 127.5, grid1[nl])`; :109 the gray plane through the last field; :195-197
`functional.affine_grid(theta.view(-1, 2, 3), fake.size())` + `functional.grid_sample(fake2, grid)`; :214 backward; :697-716
`netG(images, False)`, `.permute(0, 3, 1, 2)`, `torch.nn.UpsamplingBilinear2d(size=(H, W))`, `.permute(0, 2, 3, 1)`,
`functional.grid_sample(now, grid_resize)` -- the video loop runs with gradients enabled (no no_grad in the reference).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]]
sys.path[:0] = [ROOT, os.path.join(ROOT, "dropin")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as functional  # noqa: E402  (bound BEFORE the drop-in is imported, as in the driver)
from lib.networks_cascading import define_G  # noqa: E402  (dropin/lib: installs the routing)
from lib.cfg import opt, period  # noqa: E402

from pwstablenet_amd import hipabi, routing, synth  # noqa: E402

NGF, N = 16, 2


def main():
    out = {}
    L = hipabi.lib()
    weights = synth.make_weights("W1", seed=123, ngf=NGF)
    netG = define_G(opt.input_nc, opt.output_nc, NGF, 'normal', 0.02)
    netG.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in weights})
    netG.cuda()
    netG.train()
    rs = np.random.RandomState(7)
    win = synth.make_window(N, 31, 256, seed=5)
    rgb = synth.make_frames(N, 3, 256, 256, seed=6) / 127.5 - 1
    unstable1 = torch.from_numpy(np.concatenate([win, rgb], 1).astype(np.float32)).cuda()
    unstable2 = torch.from_numpy(np.concatenate([synth.make_window(N, 31, 256, seed=8),
                                                 synth.make_frames(N, 3, 256, 256, seed=9) / 127.5 - 1], 1).astype(np.float32)).cuda()
    theta_adj = (np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (N, 1)) + 0.02 * rs.randn(N, 6)).astype(np.float32)
    feature_adjacent = torch.from_numpy(theta_adj).cuda()

    # ---- the training step's calls (main_new.py:101-118,195-197,214)
    L.pws_prof_enable(1)
    grid1, affine1 = netG(unstable1[:, 0:period + 1, :, :])
    fake1 = []
    for nl in range(3):
        fake1_temp = functional.grid_sample((unstable1[:, period + 1:period + 1 + 3, :, :] + 1) * 127.5, grid1[nl])
        fake1.append(fake1_temp / 127.5 - 1)
    fake1_gray = functional.grid_sample((unstable1[:, period // 2: period // 2 + 1:, :] + 1) * 127.5, grid1[2]) / 127.5 - 1
    grid2, affine2 = netG(unstable2[:, 0:period + 1, :, :])
    fake2 = []
    for nl in range(3):
        fake2_temp = functional.grid_sample((unstable2[:, period + 1:period + 1 + 3, :, :] + 1) * 127.5, grid2[nl])
        fake2.append(fake2_temp / 127.5 - 1)
    loss = 0
    for nl in range(3):
        fa = feature_adjacent.view(-1, 2, 3)
        grid = functional.affine_grid(fa, fake1[nl].size())
        output2_to_output1 = functional.grid_sample(fake2[nl], grid)
        loss = loss + torch.mean(torch.abs(output2_to_output1 - fake1[nl]))
    loss = loss + torch.mean(torch.abs(fake1_gray))
    loss.backward()
    torch.cuda.synchronize()
    names = [r[0] for r in hipabi.prof_collect(65536)]
    L.pws_prof_enable(0)
    out["train_kernels"] = {k: names.count(k) for k in ("grid_sample_fwd_kernel", "grid_sample_bwd_kernel", "affine_grid_kernel")}
    out["train_loss"] = float(loss)
    gw = netG.module.up1.mpconv[0].weight.grad
    out["train_grad_up1"] = gw.detach().cpu().numpy().astype(np.float64).ravel()[::97].tolist()
    out["train_fake1_2"] = fake1[2].detach().cpu().numpy()[:, :, ::16, ::16].ravel().tolist()

    # ---- the video loop's calls (main_new.py:697-716), gradients enabled as in the reference
    netG.eval()
    H, W = 720, 1280
    now = torch.from_numpy(synth.make_frames(1, 3, H, W, seed=11)).cuda()
    images = unstable1[:1, 0:period + 1, :, :]
    L.pws_prof_enable(1)
    grid = netG(images, False)
    grid = grid.permute(0, 3, 1, 2)
    m = torch.nn.UpsamplingBilinear2d(size=(H, W))
    grid_resize = m(grid)
    grid_resize = grid_resize.permute(0, 2, 3, 1)
    fake = functional.grid_sample(now, grid_resize)
    samples = fake[0, :, :, :].data.cpu().numpy()
    torch.cuda.synchronize()
    names = [r[0] for r in hipabi.prof_collect(65536)]
    L.pws_prof_enable(0)
    out["video_kernels"] = {k: names.count(k) for k in ("grid_sample_fwd_kernel", "upsample_bilinear_ac_kernel")}
    out["video_field"] = grid.detach().permute(0, 2, 3, 1).cpu().numpy()[:, ::8, ::8].ravel().tolist()
    out["video_samples"] = samples[:, ::24, ::40].ravel().tolist()
    out["video_is_module"] = isinstance(m, torch.nn.Module) and type(m).__name__ == "UpsamplingBilinear2d"

    # ---- CPU tensors pass through to torch untouched
    before = {k: list(v) for k, v in routing.stats.items()}
    c_img, c_grid = torch.rand(1, 3, 9, 11), torch.rand(1, 5, 7, 2) * 2 - 1
    a = functional.grid_sample(c_img, c_grid, align_corners=False)
    b = torch.nn.UpsamplingBilinear2d(size=(6, 8))(c_img)
    c = functional.affine_grid(torch.eye(2, 3).unsqueeze(0), (1, 3, 4, 5), align_corners=False)
    out["cpu_devices"] = [str(a.device), str(b.device), str(c.device)]
    out["cpu_passed"] = [routing.stats[k][1] - before[k][1] for k in ("grid_sample", "upsample", "affine_grid")]
    out["routed"] = {k: v[0] for k, v in routing.stats.items()}
    out["installed"] = routing.installed() and torch.nn.functional.grid_sample is functional.grid_sample
    print("DROPIN_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()

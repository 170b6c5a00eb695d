"""How much of the whole-network bounds (tests/test_hip_timed_path.py: 5e-4 on the field, 1e-3 on the warped frames, batch 8, fp32) the first layer's
arithmetic uses: the CPU restatement against the HIP path with wino5_first_kernel (PWS_OPT_EXPERIMENT 0) and with conv_first_kernel (26)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pwstablenet_amd import functional as PF, hipabi as A, synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from oracle import torch_ref  # noqa: E402

for kind in ("W1", "W2"):
    weights = synth.make_weights(kind, seed=123, ngf=64)
    params = [torch.from_numpy(v) for _, v in weights]
    x = torch.from_numpy(synth.noise_window(8, 31, 256, seed=123))
    fr = torch.from_numpy(synth.make_frames(8, 3, 256, 256, seed=321))
    with torch.no_grad():
        field = torch_ref.netg_forward(params, x, is_training=False)
        warped = torch.nn.functional.grid_sample(fr, field, mode="bilinear", padding_mode="zeros", align_corners=False)
    for exp in (0, 26):
        A.lib().pws_set_option(A.OPT_EXPERIMENT, exp)
        net = define_G(31, 2, 64, "normal", 0.02)
        net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in weights})
        net = net.cuda()
        with torch.no_grad():
            f = net(x.cuda(), False)
            w = PF.grid_sample(fr.cuda(), f)
        print("%s first layer %-18s: field max-abs err %.3g, warped-frame err %.3g (bounds 5e-4 / 1e-3)" % (
            kind, "F(2x2,5x5)" if exp == 0 else "direct (exp 26)", float((f.cpu() - field).abs().max()), float((w.cpu() - warped).abs().max()) / 127.5), flush=True)
    A.lib().pws_set_option(A.OPT_EXPERIMENT, 0)

#!/usr/bin/env python3
"""Per-launch timing of one training step (forward is_training=1 + backward + Adam) through the pws_prof_* hooks."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import spec, synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--list", type=int, default=0, help="also list the N slowest launches of any kernel")
    a = ap.parse_args()
    A.lib().pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))   # A/B switches of measured variants
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=64)})
    net = net.cuda()
    net.module.set_math(a.math)
    A.lib().pws_set_option(A.OPT_TWO_QUEUES, 0)  # one queue: per-kernel durations are not inflated by overlap
    opt = Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.999))
    x = torch.from_numpy(synth.noise_window(a.batch, 31, 256, 123)).cuda()
    fr = torch.from_numpy(synth.make_frames(a.batch, 3, 256, 256, 321)).cuda()
    tg = torch.roll(fr, shifts=(2, -3), dims=(2, 3))
    names = [ls.name.replace(".0", "").replace(".mpconv", "") for ls in spec.layer_specs()]

    def step():
        opt.zero_grad()
        grids, resid = net(x)
        loss = sum(torch.nn.functional.l1_loss(PF.grid_sample(fr, g) / 127.5 - 1, tg / 127.5 - 1) for g in grids)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.reps
    A.lib().pws_prof_enable(1)
    step()
    A.lib().pws_prof_enable(0)
    recs = A.prof_collect(1 << 16)
    agg = {}
    tot = 0.0
    for name, tag, fl, by, ms in recs:
        e = agg.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1; e[1] += fl; e[2] += ms; tot += ms  # noqa: E702
    print("%-44s %6s %10s %9s" % ("kernel", "calls", "ms", "TFLOP/s"))
    for k, (c, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print("%-44s %6d %10.3f %9.1f" % (k, c, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    print("sum of profiled kernels %.2f ms; wall per step %.2f ms (batch %d) -> %.1f samples/s" % (tot, wall * 1e3, a.batch, a.batch / wall))
    slow = sorted([r for r in recs if r[0] in ("wgrad_mfma_kernel", "wgrad_bf16_kernel", "conv_bf16_kernel", "conv_mfma_kernel<dgrad k4s2>", "conv_mfma_kernel<dgrad subpix k3s2>")],
                  key=lambda r: -r[4])[:14]
    for name, tag, fl, by, ms in slow:
        print("  %-40s %-26s %8.1f us %7.1f TF/s" % (name, names[tag] if 0 <= tag < len(names) else "-", ms * 1e3, fl / (ms * 1e-3) / 1e12))
    if a.list:   # every launch, slowest first, with its algorithmic bytes
        for name, tag, fl, by, ms in sorted(recs, key=lambda r: -r[4])[:a.list]:
            print("  . %-38s %-26s %8.1f us %7.1f TF/s %8.1f MB algorithmic %7.1f GB/s" % (name, names[tag] if 0 <= tag < len(names) else "-", ms * 1e3,
                                                                                    fl / (ms * 1e-3) / 1e12, by / 1e6, by / (ms * 1e-3) / 1e9))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-launch timing of one generator forward (hipEvents through the pws_prof_* hooks): which layer costs what.

    python tools/layer_profile.py [--batch 8] [--train] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import spec, synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32")
    a = ap.parse_args()
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=64)})
    net = net.cuda()
    net.module.set_math(a.math)
    A.lib().pws_set_option(A.OPT_EXPERIMENT, int(os.environ.get("PWS_EXPERIMENT", "0")))
    A.lib().pws_set_option(A.OPT_TWO_QUEUES, 0)  # one queue: per-kernel durations are not inflated by overlap
    x = torch.from_numpy(synth.noise_window(a.batch, 31, 256, 123)).cuda()
    names = [ls.name.replace(".0", "").replace(".mpconv", "") for ls in spec.layer_specs()]
    with torch.no_grad():
        for _ in range(2):
            net(x, a.train)
        torch.cuda.synchronize()
        A.lib().pws_prof_enable(1)
        for _ in range(a.reps):
            net(x, a.train)
        A.lib().pws_prof_enable(0)
    recs = A.prof_collect(1 << 16)
    per = len(recs) // a.reps
    tot = 0.0
    print("%-4s %-28s %-34s %9s %9s %8s" % ("#", "layer", "kernel", "us", "GFLOP", "TFLOP/s"))
    for i in range(per):
        ms = min(recs[i + r * per][4] for r in range(a.reps))
        name, tag, fl, by, _ = recs[i]
        tot += ms
        lname = names[tag] if 0 <= tag < len(names) else "-"
        print("%-4d %-28s %-34s %9.1f %9.3f %8.1f" % (i, lname, name, ms * 1e3, fl / 1e9, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    print("sum of kernel times: %.3f ms for batch %d  -> %.1f frames/s (kernels only)" % (tot, a.batch, a.batch / (tot * 1e-3)))


if __name__ == "__main__":
    main()

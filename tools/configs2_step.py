#!/usr/bin/env python3
"""BASELINE configs[2] as bench.py times it (objective.train_step: 32 item pairs -> 64 generator forwards, fused objective,
backward, fused Adam; bf16 math + storage), alone, for rocprofv3:   python tools/configs2_step.py [--items 32] [--reps 3]"""
import argparse
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--math", default="bf16")
    ap.add_argument("--per-rep", action="store_true", help="sync after every step and print when the host was done issuing / when the device was done")
    a = ap.parse_args()
    A.lib().pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
    with contextlib.redirect_stdout(sys.stderr):
        net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
    net = net.cuda()
    import time
    per = [] if a.per_rep else None
    t0 = time.perf_counter()
    dt, loss = bench.configs2_step_leg(net, torch.device("cuda"), a.items, a.math, a.reps, per_rep=per)
    if per:
        prev = per[0][0] - 1e-9
        print("leg total %.1f ms (with its warm-up step); per step: host issued after / device done after (ms):" % (1e3 * (time.perf_counter() - t0)))
        last = None
        for th, td in per:
            start = last if last is not None else None
            print("   host %s  device %s" % ("%.1f" % (1e3 * (th - start)) if start else "?", "%.1f" % (1e3 * (td - start)) if start else "?"))
            last = td
    st = torch.cuda.memory_stats()
    print("configs[2] step (%d item pairs, %s): %.2f ms, loss_g %.4f   [hipMalloc calls %d, hipFree calls %d, reserved %.1f GB, peak allocated %.1f GB]"
          % (a.items, a.math, 1e3 * dt, loss, st.get("num_device_alloc", -1), st.get("num_device_free", -1), st["reserved_bytes.all.current"] / 1e9,
             st["allocated_bytes.all.peak"] / 1e9))


if __name__ == "__main__":
    main()

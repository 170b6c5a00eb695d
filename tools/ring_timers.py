#!/usr/bin/env python3
"""Where conv_ring_kernel's matrix waves spend their cycles (diagnostic build, tools/ring_timers.sh):
    PWS_LIB_PATH=$PWD/tools/_bin/ring_timers.so python tools/ring_timers.py KIND N H W CIN COUT [dgrad]
Per matrix wave the kernel accumulates s_memrealtime (100 MHz) deltas: waiting at barrier B (operands of the next group not landed / other waves
late), group set-up, matrix phase, epilogue (barrier E included)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import conv_bench  # noqa: E402


def main():
    v = sys.argv[1:]
    t = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    os.environ["PWS_RING_TIMERS_PTR"] = str(t.data_ptr())
    os.environ.setdefault("CONV_BENCH_STORE16", "1")
    os.environ.setdefault("CONV_BENCH_ROTATE", "3")
    if len(v) > 6 and v[6] == "dgrad":
        conv_bench.bench_dgrad(v[0], int(v[1]), int(v[2]), int(v[3]), int(v[4]), int(v[5]))
    else:
        conv_bench.bench(v[0], int(v[1]), int(v[2]), int(v[3]), int(v[4]), int(v[5]), False, True)
    torch.cuda.synchronize()
    r = t.view(256, 8, 8).double().cpu() / 100.0      # s_memrealtime ticks (100 MHz) -> us
    tot = r[:, :, 4]
    live = tot > 0
    if not bool(live.any()):
        print("no timers written (not the diagnostic library, or the launch did not take the ring kernel)")
        return
    m = live.double()
    n = float(m.sum())
    groups, units = float((r[:, :, 5] * m).sum() / n) * 100.0, float((r[:, :, 6] * m).sum() / n) * 100.0
    life = float((tot * m).sum() / n)
    print("  matrix waves reporting: %d; groups per wave %.0f, units per wave %.1f; wave life %.1f us" % (n, groups, units, life))
    for i, nm in enumerate(["wait at B", "group set-up", "matrix phase", "epilogue"]):
        c = float((r[:, :, i] * m).sum() / n)
        print("  %-13s %8.1f us  %5.1f %%   per group %6.2f us   per unit %6.2f us" % (nm, c, 100 * c / life, c / groups, c / units))


if __name__ == "__main__":
    main()

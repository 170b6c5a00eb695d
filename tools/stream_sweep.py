#!/usr/bin/env python3
"""configs[4] streaming leg under load: VideoStabilizer(batch) x generator math on T decoded uint8 720p
frames, pinned host -> device -> window planes + netG + fused u8 warp + 2x INTER_AREA -> pinned host.  Reports, per setting, the
host-to-host rate, PCIe GB/s each way, and how much of the slowest of the three activities (upload, compute, download -- each
timed alone on the same clip) the pipelined run costs: overlap = max(alone) / pipelined wall (1.0 = perfectly hidden).
python tools/stream_sweep.py [--frames 256] [--batches 8,16,32] [--chunks 64]"""
import argparse
import contextlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.stream import VideoStabilizer  # noqa: E402


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def sweep(net, T=256, batches=(8, 16, 32), chunks=(64,), maths=("fp32", "bf16"), graph=(False, True)):
    dev = torch.device("cuda")
    u8_h = torch.randint(0, 256, (T, 720, 1280, 3), dtype=torch.uint8).pin_memory()
    out_h = torch.empty((T, 360, 640, 3), dtype=torch.uint8).pin_memory()
    u8_d = u8_h.to(dev)
    small_d = torch.empty((T, 360, 640, 3), dtype=torch.uint8, device=dev)
    t_up = timed(lambda: u8_d.copy_(u8_h, non_blocking=True))
    t_down = timed(lambda: out_h.copy_(small_d, non_blocking=True))
    rows = []
    for math in maths:
        net.module.set_math(math)
        for g in graph:
            net.module.enable_graph(g)
            for b in batches:
                for ch in chunks:
                    vs = VideoStabilizer(net, batch=b, swap_rb=True)
                    t_comp = timed(lambda: vs.run_video(u8_d, chunk=ch, half_size_output=True))      # frames resident: no PCIe
                    t_all = timed(lambda: vs.run_video(u8_h, chunk=ch, half_size_output=True))
                    rows.append({"math": math, "graph": g, "batch": b, "chunk": ch, "frames_per_s": round(T / t_all, 1),
                                 "frames_per_s_resident": round(T / t_comp, 1), "h2d_gb_per_s": round(T * 2.7648e-3 / t_all, 2),
                                 "d2h_gb_per_s": round(T * 0.6912e-3 / t_all, 2),
                                 "alone_ms": {"upload": round(1e3 * t_up, 1), "compute": round(1e3 * t_comp, 1), "download": round(1e3 * t_down, 1)},
                                 "pipelined_ms": round(1e3 * t_all, 1), "overlap": round(max(t_up, t_comp, t_down) / t_all, 3)})
    net.module.enable_graph(False)
    net.module.set_math("fp32")
    return {"frames": T, "pcie_alone_gb_per_s": {"h2d": round(T * 2.7648e-3 / t_up, 1), "d2h": round(T * 0.6912e-3 / t_down, 1)}, "rows": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--batches", default="8,16,32")
    ap.add_argument("--chunks", default="64")
    a = ap.parse_args()
    with contextlib.redirect_stdout(sys.stderr):
        net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
    net = net.cuda()
    r = sweep(net, a.frames, tuple(int(x) for x in a.batches.split(",")), tuple(int(x) for x in a.chunks.split(",")))
    print("PCIe alone: %s GB/s" % r["pcie_alone_gb_per_s"])
    for row in r["rows"]:
        print("%-5s graph=%-5s batch %2d chunk %3d: %7.1f f/s host-to-host (%7.1f resident)  h2d %.2f GB/s  alone up/comp/down %s ms  pipelined %.1f ms  overlap %.2f"
              % (row["math"], row["graph"], row["batch"], row["chunk"], row["frames_per_s"], row["frames_per_s_resident"], row["h2d_gb_per_s"],
                 "/".join(str(row["alone_ms"][k]) for k in ("upload", "compute", "download")), row["pipelined_ms"], row["overlap"]))
    print(json.dumps(r))


if __name__ == "__main__":
    main()

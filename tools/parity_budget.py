"""The parity ledger of the fp32 forward (BASELINE north_star: warped frames within 1e-3 max-abs of the reference CPU path).

tests/test_hip_timed_path.py holds the whole-network bounds at the timed batch (8, graph replay, two queues): 5e-4 on the field, 1e-3 on
the warped frames scaled to [-1, 1].  Every multiply-count reduction (Winograd F(2x2,3x3), F(2x2,2x2) per parity class, F(2x2,5x5)) spends
part of that margin; this tool says which kernel family spends how much: for the weight sets W1 and W2 it runs the product path, then the
path with ONE family at a time taken back to the plain direct fp32 kernel (PWS_OPT_EXPERIMENT 3000 + bit, csrc/conv_mfma.hip:
ledger_direct), then everything direct, each against the CPU restatement of the reference graph on the same inputs.

    python tools/parity_budget.py [--batch 8] > profiles/rNN_parity_budget.txt

A change that adds a reduction ships with its row in this table; the test's tripwire (W2 warped-frame error < 0.92e-3) names this tool.
(The oracle is the checker here, as in the tests: this is a measurement tool, not product code.)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import torch_ref  # noqa: E402
from pwstablenet_amd import functional as PF, hipabi as A, synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402

FAMILIES = [
    (0, "product path (all reductions on)"),
    (1, "first layer direct            (product: wino5_first_kernel, F(2x2,5x5))"),
    (2, "3x3 stride-1 layers direct    (product: wino_ring_kernel<F(2x2,3x3)>)"),
    (4, "transposed 4x4 layers direct  (product: wino_ring_kernel<convT4,F(2x2,2x2)>)"),
    (8, "conv_ringf_kernel -> conv_mfma_kernel (both direct: summation order only)"),
    (16, "conv_skinny_kernel -> conv_mfma_kernel (both direct: summation order only)"),
    (31, "everything on conv_mfma_kernel (no reduction anywhere)"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    L = A.lib()
    print("# parity ledger: fp32 forward, batch %d, graph replay + two queues, against oracle/torch_ref.py (CPU) on the same inputs" % a.batch)
    print("# bounds (tests/test_hip_timed_path.py): field 5e-4, warped frame 1e-3 (frames scaled to [-1, 1]); tripwire on W2: 0.92e-3")
    for kind in ("W1", "W2"):
        weights = synth.make_weights(kind, seed=123, ngf=64)
        params = [torch.from_numpy(v) for _, v in weights]
        x = torch.from_numpy(synth.noise_window(a.batch, 31, 256, seed=123))
        fr = torch.from_numpy(synth.make_frames(a.batch, 3, 256, 256, seed=321))
        with torch.no_grad():
            field = torch_ref.netg_forward(params, x, is_training=False)
            warped = torch.nn.functional.grid_sample(fr, field, mode="bilinear", padding_mode="zeros", align_corners=False)
        print("\n## %s   (|field| max %.3f)" % (kind, float(field.abs().max())))
        print("%-86s %12s %12s %14s" % ("path", "field err", "warped err", "warped - product"))
        base = None
        for mask, what in FAMILIES:
            L.pws_set_option(A.OPT_EXPERIMENT, 3000 + mask if mask else 0)
            try:
                net = define_G(31, 2, 64, "normal", 0.02)
                net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in weights})
                net = net.cuda()
                with torch.no_grad():
                    net.module.enable_graph(True)
                    net(x.cuda(), False)
                    f = net(x.cuda(), False)
                    w = PF.grid_sample(fr.cuda(), f)
                ferr = float((f.cpu() - field).abs().max())
                werr = float((w.cpu() - warped).abs().max()) / 127.5
            finally:
                L.pws_set_option(A.OPT_EXPERIMENT, 0)
            if base is None:
                base = werr
            print("%-86s %12.3e %12.3e %+14.3e" % (what, ferr, werr, werr - base), flush=True)
            del net
    print("\n# reading: a family's row LOWER than the product row by d means that family's reduction costs about d of the bound; the last row is")
    print("# what summation order and the fp32 heads alone leave (the floor no kernel choice goes below).")


if __name__ == "__main__":
    main()

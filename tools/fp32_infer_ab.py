#!/usr/bin/env python3
"""fp32 inference (batch 8, graph + two queues, as bench.py's headline leg) under PWS_EXPERIMENT values:
    python tools/fp32_infer_ab.py 0 26 ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402

net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=64)})
net = net.cuda()
x = torch.from_numpy(synth.noise_window(8, 31, 256, 123)).cuda()
for rep in range(2):
    for e in [int(v) for v in sys.argv[1:]] or [0]:
        A.lib().pws_set_option(100, e)
        net.module.enable_graph(False)
        net.module.enable_graph(True)
        with torch.no_grad():
            for _ in range(5):
                net(x, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                net(x, False)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print("exp %4d: %.3f ms per batch of 8 = %.0f frames/s" % (e, dt * 1e3, 8 / dt))
A.lib().pws_set_option(100, 0)

"""run_video(in_flight = 1 / 2) on a device-resident 720p clip: frames/s by generator batch, graph mode and math"""
import os, sys, time, contextlib
sys.path.insert(0, os.getcwd())
import torch
from pwstablenet_amd import synth
from pwstablenet_amd.lib.networks_cascading import define_G
from pwstablenet_amd.stream import VideoStabilizer
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(31, 2, 64, "normal", 0.02)
net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)}); net = net.cuda()
T = 192
u8 = torch.randint(0, 256, (T, 720, 1280, 3), dtype=torch.uint8, device="cuda")
if os.environ.get("HOST") == "1":
    u8 = u8.cpu().pin_memory()
for math in ("fp32", "bf16"):
    net.module.set_math(math)
    for graph in (False, True):
        net.module.enable_graph(graph)
        for SB in (8, 32):
            vs = VideoStabilizer(net, batch=SB, swap_rb=True)
            row = []
            for k in (1, 2, 1, 2):
                vs.run_video(u8[:4 * SB], chunk=SB, half_size_output=True, in_flight=k)
                torch.cuda.synchronize(); t = time.perf_counter()
                vs.run_video(u8, chunk=SB, half_size_output=True, in_flight=k)
                torch.cuda.synchronize(); row.append("%d: %.0f" % (k, T / (time.perf_counter() - t)))
            print(math, "graph" if graph else "eager", "batch/chunk", SB, "  ".join(row), flush=True)
net.module.enable_graph(False); net.module.set_math("fp32")

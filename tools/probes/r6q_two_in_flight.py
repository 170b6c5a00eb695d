"""Two batch-8 forwards in flight, EAGER launches (round 5 tried graphs: graph launches on two streams did not overlap): two generators with the same weights, each on
its own torch stream, alternating steps without a sync in between, against one generator.  usage: python tools/probes/r6q_two_in_flight.py [fp32|bf16]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pwstablenet_amd import functional as PF, synth
from pwstablenet_amd.lib.networks_cascading import define_G

math = sys.argv[1] if len(sys.argv) > 1 else "fp32"
w = synth.make_weights("W1", 123, ngf=64)
nets = []
for i in range(4):
    n = define_G(31, 2, 64, "normal", 0.02)
    n.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in w})
    n = n.cuda()
    n.module.set_math(math)
    nets.append(n)
xs = [torch.from_numpy(synth.noise_window(8, 31, 256, 123 + i)).cuda() for i in range(4)]
fr = torch.from_numpy(synth.make_frames(8, 3, 256, 256, seed=321)).cuda()
streams = [torch.cuda.Stream() for _ in range(4)]


def run(depth, graph, two_queues, steps=60):
    for n in nets:
        n.module.enable_graph(graph)
        n.module.two_queues = two_queues
    with torch.no_grad():
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                k = s % depth
                with torch.cuda.stream(streams[k]):
                    PF.grid_sample(fr, nets[k](xs[k], False))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
    return 8 / dt


for graph in (True, False):
    for tq in (True, False):
        r = [run(d, graph, tq) for d in (1, 2, 3, 4)]
        print("%s %-5s %-10s: 1 / 2 / 3 / 4 in flight: %s f/s" % (math, "graph" if graph else "eager", "two queues" if tq else "one queue", " / ".join("%.0f" % v for v in r)), flush=True)

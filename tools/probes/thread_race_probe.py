#!/usr/bin/env python3
"""Which tensor of a train_step first differs when two steps run on two host threads? (round 4: tests/test_hip_threads.py bf16 case)
python tools/probes/thread_race_probe.py [bf16|fp32] [rounds] [two_queues 0|1]"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.objective import StabObjective, u8_normalize  # noqa: E402

math = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3:
    A.lib().pws_set_option(A.OPT_TWO_QUEUES, int(sys.argv[3]))
NGF, ITEMS = 32, 2


def make_work(kind, seed):
    def work():
        net = define_G(31, 2, NGF, "normal", 0.02)
        net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(kind, seed=seed, ngf=NGF)})
        net = net.cuda()
        net.module.set_math(math)
        net.module.deterministic = True
        images1, features1, _a1, images2, features2, _a2, adj = [torch.from_numpy(t).cuda() for t in synth.make_train_batch(ITEMS, seed=seed)]
        n, c, h, w = images1.shape
        win = torch.empty((2 * n, 31, h, w), device="cuda")
        rest = torch.empty((2 * n, c - 31, h, w), device="cuda")
        for half, img in enumerate((images1, images2)):
            u8_normalize(img[:, :31], win[half * n:(half + 1) * n])
            u8_normalize(img[:, 31:], rest[half * n:(half + 1) * n])
        features = torch.cat([features1, features2], 0).float()
        grids, resid = net(win)
        obj = StabObjective(batchSize=ITEMS)
        out = obj(grids, resid, rest[:, 0:3], rest[:, 3:], features, adj, deterministic=True)
        rec = {"win": win, "grid0": grids[0], "grid1": grids[1], "grid2": grids[2], "resid2": resid[2], "fake0": out.fake[0], "fake2": out.fake[2]}
        for k in ("loss_g", "loss_mse", "loss_feature", "loss_delta", "loss_g2", "loss_pixel"):
            rec[k] = out[k].detach().reshape(1)
        rec = {k: v.detach().clone() for k, v in rec.items()}
        out.loss_g.backward()
        for i, p in enumerate(net.parameters()):
            rec["g%02d" % i] = p.grad.detach().clone()
        torch.cuda.current_stream().synchronize()
        return {k: v.cpu() for k, v in rec.items()}
    return work


def in_threads(fns):
    out = [None] * len(fns)
    go = threading.Barrier(len(fns))

    def body(i):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            go.wait()
            out[i] = fns[i]()
        s.synchronize()
    ts = [threading.Thread(target=body, args=(i,)) for i in range(len(fns))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return out


third = sys.argv[4] if len(sys.argv) > 4 else "none"    # none | eager | graph | capture


def make_infer():
    net = define_G(31, 2, NGF, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=5, ngf=NGF)})
    net = net.cuda()
    x = [torch.from_numpy(synth.noise_window(2, 31, 256, seed=33 + r)).cuda() for r in range(2)]
    if third == "graph":
        net.module.enable_graph(True)
        with torch.no_grad():
            net(x[0], False), net(x[1], False)
        torch.cuda.synchronize()

    def work():
        res = {}
        with torch.no_grad():
            if third == "capture":     # the capture itself happens on this thread, beside the training threads
                net.module.enable_graph(False)
                net.module.enable_graph(True)
            for r in range(8):
                res["f%d" % r] = net(x[r % 2], False).clone()
        torch.cuda.current_stream().synchronize()
        return {k: v.cpu() for k, v in res.items()}
    return work


works = [make_work("W1", 11), make_work("W2", 12)]
serial = [w() for w in works]
serial2 = [w() for w in works]
for t in range(2):
    bad = [k for k in serial[t] if not torch.equal(serial[t][k], serial2[t][k])]
    print("serial vs serial, work %d: differing %s" % (t, bad[:8]))
if third != "none":
    works.append(make_infer())
    serial.append(works[2]())
for r in range(rounds):
    got = in_threads(works)
    for t in range(len(works)):
        bad = [(k, float((got[t][k].double() - serial[t][k].double()).abs().max()), float(serial[t][k].double().abs().max())) for k in serial[t]
               if not torch.equal(got[t][k], serial[t][k])]
        print("round %d work %d: %d differing; first: %s" % (r, t, len(bad), bad[:6]))

#!/usr/bin/env python3
"""Data-parallel training step loop on synthetic data: one process per GPU, replicated weights, ONE flat-bucket RCCL
all-reduce of the 48.5 M gradients per step (the reference: single-process nn.DataParallel, lib/networks_cascading.py:51-52;
train loop main_new.py:81-216 -- two generator forwards per item pair, grid_sample of the RGB frame per stage, L1 loss,
backward, Adam(beta1=0.5)).

    python tools/train_ddp.py --batch 8 --steps 5                       # 1 GPU
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_ddp.py --batch 32 --steps 10
Prints one JSON line on rank 0: samples/s over all ranks (a "sample" = one item pair = two forwards, as in the reference).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pwstablenet_amd import distributed as D  # noqa: E402
from pwstablenet_amd import functional as PF  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="item pairs per GPU per step")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    rank, world = D.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=64)})  # same on all ranks
    net = net.to(dev)
    opt = Adam(net.parameters(), lr=a.lr, betas=(0.5, 0.999))
    B = a.batch
    x1 = torch.from_numpy(synth.noise_window(B, 31, 256, 100 + rank)).to(dev)
    x2 = torch.from_numpy(synth.noise_window(B, 31, 256, 200 + rank)).to(dev)
    rgb = torch.from_numpy(synth.make_frames(B, 3, 256, 256, 300 + rank)).to(dev)
    stable = torch.roll(rgb, shifts=(2, -3), dims=(2, 3))

    def step():
        opt.zero_grad()
        loss = 0.0
        for x in (x1, x2):  # frame t and frame t+1 (main_new.py:101,112)
            grids, _ = net(x)
            for g in grids:
                fake = PF.grid_sample(rgb, g) / 127.5 - 1  # main_new.py:106-107
                loss = loss + F.l1_loss(fake, stable / 127.5 - 1)
        loss.backward()
        D.allreduce_gradients(list(net.parameters()))
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank == 0:
        print(json.dumps({"metric": "training item pairs/s (2 forwards + backward + all-reduce + Adam per pair), whole job",
                          "value": round(world * B * a.steps / dt, 2), "unit": "pairs/s", "n_gpus": world,
                          "ms_per_step": round(1e3 * dt / a.steps, 2), "dtype": "f32", "data": "synthetic",
                          "loss": round(float(loss), 6), "batch_per_gpu": B}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Data-parallel training loop on synthetic data: one process per GPU, replicated weights, the generator step of the
reference's train() (main_new.py:84-216: two generator forwards per item pair -- batched as one --, the fused warp / L1 /
temporal / feature / shape objective, optionally the VGG-16 perceptual term, backward, fused Adam(beta1=0.5)) and the RCCL
all-reduce of the 48.5 M gradients, either after backward or overlapped with it (the reference: single-process
nn.DataParallel, lib/networks_cascading.py:51-52).

    python tools/train_ddp.py --batch 8 --steps 5                                   # 1 GPU
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_ddp.py --batch 32 --steps 10 --math bf16 --overlap
Prints one JSON line on rank 0: item pairs/s over all ranks.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import distributed as D  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.objective import StabObjective, train_step  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="item pairs per GPU per step")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--overlap", action="store_true", help="all-reduce overlapped with backward (distributed.OverlappedGradSync)")
    ap.add_argument("--vgg", action="store_true", help="add the VGG-16 perceptual term (random weights: no pretrained ones offline)")
    a = ap.parse_args()
    rank, world = D.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=64)})  # same on all ranks
    net = net.to(dev)
    net.module.set_math(a.math)
    D.broadcast_parameters(net, src=0)   # replicas start equal whatever each rank initialised / loaded
    opt = Adam(net.parameters(), lr=a.lr, betas=(0.5, 0.999))
    small = synth.make_train_batch(min(a.batch, 4), seed=100 + rank)
    rep = (a.batch + 3) // 4
    batch = [torch.from_numpy(t).repeat((rep,) + (1,) * (t.ndim - 1))[:a.batch].to(dev) for t in small]
    objective = StabObjective(batchSize=a.batch)
    perceptual = None
    if a.vgg:
        from pwstablenet_amd.perceptual import GeneratorLoss, VGG16Features, perceptual_term
        perceptual = perceptual_term(GeneratorLoss(VGG16Features(a.math).init_random(0)).to(dev))
    sync = None
    if a.overlap:
        D.enable_overlapped_grad_sync(net)
    elif world > 1:
        sync = D.allreduce_gradients

    def step():
        return train_step(net, opt, batch, objective, perceptual=perceptual, sync_gradients=sync)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank == 0:
        print(json.dumps({"metric": "training item pairs/s (2 forwards + objective + backward + all-reduce + Adam per pair), whole job",
                          "value": round(world * a.batch * a.steps / dt, 2), "unit": "pairs/s", "n_gpus": world,
                          "ms_per_step": round(1e3 * dt / a.steps, 2), "dtype": a.math, "data": "synthetic",
                          "grad_sync": "overlapped" if a.overlap else ("after backward" if world > 1 else "none"),
                          "vgg": bool(a.vgg), "loss_g": round(float(out.loss_g.detach()), 4), "batch_per_gpu": a.batch}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""One rank of the data-parallel equivalence check (test infrastructure; started as fresh child processes by
tests/test_hip_ddp.py before they touch the GPU).  Every rank sits on cuda:0 and the process group runs over gloo -- the
one-GPU box cannot run RCCL across ranks, but everything above the transport is the product path: broadcast_parameters,
train_step, allreduce_gradients / OverlappedGradSync (reference: nn.DataParallel, lib/networks_cascading.py:51-52; the step
main_new.py:101-118,214-216).

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  argv: mode math items_total ngf out_dir
mode: single (world 1, all items) | allreduce (allreduce_gradients after backward) | overlap (OverlappedGradSync, 4 runs) |
      slab (OverlappedGradSync(nparts=1): the whole gradient slab averaged in place before the one unpack) |
      rccl1 / rccl1_overlap (world 1, backend "nccl" = RCCL through distributed.init_from_env exactly as bench.py does, collectives
      FORCED in the one-rank group: first contact with librccl, its streams and the ordering against the backward's side queue).
Rank 0 writes out_dir/<mode>.pt = {grads, weights_before, weights_after, loss, collectives}.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, math, items_total, ngf, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.argv = sys.argv[:1]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pwstablenet_amd import distributed as D  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402
from pwstablenet_amd.lib.networks_cascading import define_G  # noqa: E402
from pwstablenet_amd.objective import StabObjective, train_step  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rccl = mode.startswith("rccl1")
    if rccl:
        os.environ["PWS_FORCE_PROCESS_GROUP"] = "1"
        assert D.init_from_env("nccl") == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    elif world > 1:
        dist.init_process_group("gloo")
    assert (mode == "single" or rccl) == (world == 1)
    torch.manual_seed(1000 + rank)               # define_G draws from the LOCAL RNG: the replicas differ ...
    net = define_G(31, 2, ngf, "normal", 0.02)
    if rank == 0:                                # ... and only rank 0 "loads the checkpoint"
        net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", 123, ngf=ngf)})
    net = net.to(dev)
    net.module.set_math(math)
    if world > 1:
        try:   # the replicas really differ before the broadcast; check=True is collective and raises on EVERY rank
            D.broadcast_parameters(net, src=0, check=True)
            raise SystemExit("replicas unexpectedly equal before the broadcast")
        except RuntimeError as e:
            assert "not identical" in str(e), e
        D.broadcast_parameters(net, src=0)
        D.broadcast_parameters(net, src=0, check=True)
    per = items_total // world
    full = synth.make_train_batch(items_total, seed=77)
    batch = [torch.from_numpy(t[rank * per:(rank + 1) * per]).to(dev) for t in full]
    objective = StabObjective(batchSize=per)     # the feature term divides by opt.batchSize (lib/utils.py:347): the local batch
    opt = Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    before = [p.detach().clone() for p in net.parameters()]
    sync, gs = None, None
    if mode == "allreduce":
        ncoll = [0]

        def sync(params):
            ncoll[0] = D.allreduce_gradients(params)
    elif mode == "overlap":
        gs = D.enable_overlapped_grad_sync(net, nparts=4, broadcast=False)
    elif mode == "slab":
        gs = D.enable_overlapped_grad_sync(net, nparts=1, broadcast=False)
    elif rccl:
        net.module.deterministic = True           # bit-equality with the no-exchange run is the check
        D.broadcast_parameters(net, src=0)        # world 1: returns before any collective ...
        t = torch.ones(4, device=dev)
        dist.broadcast(t, src=0)                  # ... so touch the broadcast path by hand
        gs = D.enable_overlapped_grad_sync(net, nparts=4 if mode == "rccl1_overlap" else 1, broadcast=False, force=True)
    if mode == "single" and os.environ.get("PWS_DDP_DETERMINISTIC") == "1":
        net.module.deterministic = True
    out = train_step(net, opt, batch, objective, sync_gradients=sync)
    torch.cuda.synchronize()
    from pwstablenet_amd.objective import LOSS_NAMES
    losses = torch.stack([out[k].detach().double().cpu().reshape(()) for k in LOSS_NAMES])
    if world > 1:
        dist.all_reduce(losses)                   # SUM over ranks; the test turns the mean-type terms into means
    if rank == 0:
        torch.save({"grads": [p.grad.detach().cpu() for p in net.parameters()], "before": [b.cpu() for b in before],
                    "after": [p.detach().cpu() for p in net.parameters()], "losses": dict(zip(LOSS_NAMES, losses.tolist())), "world": world,
                    "collectives": (gs.collectives if gs is not None else (ncoll[0] if mode == "allreduce" else 0)),
                    "bytes_reduced": gs.bytes_reduced if gs is not None else 0, "backend": dist.get_backend() if dist.is_initialized() else None},
                   os.path.join(out_dir, mode + ".pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

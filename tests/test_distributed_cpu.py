"""world_size-2 tests of the multi-GPU host logic on CPU (gloo): frame sharding with window halos (no collective on
the data path), MAX-over-ranks timing, flat-bucket gradient averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pwstablenet_amd import distributed as D


def test_shard_frames_cover_exactly_once_with_halo():
    for frames in (0, 1, 7, 31, 100, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, e, rs, re_ = D.shard_frames(frames, r, world, halo=15)
                assert 0 <= rs <= s <= e <= re_ <= frames
                assert s - rs == min(15, s) and re_ - e == min(15, frames - e)
                seen += list(range(s, e))
            assert seen == list(range(frames))
            sizes = [D.shard_frames(frames, r, world)[1] - D.shard_frames(frames, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_frames(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    # timing: the slowest rank defines the step
    assert D.max_over_ranks(1.0 + rank) == float(world)
    # gradient averaging over flat buckets (tiny bucket size forces several collectives)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((5, 3), (17,), (2, 3, 4), (1,))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    ncoll = D.allreduce_gradients(params, bucket_bytes=64)
    mean = sum(range(1, world + 1)) / world
    for i, p in enumerate(params):
        assert torch.allclose(p.grad, torch.full_like(p, mean * (i + 1)))
    # inference sharding: ranks process disjoint frames, results gathered for the check only
    s, e, _, _ = D.shard_frames(10, rank, world)
    mine = torch.zeros(10)
    mine[s:e] = 1
    dist.all_reduce(mine)
    assert torch.equal(mine, torch.ones(10))
    q.put((rank, ncoll))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == [0, 1] and all(g[1] >= 2 for g in got)

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
    # the generator's gradient slab: ranges of ONE flat buffer averaged in place (views: no flatten / copy back), the rest untouched
    slab = torch.arange(100, dtype=torch.float32) * float(rank + 1)
    keep = slab.clone()
    ptr = slab.data_ptr()
    assert D.allreduce_slab(slab, [(0, 10), (40, 90)], bucket_bytes=64) == 1 + 4      # 16 floats per message
    want = keep.clone()
    for a, b in ((0, 10), (40, 90)):
        want[a:b] = torch.arange(a, b, dtype=torch.float32) * mean
    assert torch.allclose(slab, want) and slab.data_ptr() == ptr
    assert D.allreduce_slab(torch.zeros(4), []) == 0
    # replicas start equal: parameters + BatchNorm buffers of rank 0 reach every rank as one flat bucket
    torch.manual_seed(100 + rank)   # define_G initialises from the LOCAL torch RNG: the ranks differ before the broadcast
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    m[1].running_mean.fill_(float(rank + 1))
    m[1].num_batches_tracked.fill_(5 * (rank + 1))
    # check=True: EVERY rank raises when ANY differs (the src rank too -- a rank that carried on would hang in its next collective)
    with pytest.raises(RuntimeError, match="not identical"):
        D.broadcast_parameters(m, src=0, check=True)
    versions = [p._version for p in m.parameters()]
    nfl = D.broadcast_parameters(m, src=0)
    assert nfl == sum(p.numel() for p in m.parameters()) + 8
    assert all(p._version > v for p, v in zip(m.parameters(), versions))   # the packed-weight cache keys on the version counters
    D.broadcast_parameters(m, src=0, check=True)
    assert float(m[1].running_mean[0]) == 1.0 and int(m[1].num_batches_tracked) == 5   # integer buffers follow (second bucket)
    chk = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).double().sum().reshape(1)
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(b, both[0]) for b in both)
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


def test_bench_launches_its_own_ranks_and_refuses_a_world_size_mismatch():
    """`python bench.py --gpus N` without a launcher must start N ranks itself: here with the control-plane
    check hook (no GPU work) over gloo.  A WORLD_SIZE that contradicts --gpus is an error, not a warning."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PWS_BENCH_LAUNCH_CHECK="1", PWS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "gpus_arg": 2, "max_over_ranks": 2.0, "rccl_ranks": 0,
                    "control_plane": "gloo (PWS_BENCH_BACKEND test hook: RCCL not exercised)", "self_launched": True}
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr

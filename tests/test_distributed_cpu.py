"""world_size-2 and world_size-8 tests of the multi-GPU host logic on CPU (gloo): frame sharding with window halos (no collective on
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
    # the REAL gradient slab of the reference-size generator (ngf 64: 46 layers = 92 tensors, 194 MB), exchanged as the overlapped
    # backward does at nparts = 4: after each run the ranges of the layers pws_netg_backward_plan reports final, merged where they
    # abut (autograd._NetGTrain.backward), in 64 MB messages -- every float of every layer averaged exactly once, the padding and
    # the message count as on the GPU
    from pwstablenet_amd import hipabi as A
    import ctypes
    L = A.lib()
    nl, nparts = 46, 4
    first, count = (ctypes.c_size_t * nl)(), (ctypes.c_size_t * nl)()
    assert L.pws_netg_grad_layout(31, 64, first, count) == 0
    plan = (ctypes.c_ubyte * nl)()
    assert L.pws_netg_backward_plan(31, 64, nparts, plan) == 0
    nfl = L.pws_netg_grad_floats(31, 64)
    assert nfl >= 48535944 and sum(count) == nfl and all(first[i] + count[i] == first[i + 1] for i in range(nl - 1))
    assert set(plan) == set(range(nparts)) and plan[0] == nparts - 1   # `transfer` (first op of the forward) is final in the last run
    one = (ctypes.c_ubyte * nl)()
    assert L.pws_netg_backward_plan(31, 64, 1, one) == 0 and not any(one)
    pattern = (torch.arange(nfl, dtype=torch.float32) % 1021.0) - 510.0
    gslab = pattern * float(rank + 1)
    n_msgs, seen = 0, torch.zeros(nl, dtype=torch.int32)
    for part in range(nparts):
        newly = [i for i in range(nl) if plan[i] == part]
        ranges = []
        for i in newly:
            if ranges and ranges[-1][1] == first[i]:
                ranges[-1][1] = first[i] + count[i]
            else:
                ranges.append([first[i], first[i] + count[i]])
        n_msgs += D.allreduce_slab(gslab, ranges)
        seen[newly] += 1
    assert bool((seen == 1).all())
    assert torch.allclose(gslab, pattern * mean, rtol=1e-6, atol=1e-4)
    assert 4 <= n_msgs <= 16, n_msgs            # a few large messages (GPU: 9 at nparts 4), never one per tensor
    # configs[4]: a 1 001-frame clip sharded over the ranks, 15-frame halos, every frame owned once
    s, e, rs, re_ = D.shard_frames(1001, rank, world, halo=15)
    own = torch.zeros(1001)
    own[s:e] = 1
    dist.all_reduce(own)
    assert torch.equal(own, torch.ones(1001)) and rs == max(0, s - 15) and re_ == min(1001, e + 15)
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


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_gloo(world):
    """world 2 and world 8 (BASELINE configs[3] / [4] name 8 GPUs: the control plane -- rendezvous, slab exchange plan, frame
    shards, broadcast, MAX timing -- rehearsed at that size on CPU; the data plane is RCCL's)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == list(range(world)) and all(g[1] >= 2 for g in got)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks_and_refuses_a_world_size_mismatch(world):
    """`python bench.py --gpus N` without a launcher must start N ranks itself: here with the control-plane
    check hook (no GPU work) over gloo.  A WORLD_SIZE that contradicts --gpus is an error, not a warning."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PWS_BENCH_LAUNCH_CHECK="1", PWS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, bench, "--gpus", str(world), "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": world, "gpus_arg": world, "max_over_ranks": float(world), "rccl_ranks": 0,
                    "control_plane": "gloo (PWS_BENCH_BACKEND test hook: RCCL not exercised)", "self_launched": True}
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", str(world)], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr

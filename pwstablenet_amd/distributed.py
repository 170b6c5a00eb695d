"""One process per GPU over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests).

Replaces the reference's single-process ``nn.DataParallel`` (reference lib/networks_cascading.py:51-52):
  * inference: frames are independent units -> contiguous chunk per rank, NO data-path collective; the streaming
    loop needs the 15 frames before and after each frame (31-frame window, reference main_new.py:622-673), so a
    chunk carries a halo of ``period//2`` frames on both sides;
  * training: replicated weights + ONE gradient all-reduce per step, issued as a few large flat buckets
    (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: large messages, few of them), sum then divide by world.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # PWS_FORCE_PROCESS_GROUP=1: a group of ONE rank as well (RCCL's communicator, streams and IPC set-up on a single GPU)
    if (world > 1 or os.environ.get("PWS_FORCE_PROCESS_GROUP") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_frames(num_frames, rank, world, halo=15):
    """Contiguous chunk of frame indices for ``rank`` plus the window halo it must also read.

    Returns (start, stop, read_start, read_stop): the rank stabilises frames [start, stop) and needs source frames
    [read_start, read_stop) (clamped to the video; the reference repeats the first/last frame at the ends,
    main_new.py:627-633,653-660).  Chunks differ by at most one frame and cover every frame exactly once.
    """
    if world < 1 or not (0 <= rank < world) or num_frames < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(num_frames, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return start, stop, max(0, start - halo), min(num_frames, stop + halo)


def max_over_ranks(value, device=None):
    """MAX-reduce of a Python float (bench timing: the slowest rank defines the step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _average_in_place(flat):
    """ONE collective that leaves the mean over ranks in ``flat``.  RCCL averages inside the collective (ReduceOp.AVG: no extra
    pass over the buffer); gloo (the CPU tests' backend) has no AVG: SUM, then one division."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())


def allreduce_slab(slab, ranges, bucket_bytes=64 << 20, force=False):
    """Averages ``slab[a:b]`` for every (a, b) of ``ranges`` over the ranks IN PLACE -- the slab is the generator's flat gradient
    buffer (``pws_netg_grad_floats``), the collectives run on views of it: no flatten, no copy back (under RCCL the wire is the
    only traffic besides the collective's own read + write).  Ranges longer than ``bucket_bytes`` go out as several messages
    (xGMI is point-to-point: a few large messages, pipelined).  On the CURRENT stream.  ``force``: issue the collectives in a
    group of ONE rank too (first contact with RCCL on a single GPU).  Returns the number of collectives issued."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return 0
    step = max(1, bucket_bytes // slab.element_size())
    n_coll = 0
    for a, b in ranges:
        for lo in range(a, b, step):
            _average_in_place(slab[lo:min(b, lo + step)])
            n_coll += 1
    return n_coll


def allreduce_tensors(tensors, bucket_bytes=64 << 20, force=False):
    """Averages the given fp32 tensors over ranks in place, in flat buckets of ~bucket_bytes, on the CURRENT stream (separate
    tensors have to be flattened and copied back: the generator's own backward uses ``allreduce_slab`` on its gradient slab
    instead).  Returns the number of collectives issued (0 without a process group)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return 0
    n_coll, i = 0, 0
    while i < len(tensors):
        j, size = i, 0
        while j < len(tensors) and (size == 0 or size + tensors[j].numel() * 4 <= bucket_bytes):
            size += tensors[j].numel() * 4
            j += 1
        flat = torch.cat([g.reshape(-1) for g in tensors[i:j]])
        _average_in_place(flat)
        off = 0
        for g in tensors[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        i = j
    return n_coll


def allreduce_gradients(params, bucket_bytes=64 << 20, force=False):
    """Averages ``p.grad`` over ranks in place, in flat buckets of ~bucket_bytes (generic: any parameter list -- it flattens and
    copies back; for the generator prefer ``enable_overlapped_grad_sync(netG, nparts=1)``, whose backward all-reduces its
    gradient slab in place before unpacking).

    Shared-weight gradients (stages 2 and 3 use the same modules) are already accumulated locally by backward before
    this is called.  Returns the number of collectives issued.
    """
    return allreduce_tensors([p.grad for p in params if p.grad is not None], bucket_bytes, force)


def broadcast_parameters(netG, src=0, check=False):
    """Makes every rank start from rank ``src``'s generator: parameters AND buffers (BatchNorm running statistics) go out as
    ONE flat fp32 bucket (194 MB: a single large xGMI message, not 92 small ones) and are copied back in place (version
    counters bump, so the packed-weight cache re-packs); integer buffers (BatchNorm's ``num_batches_tracked``) follow in a
    second, int64 bucket.  ``define_G`` initialises from the local torch RNG and a checkpoint may have been loaded on one rank
    only: without this the replicas diverge silently.  The reference's ``nn.DataParallel`` re-broadcasts rank 0's parameters on
    every forward (lib/networks_cascading.py:51-52); one process per GPU does it once.
    This is a COLLECTIVE: every rank of the default group must call it, with the tensors where the backend can reach them
    (device-resident under RCCL -- call ``netG.cuda()`` first).
    check=True: instead of overwriting, compare with ``src``'s values; the verdict is MAX-reduced, so EVERY rank raises together
    if ANY rank differs (no rank runs on into the next collective and hangs).  Returns the number of floats."""
    target = getattr(netG, "module", netG)
    every = list(target.parameters()) + list(target.buffers())
    tensors = [t for t in every if t.dtype.is_floating_point]   # copy_ under no_grad bumps ._version
    ints = [t for t in every if not t.dtype.is_floating_point]
    if not tensors:
        return 0
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sum(t.numel() for t in tensors)
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
        mine = flat.clone() if check else None
        dist.broadcast(flat, src=src)
        iflat = imine = None
        if ints:
            iflat = torch.cat([t.detach().reshape(-1).to(torch.int64) for t in ints])
            imine = iflat.clone() if check else None
            dist.broadcast(iflat, src=src)
        if check:
            ndiff = int((mine != flat).sum()) + (int((imine != iflat).sum()) if ints else 0)
            worst = torch.tensor([float(ndiff)], dtype=torch.float64, device=flat.device)
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            if float(worst.item()) > 0:
                raise RuntimeError("broadcast_parameters(check=True): the replicas are not identical -- rank %d differs from rank "
                                   "%d in %d values (worst rank: %d values)" % (dist.get_rank(), src, ndiff, int(worst.item())))
            return flat.numel()
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        ioff = 0
        for t in ints:
            t.copy_(iflat[ioff:ioff + t.numel()].view_as(t).to(t.dtype))
            ioff += t.numel()
    return off


class OverlappedGradSync:
    """Gradient exchange overlapped with backward.  ``enable_overlapped_grad_sync(netG)`` attaches one to the
    generator; its backward then runs as ``nparts`` consecutive runs of the reversed layer tape (``pws_netg_backward_part``)
    and, after each run, the weight gradients that are already final (the C side reports them) are unpacked and all-reduced
    on a second stream while the next run computes.  ``p.grad`` arrive averaged: no ``allreduce_gradients`` call after
    ``backward()``.  Without a process group (one GPU) the collectives are skipped, everything else runs the same."""

    def __init__(self, nparts=4, bucket_bytes=64 << 20, force=False):
        if nparts < 1:
            raise ValueError("nparts must be >= 1")
        self.nparts, self.bucket_bytes = int(nparts), int(bucket_bytes)
        self.force = bool(force)   # issue the collectives in a one-rank group as well (RCCL on a single GPU)
        self._streams = {}
        self.collectives = 0   # issued by the last backward (diagnostics)
        self.bytes_reduced = 0  # by the last backward

    def stream(self, device):
        key = str(device)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device)
        return self._streams[key]

    def allreduce(self, tensors):
        self.collectives += allreduce_tensors(tensors, self.bucket_bytes, self.force)

    def allreduce_slab(self, slab, ranges):
        """In-place average of the given [a, b) ranges of the gradient slab (see ``allreduce_slab``)."""
        k = allreduce_slab(slab, ranges, self.bucket_bytes, self.force)
        if self.collectives == 0:
            self.bytes_reduced = 0
        if k:
            self.bytes_reduced += sum(b - a for a, b in ranges) * slab.element_size()
        self.collectives += k


def enable_overlapped_grad_sync(netG, nparts=4, bucket_bytes=64 << 20, broadcast=True, force=False):
    """netG: what ``define_G`` returned (or its ``.module``).  Returns the OverlappedGradSync; ``netG.grad_sync = None`` turns
    it off again.  broadcast: first make every replica equal to rank 0's (``broadcast_parameters`` -- a COLLECTIVE: every
    rank must make this call, after ``netG.cuda()`` under RCCL; pass broadcast=False to attach the exchange without it)."""
    target = getattr(netG, "module", netG)
    if broadcast:
        broadcast_parameters(target, src=0)
    target.grad_sync = OverlappedGradSync(nparts, bucket_bytes, force)
    return target.grad_sync

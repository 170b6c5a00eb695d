#!/usr/bin/env python3
"""Does any kernel of a training step write outside the buffer it was given?  Every CUDA tensor torch allocates during the step gets a
guard band in front of and behind it (0xA5 bytes); after the step every band must be intact.  An out-of-bounds write is harmless by luck in a
single-threaded run (it lands in a free block or in the next temporary before that is written) and shows as run-to-run differences once
the allocator's layout depends on thread timing (tools/probes/thread_grad_probe.py).
usage: python tools/probes/guard_band_probe.py [math: bf16|fp32] [ngf] [items]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from pwstablenet_amd import synth  # noqa: E402

math = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ngf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
items = int(sys.argv[3]) if len(sys.argv) > 3 else 2
G = 1 << 20
REG = []
_empty, _zeros, _empty_like, _zeros_like = torch.empty, torch.zeros, torch.empty_like, torch.zeros_like
ACTIVE = [False]


def guarded(shape, dtype, device):
    shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
    n = 1
    for s_ in shape:
        n *= int(s_)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    pad = (-nbytes) % 256
    raw = _empty(G + nbytes + pad + G, dtype=torch.uint8, device=device)
    raw[:G] = 0xA5
    raw[G + nbytes:] = 0xA5
    t = raw[G:G + nbytes].view(dtype).view(shape)
    REG.append((raw, nbytes, "".join(traceback.format_stack(limit=6)[:-2])))
    return t


def is_cuda(device):
    return device is not None and torch.device(device).type == "cuda"


def p_empty(*shape, dtype=None, device=None, **kw):
    if ACTIVE[0] and is_cuda(device) and not kw:
        return guarded(shape, dtype or torch.float32, device)
    return _empty(*shape, dtype=dtype, device=device, **kw)


def p_zeros(*shape, dtype=None, device=None, **kw):
    if ACTIVE[0] and is_cuda(device) and not kw:
        return guarded(shape, dtype or torch.float32, device).zero_()
    return _zeros(*shape, dtype=dtype, device=device, **kw)


def p_empty_like(t, **kw):
    if ACTIVE[0] and t.is_cuda and not kw and t.is_contiguous():
        return guarded(t.shape, t.dtype, t.device)
    return _empty_like(t, **kw)


def p_zeros_like(t, **kw):
    if ACTIVE[0] and t.is_cuda and not kw and t.is_contiguous():
        return guarded(t.shape, t.dtype, t.device).zero_()
    return _zeros_like(t, **kw)


torch.empty, torch.zeros, torch.empty_like, torch.zeros_like = p_empty, p_zeros, p_empty_like, p_zeros_like
import test_hip_threads as T  # noqa: E402
from pwstablenet_amd.objective import StabObjective, train_step  # noqa: E402
from pwstablenet_amd.optim import Adam  # noqa: E402

ACTIVE[0] = True
net = T.make_net("W2", 12, ngf=ngf)
net.module.set_math(math)
net.module.deterministic = bool(int(os.environ.get("PROBE_DET", "1")))
batch = [torch.from_numpy(t).cuda() for t in synth.make_train_batch(items, seed=12)]
opt = Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
obj = StabObjective(batchSize=items)
for step in range(2):
    out = train_step(net, opt, batch, obj)
    torch.cuda.synchronize()
with torch.no_grad():   # inference on the same generator too
    x = torch.from_numpy(synth.noise_window(2, 31, 256, seed=3)).cuda()
    net(x, False)
torch.cuda.synchronize()
ACTIVE[0] = False
bad = 0
for raw, nbytes, where in REG:
    front, back = raw[:G], raw[G + nbytes:]
    for name, band in (("in FRONT of", front), ("BEHIND", back)):
        hit = (band != 0xA5).nonzero()
        if hit.numel():
            bad += 1
            first, last = int(hit.min()), int(hit.max())
            print("guard band %s a %d-byte tensor overwritten: %d bytes, band offsets %d .. %d; allocated at\n%s" % (name, nbytes, hit.numel(), first, last, where))
print("%d tensors checked, %d guard bands overwritten [%s, ngf %d, %d items, deterministic %s]" % (len(REG), bad, math, ngf, items, net.module.deterministic))

"""Fused Adam on the HIP kernel, with the call surface the reference uses: ``optim.Adam(netG.parameters(), lr=...,
betas=(beta1, 0.999))``, ``.zero_grad()``, ``.step()`` (reference main_new.py:63,213-216).  No weight decay / amsgrad
(the reference does not use them).  State is fp32 ``exp_avg`` / ``exp_avg_sq`` per parameter, as in torch."""
import ctypes

import torch

from . import hipabi as A


def _bump_versions(tensors):
    """The packed-weight cache of the generator is keyed on the parameters' version counters: the fused kernel writes through raw
    pointers, so the counters are advanced here.  torch 2.10: the private setter takes (list, list) on the host; earlier releases
    expose the same name as (Tensor, int) or not at all -- any failure falls back to a no-op in-place add (three multi-tensor
    launches), which bumps them the documented way."""
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        try:
            setter(tensors, [t._version + 1 for t in tensors])
            return
        except TypeError:
            pass
    torch._foreach_add_(tensors, 0.0)


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}
        self.step_count = 0
        # the moments of parameters that already sit on the device are made NOW (torch makes them on the first step: there they
        # would be carved out of whatever large block the first backward has just freed -- see UnetGenerator._take_train_arena)
        # -- as views of TWO flat zero buffers per device (two fill launches instead of 184)
        by_dev = {}
        for p in self.params:
            if p.is_cuda and p.requires_grad and p.dtype == torch.float32:
                by_dev.setdefault(p.device, []).append(p)
        for dev, ps in by_dev.items():
            sizes = [(p.numel() + 63) // 64 * 64 for p in ps]   # 256-byte aligned views (the kernel loads float4)
            m, v = torch.zeros(sum(sizes), device=dev), torch.zeros(sum(sizes), device=dev)
            off = 0
            for p, sz in zip(ps, sizes):
                self.state[p] = (m[off:off + p.numel()].view_as(p), v[off:off + p.numel()].view_as(p))
                off += sz

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        touched, grads, ms, vs = [], [], [], []
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
            touched.append(p), grads.append(p.grad.contiguous()), ms.append(st[0]), vs.append(st[1])
        if touched:
            A.require_cuda(*touched, *grads)
            n = len(touched)
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
            counts = (ctypes.c_size_t * n)(*[t.numel() for t in touched])
            # all tensors in one launch per 48 tensors (pws_adam_step_multi); the reference's optim.Adam (main_new.py:63,216)
            A.check(A.lib().pws_adam_step_multi(arr(touched), arr(grads), arr(ms), arr(vs), counts, n, self.lr, self.betas[0],
                                                self.betas[1], self.eps, self.step_count, A.current_stream()), "pws_adam_step_multi")
            # the kernel wrote through raw pointers: bump the tensors' version counters so that everything keyed on them -- the
            # generator's packed-weight cache -- sees the update.  On the host (torch._foreach_add_(touched, 0.0) did it with three
            # launches over all parameters: 100 us of a 28 ms step)
            _bump_versions(touched)

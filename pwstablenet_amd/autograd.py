"""Autograd bridge for the generator: ONE ``torch.autograd.Function`` around the whole HIP executor.

torch only records that the six fields depend on the 92 parameters; forward is ``pws_netg_forward`` (is_training=1,
activations kept in an arena private to the call) and backward is ``pws_netg_backward`` (act'/bias grad, MFMA weight
grad, MFMA data grad per layer, head backward) followed by ``pws_netg_unpack_grads`` into torch-layout tensors.
Replaces autograd through ATen conv / conv_transpose / cat of the reference (main_new.py:214).

``netG(x, False)`` with grad mode on (the reference's video loop, main_new.py:697) runs the fast inference forward;
its backward is refused loudly (train with ``netG(x)``).
"""
import ctypes

import torch

from . import hipabi as A


class _NetGTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net_ref, input1, *params):
        net = net_ref[0]
        saved = {}
        grids, resid = net._run(input1, True, train_ctx=saved)
        ctx.net_ref, ctx.saved = net_ref, saved
        return tuple(grids) + tuple(resid)

    @staticmethod
    def backward(ctx, *gouts):
        net, sv = ctx.net_ref[0], ctx.saved
        if sv is None:
            raise RuntimeError("pwstablenet_amd: backward through the generator twice (the activation arena was released)")
        n = sv["x"].shape[0]
        dev = sv["x"].device

        def stack(gs):
            if all(g is None for g in gs):
                return None
            return torch.stack([g.contiguous() if g is not None else torch.zeros((n, 256, 256, 2), device=dev) for g in gs])

        g_grids, g_resid = stack(gouts[0:3]), stack(gouts[3:6])
        if g_grids is None and g_resid is None:
            return (None, None) + (None,) * 92
        L = A.lib()
        st = A.current_stream()
        packed, packed_dg = sv["packed"], net.packed_dgrad_weights()
        dpacked = torch.empty_like(packed)
        net._apply_math(sv.get("math", "fp32"), sv.get("store", "fp32"))  # the arena holds what the forward's mode wrote
        params = net._ordered_params()
        grads = [torch.empty_like(p) for p in params]
        sync = getattr(net, "grad_sync", None)
        if sync is None or sync.nparts == 1:
            A.check(L.pws_netg_backward(A.ptr(packed), A.ptr(packed_dg), A.ptr(sv["x"]), n, net.input_nc, net.ngf, 0,
                                        ctypes.c_void_p(sv["ws_ptr"]), sv["ws_bytes"], A.ptr(sv["resid"]), A.ptr(sv["thetas"]),
                                        A.ptr(g_grids), A.ptr(g_resid), A.ptr(dpacked), st), "pws_netg_backward")
            ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            A.check(L.pws_netg_unpack_grads(A.ptr(dpacked), ptrs, net.input_nc, net.ngf, st), "pws_netg_unpack_grads")
            if sync is not None:
                sync.collectives = 0
                sync.allreduce(grads)
        else:
            # overlapped gradient exchange (distributed.OverlappedGradSync): backward in runs; after each run the layers whose
            # gradients are final are unpacked + all-reduced on the communication stream while the next run computes
            nl = len(params) // 2
            main, comm = torch.cuda.current_stream(dev), sync.stream(dev)
            mask = (ctypes.c_ubyte * nl)()
            done = [False] * nl
            sync.collectives = 0
            for part in range(sync.nparts):
                A.check(L.pws_netg_backward_part(A.ptr(packed), A.ptr(packed_dg), A.ptr(sv["x"]), n, net.input_nc, net.ngf, 0,
                                                 ctypes.c_void_p(sv["ws_ptr"]), sv["ws_bytes"], A.ptr(sv["resid"]),
                                                 A.ptr(sv["thetas"]), A.ptr(g_grids), A.ptr(g_resid), A.ptr(dpacked), part,
                                                 sync.nparts, mask, st), "pws_netg_backward_part")
                newly = [i for i in range(nl) if mask[i] and not done[i]]
                if not newly:
                    continue
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ready)
                    ptrs = (ctypes.c_void_p * len(grads))()
                    for i in newly:
                        ptrs[2 * i], ptrs[2 * i + 1] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
                    A.check(L.pws_netg_unpack_grads(A.ptr(dpacked), ptrs, net.input_nc, net.ngf, A.current_stream()),
                            "pws_netg_unpack_grads")
                    sync.allreduce([grads[k] for i in newly for k in (2 * i, 2 * i + 1)])
                for i in newly:
                    done[i] = True
            assert all(done), "pws_netg_backward_part: a layer never became final"
            main.wait_stream(comm)
        ctx.saved = None  # release the arena
        return (None, None) + tuple(grads)


class _NetGInferNoGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net_ref, input1, *params):
        return net_ref[0]._run(input1, False)

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError("pwstablenet_amd: netG(x, False) is the inference path and keeps no activations; "
                           "call netG(x) (is_training=True) to train")


def netg_apply(net, input1, is_training):
    params = net._ordered_params()
    if input1.requires_grad:
        raise NotImplementedError("pwstablenet_amd: gradient wrt the input window is not provided (the window is data)")
    if not is_training:
        return _NetGInferNoGrad.apply([net], input1, *params)
    out = _NetGTrain.apply([net], input1, *params)
    return list(out[:3]), list(out[3:])

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
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward (handled there), not as materialised zeros
        return tuple(grids) + tuple(resid)

    @staticmethod
    def backward(ctx, *gouts):
        net, sv = ctx.net_ref[0], ctx.saved
        if sv is None:
            raise RuntimeError("pwstablenet_amd: backward through the generator twice (the activation arena was released)")
        n = sv["x"].shape[0]
        dev = sv["x"].device
        # the six upstream gradients go over as two lists of pointers (pws_netg_backward_lists): no stacked copy, and an output
        # nobody differentiated is a NULL entry instead of a zero-filled stand-in
        keep = [None if g is None else g.contiguous() for g in gouts[:6]]
        if all(g is None for g in keep):
            return (None, None) + (None,) * 92
        for g in keep:
            if g is not None:
                A.require_cuda(g)
                if tuple(g.shape) != (n, 256, 256, 2):
                    raise RuntimeError("pwstablenet_amd: upstream gradient of shape %s for an (%d, 256, 256, 2) field" % (tuple(g.shape), n))
        g_grids = (ctypes.c_void_p * 3)(*[None if g is None else g.data_ptr() for g in keep[0:3]])
        g_resid = (ctypes.c_void_p * 3)(*[None if g is None else g.data_ptr() for g in keep[3:6]])
        L = A.lib()
        st = A.current_stream()
        # the packed weights are ONE buffer shared by every forward of this generator and re-packed in place when a parameter
        # changes: a backward that runs after optimizer.step() / load_state_dict() would silently use the new weights (torch
        # raises "modified by an inplace operation" in this situation; so does this)
        if sv["packed"] is not net._packed or net._weights_key() != sv["weights_key"]:
            raise RuntimeError("pwstablenet_amd: one of the generator's parameters was modified (optimizer.step / load_state_dict / "
                               "another forward after a change) between netG(x) and its backward; the packed weights of that "
                               "forward are gone -- call backward() before updating the parameters")
        packed, packed_dg = sv["packed"], net.packed_dgrad_weights()
        # the gradient slab: weight + bias gradients of the 46 layers in the kernels' layout and nothing else (195 MB at ngf 64);
        # a data-parallel job all-reduces it IN PLACE and unpacks once
        dpacked = torch.empty(L.pws_netg_grad_floats(net.input_nc, net.ngf), device=dev, dtype=torch.float32)
        opts = net._opts(sv.get("math", "fp32"), sv.get("store", "fp32"))  # the arena holds what the forward's mode wrote
        ac = sv.get("ac", 0)
        params = net._ordered_params()
        grads = [torch.empty_like(p) for p in params]
        sync = getattr(net, "grad_sync", None)

        def run(part, nparts, mask):
            A.check(L.pws_netg_backward_lists(A.ptr(packed), A.ptr(packed_dg), A.ptr(sv["x"]), n, net.input_nc, net.ngf, ac,
                                              ctypes.c_void_p(sv["ws_ptr"]), sv["ws_bytes"], A.ptr(sv["resid"]), A.ptr(sv["thetas"]),
                                              g_grids, g_resid, A.ptr(dpacked), part, nparts, mask, ctypes.byref(opts), st),
                    "pws_netg_backward_lists")

        if sync is None or sync.nparts == 1:
            run(0, 1, None)
            if sync is not None:
                sync.collectives = 0
                sync.allreduce_slab(dpacked, [(0, dpacked.numel())])
            ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            A.check(L.pws_netg_unpack_grads(A.ptr(dpacked), ptrs, net.input_nc, net.ngf, st), "pws_netg_unpack_grads")
        else:
            # overlapped gradient exchange (distributed.OverlappedGradSync): backward in runs; after each run the slab ranges of the
            # layers whose gradients are final are all-reduced in place and unpacked on the communication stream while the next
            # run computes
            nl = len(params) // 2
            first, count = net._grad_layout()
            main, comm = torch.cuda.current_stream(dev), sync.stream(dev)
            mask = (ctypes.c_ubyte * nl)()
            done = [False] * nl
            sync.collectives = 0
            sync.final_part = [None] * nl   # the run after which the C side reported each layer final (tests: equals pws_netg_backward_plan)
            for part in range(sync.nparts):
                run(part, sync.nparts, mask)
                newly = [i for i in range(nl) if mask[i] and not done[i]]
                if not newly:
                    continue
                ranges = []   # consecutive layers abut in the slab: merge them into a few large messages
                for i in newly:
                    if ranges and ranges[-1][1] == first[i]:
                        ranges[-1][1] = first[i] + count[i]
                    else:
                        ranges.append([first[i], first[i] + count[i]])
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(ready)
                    sync.allreduce_slab(dpacked, ranges)
                    ptrs = (ctypes.c_void_p * len(grads))()
                    for i in newly:
                        ptrs[2 * i], ptrs[2 * i + 1] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
                    A.check(L.pws_netg_unpack_grads(A.ptr(dpacked), ptrs, net.input_nc, net.ngf, A.current_stream()),
                            "pws_netg_unpack_grads")
                for i in newly:
                    done[i] = True
                    sync.final_part[i] = part
            assert all(done), "pws_netg_backward_part: a layer never became final"
            main.wait_stream(comm)
            dpacked.record_stream(comm)
            for g in grads:
                g.record_stream(comm)
        ctx.saved = None  # release the arena: back to the generator for the next forward on this stream (UnetGenerator._take_train_arena)
        ws = sv.pop("ws", None)
        del sv
        net._give_train_arena(ws)
        return (None, None) + tuple(grads)


class _NetGInferNoGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net_ref, input1, *params):
        return net_ref[0]._run(input1, False)

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError("pwstablenet_amd: netG(x, False) is the inference path and keeps no activations; "
                           "call netG(x) (is_training=True) to train")


# calls of each module per forward of the reference (lib/networks_cascading.py:152-237): nn.BatchNorm2d counts them
def _bn_uses(net):
    uses = []
    for ls in net._specs:
        block = ls.name.split(".")[0]
        if block in ("out", "flatten", "linear"):
            uses.append(3)
        elif block.startswith(("down_bottom", "up_bottom")):
            uses.append(2)
        else:
            uses.append(1)
    return uses


class _NetGTrainBN(torch.autograd.Function):
    """use_BN=True in train() mode: the whole forward / backward through pws_netg_forward_bn / pws_netg_backward_bn (BatchNorm2d
    with batch statistics after every conv; running statistics updated in place as nn.BatchNorm2d does)."""

    @staticmethod
    def forward(ctx, net_ref, input1, *params):
        net = net_ref[0]
        A.require_cuda(input1)
        L, st = A.lib(), A.current_stream()
        x = input1.contiguous()
        n, dev = x.shape[0], x.device
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training (use_BN=True: the theta head normalises over "
                             "the batch), got batch size %d" % n)
        bns = net._bn_layers()
        packed = net.packed_weights(raw=True)
        with torch.no_grad():
            bn_params = torch.cat([t.detach().reshape(-1) for bn in bns for t in (bn.weight, bn.bias)]).contiguous()
            running = torch.cat([t.reshape(-1) for bn in bns for t in (bn.running_mean, bn.running_var)]).contiguous()
        assert bn_params.numel() == L.pws_netg_bn_floats(net.input_nc, net.ngf)
        nbytes = L.pws_netg_train_workspace_bytes_bn(n, net.input_nc, net.ngf)
        ws = torch.empty(nbytes + 256, device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        ws_bytes = ws.numel() - (ws_ptr - ws.data_ptr())
        grids = torch.empty((3, n, 256, 256, 2), device=dev, dtype=torch.float32)
        resid = torch.empty((3, n, 256, 256, 2), device=dev, dtype=torch.float32)
        thetas = torch.empty((3, n, 6), device=dev, dtype=torch.float32)
        eps, mom = float(bns[0].eps), float(bns[0].momentum if bns[0].momentum is not None else 0.1)
        # the BatchNorm path stores fp32 activations whatever set_math's storage says; its conv contractions follow net.math
        opts = net._opts(net.math, "fp32")
        A.check(L.pws_netg_forward_bn_opts(A.ptr(packed), A.ptr(bn_params), A.ptr(running), mom, eps, A.ptr(x), n, net.input_nc, net.ngf,
                                           int(net.align_corners), ctypes.c_void_p(ws_ptr), ws_bytes, A.ptr(grids), A.ptr(resid), A.ptr(thetas),
                                           ctypes.byref(opts), st), "pws_netg_forward_bn_opts")
        with torch.no_grad():   # running statistics back into the modules (plumbing), call counters as nn.BatchNorm2d keeps them
            off = 0
            for bn, uses in zip(bns, _bn_uses(net)):
                c = bn.num_features
                bn.running_mean.copy_(running[off:off + c])
                bn.running_var.copy_(running[off + c:off + 2 * c])
                bn.num_batches_tracked += uses
                off += 2 * c
        net.last_thetas = thetas
        ctx.net_ref = net_ref
        ctx.saved = dict(x=x, ws=ws, ws_ptr=ws_ptr, ws_bytes=ws_bytes, resid=resid, thetas=thetas, packed=packed, bn_params=bn_params,
                         eps=eps, ac=int(net.align_corners), math=net.math)
        return (grids[0], grids[1], grids[2], resid[0], resid[1], resid[2])

    @staticmethod
    def backward(ctx, *gouts):
        net, sv = ctx.net_ref[0], ctx.saved
        if sv is None:
            raise RuntimeError("pwstablenet_amd: backward through the generator twice (the activation arena was released)")
        n, dev = sv["x"].shape[0], sv["x"].device

        def stack(gs):
            if all(g is None for g in gs):
                return None
            return torch.stack([g.contiguous() if g is not None else torch.zeros((n, 256, 256, 2), device=dev) for g in gs])
        g_grids, g_resid = stack(gouts[0:3]), stack(gouts[3:6])
        nconv = len(net._ordered_params())
        if g_grids is None and g_resid is None:
            return (None, None) + (None,) * (2 * nconv)
        L, st = A.lib(), A.current_stream()
        packed, packed_dg = sv["packed"], net.packed_dgrad_weights()
        dpacked = torch.empty(L.pws_netg_grad_floats(net.input_nc, net.ngf), device=dev, dtype=torch.float32)
        dbn = torch.empty_like(sv["bn_params"])
        opts = net._opts(sv["math"], "fp32")   # the arena holds what the forward's mode wrote
        A.check(L.pws_netg_backward_bn_opts(A.ptr(packed), A.ptr(packed_dg), A.ptr(sv["bn_params"]), sv["eps"], A.ptr(sv["x"]), n, net.input_nc,
                                            net.ngf, sv["ac"], ctypes.c_void_p(sv["ws_ptr"]), sv["ws_bytes"], A.ptr(sv["resid"]),
                                            A.ptr(sv["thetas"]), A.ptr(g_grids), A.ptr(g_resid), A.ptr(dpacked), A.ptr(dbn), ctypes.byref(opts), st),
                "pws_netg_backward_bn_opts")
        params = net._ordered_params()
        grads = [torch.empty_like(p) for p in params]
        ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        A.check(L.pws_netg_unpack_grads(A.ptr(dpacked), ptrs, net.input_nc, net.ngf, st), "pws_netg_unpack_grads")
        bn_grads, off = [], 0
        for bn in net._bn_layers():
            c = bn.num_features
            bn_grads += [dbn[off:off + c].clone(), dbn[off + c:off + 2 * c].clone()]
            off += 2 * c
        ctx.saved = None
        return (None, None) + tuple(grads) + tuple(bn_grads)


def netg_apply_bn(net, input1):
    if input1.requires_grad:
        raise NotImplementedError("pwstablenet_amd: gradient wrt the input window is not provided (the window is data)")
    params = net._ordered_params() + [t for bn in net._bn_layers() for t in (bn.weight, bn.bias)]
    out = _NetGTrainBN.apply([net], input1, *params)
    return list(out[:3]), list(out[3:])


def netg_apply(net, input1, is_training):
    params = net._ordered_params()
    if input1.requires_grad:
        raise NotImplementedError("pwstablenet_amd: gradient wrt the input window is not provided (the window is data)")
    if not is_training:
        return _NetGInferNoGrad.apply([net], input1, *params)
    out = _NetGTrain.apply([net], input1, *params)
    return list(out[:3]), list(out[3:])

"""Autograd bridge for the generator: one ``torch.autograd.Function`` around the whole HIP executor.

torch only records that the fields depend on the 92 parameters (and on the input window); the forward is
``pws_netg_forward`` and the backward will be the HIP data-/weight-gradient kernels.
Until those exist the backward raises -- loudly, never a silent CPU path -- while forward-only use with grad
mode on (the reference's video loop calls ``netG(images, False)`` without ``no_grad``, main_new.py:697) works.
"""
import torch


class _NetG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net_ref, is_training, input1, *params):
        net = net_ref[0]
        out = net._run(input1, is_training)
        ctx.net_ref = net_ref
        ctx.is_training = is_training
        if is_training:
            grids, resid = out
            return tuple(grids) + tuple(resid)
        return out

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError(
            "pwstablenet_amd: backward through the generator (conv dgrad/wgrad HIP kernels) is not implemented yet; "
            "wrap inference in torch.no_grad() or detach the fields")


def netg_apply(net, input1, is_training):
    params = net._ordered_params()
    out = _NetG.apply([net], bool(is_training), input1, *params)
    if is_training:
        return list(out[:3]), list(out[3:])
    return out

"""VGG-16 perceptual term of the reference's generator objective (lib/utils.py:11-32 ``GeneratorLoss``:
``MSELoss(vgg16.features[:31](out_images), vgg16.features[:31](target_images))``, called per stage and per branch at
main_new.py:191-192) on this library's kernels: the 13 conv3x3 + ReLU layers run on ``pws_conv2d_fwd`` /
``pws_conv2d_bwd_data`` (Winograd F(2x2,3x3) where the map is large enough, optionally the bf16 matrix cores), the five
``MaxPool2d(2, 2)`` and the MSE on ``csrc/pool.hip``.  The VGG weights are frozen (as in the reference), so the backward is
data gradients only.  ``math='bf16'`` also keeps the activations and their gradients in bf16 (NHWC, RGB zero-padded to 32
channels): half the HBM traffic of every layer and 16-byte loads / stores in the conv kernels (64 images, 2 forwards + 1
backward: 19.2 ms with fp32 activations -> 11.9 ms); the features are converted to fp32 for the MSE.

The module tree and parameter names are torchvision's (``features.0.weight`` ... ``features.28.bias``), so
``load_state_dict(torchvision_vgg16_state_dict, strict=False)`` takes the pretrained weights where they are available;
torchvision and its weights are NOT in this image (no network), so tests use seeded random weights and compare with the same
stack built from ``torch.nn.functional`` calls on the CPU (oracle/objective_ref.py).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import hipabi as A

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")   # features[:31]
_CIN_PAD = 16   # the conv kernels take sources of a multiple of 16 channels: the RGB input is zero-padded


class _VGGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x):
        A.require_cuda(x)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise RuntimeError("VGG16Features: expected (N, 3, H, W) with H, W multiples of 32, got %s" % (tuple(x.shape),))
        L, st = A.lib(), A.current_stream()
        n, _, h, w = x.shape
        x = x.contiguous()
        # bf16 math stores the activations as bf16 too (half the HBM traffic of every layer, 16-byte loads and stores in the
        # conv kernels); the RGB input is then zero-padded to the 32 channels the bf16 conv kernels take per chunk
        s16 = net.math == "bf16"
        store, dt, cpad = (A.STORE_BF16, torch.bfloat16, 32) if s16 else (A.STORE_FP32, torch.float32, _CIN_PAD)
        cur = torch.empty((n, h, w, cpad), device=x.device, dtype=dt)
        A.check(L.pws_nchw_to_nhwc_pad_s(A.ptr(x), A.ptr(cur), n, 3, h, w, cpad, store, st), "pws_nchw_to_nhwc_pad_s")
        packs = net._packed_weights(net.math)
        acts = []   # (kind, input tensor, output tensor)
        ci = 0
        for v in VGG16_CFG:
            if v == "M":
                out = torch.empty((n, h // 2, w // 2, cur.shape[3]), device=x.device, dtype=dt)
                A.check(L.pws_maxpool2x2_fwd_s(A.ptr(cur), A.ptr(out), n, h, w, cur.shape[3], store, st), "pws_maxpool2x2_fwd_s")
                acts.append(("M", cur, out))
                h, w = h // 2, w // 2
            else:
                pk = packs[ci]
                out = torch.empty((n, h, w, v), device=x.device, dtype=dt)
                a = A.PwsConvArgs()
                a.kind, a.n, a.h, a.w, a.nsrc = A.CONV_K3S1, n, h, w, 1
                a.src[0].ptr, a.src[0].channels, a.src[0].ld = cur.data_ptr(), cur.shape[3], cur.shape[3]
                a.cout, a.w_packed, a.bias, a.act = v, pk["w"].data_ptr(), pk["b"].data_ptr(), A.ACT_RELU
                a.out, a.out_ld = out.data_ptr(), v
                a.w_wino = pk["wino"].data_ptr()
                if pk["wring"] is not None:
                    a.w_wring = pk["wring"].data_ptr()
                if s16:
                    a.math, a.w_bf16, a.store = A.MATH_BF16, pk["bf16"].data_ptr(), A.STORE_BF16
                A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "pws_conv2d_fwd")
                acts.append((ci, cur, out))
                ci += 1
            cur = out
        ctx.net, ctx.acts, ctx.in_shape, ctx.s16, ctx.math = net, acts, tuple(x.shape), s16, net.math
        if s16:   # the MSE runs on fp32 features (n x H/32 x W/32 x 512: small)
            feat = torch.empty(cur.shape, device=x.device, dtype=torch.float32)
            A.check(L.pws_cvt_bf16_to_f32(A.ptr(cur), A.ptr(feat), cur.numel(), st), "pws_cvt_bf16_to_f32")
            return feat
        return cur

    @staticmethod
    def backward(ctx, g):
        net, acts, s16 = ctx.net, ctx.acts, ctx.s16
        if acts is None:
            raise RuntimeError("VGG16Features: backward twice (the activations were released)")
        L, st = A.lib(), A.current_stream()
        n = ctx.in_shape[0]
        packs = net._packed_weights(ctx.math)   # the arithmetic of the forward governs its backward
        store = A.STORE_BF16 if s16 else A.STORE_FP32
        g = g.contiguous()
        if s16:
            g16 = torch.empty(g.shape, device=g.device, dtype=torch.bfloat16)
            A.check(L.pws_cvt_f32_to_bf16(A.ptr(g), A.ptr(g16), g.numel(), 0, st), "pws_cvt_f32_to_bf16")
            g = g16
        else:
            g = g.clone()   # modified in place below
        # bf16 storage: ReLU' of a layer's output is applied by whatever PRODUCES the gradient of that output -- the next conv's
        # data-gradient epilogue (pws_dst.act_y) or the max-pool backward (relu_mask) -- so no separate pass over the gradient;
        # only the last conv's output gradient (it comes from the caller) takes the elementwise pass
        masked = False   # g already is the gradient wrt the pre-activation of the tensor it belongs to
        ra = list(reversed(acts))
        for idx, (kind, xin, out) in enumerate(ra):
            h, w = xin.shape[1], xin.shape[2]
            prev_is_conv = idx + 1 < len(ra) and ra[idx + 1][0] != "M"   # xin is a conv + ReLU output
            if kind == "M":
                dx = torch.empty_like(xin)
                fuse = s16 and prev_is_conv
                if s16:
                    A.check(L.pws_maxpool2x2_bwd_s(A.ptr(xin), A.ptr(g), A.ptr(dx), n, h, w, xin.shape[3], store, 1 if fuse else 0, st),
                            "pws_maxpool2x2_bwd_s")
                else:
                    A.check(L.pws_maxpool2x2_bwd(A.ptr(xin), A.ptr(g), A.ptr(dx), n, h, w, xin.shape[3], st), "pws_maxpool2x2_bwd")
                masked = fuse
            else:
                pk = packs[kind]
                cout = out.shape[3]
                if not masked:
                    A.check(L.pws_act_bwd_bias_s(A.ptr(g), A.ptr(out), n * h * w, cout, A.ACT_RELU, None, store, None, 0, st),
                            "pws_act_bwd_bias_s")
                dx = torch.empty_like(xin)
                d = A.PwsConvBwdDataArgs()
                d.kind, d.n, d.h, d.w, d.cout = A.CONV_K3S1, n, h, w, cout
                d.gout, d.gout_ld, d.w_dgrad, d.ndst = g.data_ptr(), cout, pk["dg"].data_ptr(), 1
                d.dst[0].ptr, d.dst[0].channels, d.dst[0].ld, d.dst[0].accumulate = dx.data_ptr(), xin.shape[3], xin.shape[3], 0
                fuse = s16 and prev_is_conv
                if fuse:
                    d.dst[0].act_y, d.dst[0].act_y_ld, d.dst[0].act = xin.data_ptr(), xin.shape[3], A.ACT_RELU
                if s16:
                    d.math, d.w_dgrad_bf16, d.store = A.MATH_BF16, pk["dg_bf16"].data_ptr(), store
                A.check(L.pws_conv2d_bwd_data(ctypes.byref(d), st), "pws_conv2d_bwd_data")
                masked = fuse
            g = dx
        ctx.acts = None
        if s16:
            g32 = torch.empty(g.shape, device=g.device, dtype=torch.float32)
            A.check(L.pws_cvt_bf16_to_f32(A.ptr(g), A.ptr(g32), g.numel(), st), "pws_cvt_bf16_to_f32")
            g = g32
        return None, g[..., :3].permute(0, 3, 1, 2).contiguous()   # padded NHWC -> the NCHW RGB gradient (layout plumbing)


class VGG16Features(nn.Module):
    """``nn.Sequential(*list(vgg16().features)[:31])`` with frozen parameters; ``forward`` returns the NHWC feature map
    (N, H/32, W/32, 512) -- only its element-wise MSE is ever used, so the layout is immaterial."""

    def __init__(self, math="fp32"):
        super().__init__()
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        for p in self.parameters():
            p.requires_grad = False
        if math not in ("fp32", "bf16"):
            raise ValueError("VGG16Features: math must be 'fp32' or 'bf16'")
        self.math = math
        self._packs, self._key = None, None

    def init_random(self, seed=0):
        """He-normal weights from numpy's legacy RandomState (bit-identical wherever it is regenerated) -- a stand-in for the
        pretrained weights, which cannot be fetched here."""
        rs = np.random.RandomState(seed)
        with torch.no_grad():
            for m in self.features:
                if isinstance(m, nn.Conv2d):
                    fan_in = m.in_channels * 9
                    m.weight.copy_(torch.from_numpy((rs.standard_normal(tuple(m.weight.shape)) * np.sqrt(2.0 / fan_in)).astype(np.float32)))
                    m.bias.copy_(torch.from_numpy((rs.standard_normal(m.out_channels) * 0.05).astype(np.float32)))
        return self

    def _convs(self):
        return [m for m in self.features if isinstance(m, nn.Conv2d)]

    def _packed_weights(self, math=None):
        math = math or self.math
        convs = self._convs()
        key = (math,) + tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in convs)
        if self._packs is not None and key == self._key:
            return self._packs
        L, st = A.lib(), A.current_stream()
        packs = []
        for m in convs:
            A.require_cuda(m.weight, m.bias)
            cin, cout = m.in_channels, m.out_channels
            w = m.weight.detach()
            cpad = 32 if math == "bf16" else _CIN_PAD
            if cin % cpad:
                w = torch.cat([w, torch.zeros((cout, cpad - cin, 3, 3), device=w.device)], 1).contiguous()   # zero taps for the padding channels
                cin = cpad
            dev = w.device
            wp = torch.empty(L.pws_packed_weight_floats(A.CONV_K3S1, cin, cout), device=dev, dtype=torch.float32)
            A.check(L.pws_pack_conv_weight(A.ptr(w), A.ptr(wp), A.CONV_K3S1, cin, cout, st), "pws_pack_conv_weight")
            ww = torch.empty(L.pws_packed_wino_floats(cin, cout), device=dev, dtype=torch.float32)
            A.check(L.pws_pack_conv_weight_wino(A.ptr(wp), A.ptr(ww), cin, cout, st), "pws_pack_conv_weight_wino")
            wr = None
            if math != "bf16" and L.pws_packed_wring_floats(A.CONV_K3S1, cin, cout):   # LDS-ring Winograd (fp32 path, cout % 32 == 0)
                wr = torch.empty(L.pws_packed_wring_floats(A.CONV_K3S1, cin, cout), device=dev, dtype=torch.float32)
                A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), A.CONV_K3S1, cin, cout, st), "pws_pack_conv_weight_wring")
            dg = torch.empty(L.pws_packed_dgrad_floats(A.CONV_K3S1, cin, cout), device=dev, dtype=torch.float32)
            A.check(L.pws_pack_conv_weight_dgrad(A.ptr(w), A.ptr(dg), A.CONV_K3S1, cin, cout, st), "pws_pack_conv_weight_dgrad")
            wb = dgb = None
            if math == "bf16" and cin % 32 == 0:
                wb = torch.empty(L.pws_packed_bf16_floats(9, cin, cout), device=dev, dtype=torch.float32)
                A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), 9, cin, cout, st), "pws_pack_weight_bf16")
            if math == "bf16" and cout % 32 == 0:   # the gradient's contraction runs over cout; cin is padded to 64 by the pack
                dgb = torch.empty(L.pws_packed_bf16_floats(9, cout, cin), device=dev, dtype=torch.float32)
                A.check(L.pws_pack_weight_bf16(A.ptr(dg), A.ptr(dgb), 9, cout, cin, st), "pws_pack_weight_bf16")
            packs.append({"w": wp, "b": m.bias.detach().contiguous(), "wino": ww, "wring": wr, "dg": dg, "bf16": wb, "dg_bf16": dgb})
        self._packs, self._key = packs, key
        return packs

    def forward(self, x):
        return _VGGFn.apply(self, x)


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        A.require_cuda(a, b)
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape or a.numel() % 4:
            raise RuntimeError("mse: shapes %s / %s" % (tuple(a.shape), tuple(b.shape)))
        L, st = A.lib(), A.current_stream()
        slots = torch.zeros(A.OBJ_SLOTS, device=a.device, dtype=torch.float64)
        A.check(L.pws_sqdiff_sum(A.ptr(a), A.ptr(b), a.numel(), A.ptr(slots), st), "pws_sqdiff_sum")
        coef = torch.tensor([1.0 / a.numel()], device=a.device, dtype=torch.float64)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        A.check(L.pws_objective_finalize(A.ptr(slots), 1, A.ptr(coef), 1, A.ptr(out), st), "pws_objective_finalize")
        ctx.save_for_backward(a, b)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        scale = g.reshape(1).to(torch.float32).contiguous()
        A.check(A.lib().pws_sqdiff_bwd(A.ptr(a), A.ptr(b), a.numel(), 1.0 / a.numel(), A.ptr(scale), A.ptr(ga), A.current_stream()),
                "pws_sqdiff_bwd")
        return ga, None   # the target is data


def mse_loss(a, b):
    """``nn.MSELoss()(a, b)`` with gradient wrt ``a`` only."""
    return _MSE.apply(a, b.detach())


class GeneratorLoss(nn.Module):
    """Same name and call as lib/utils.py:11-32: ``GeneratorLoss()(out_images, target_images)`` -> perceptual loss."""

    def __init__(self, vgg=None, math="fp32"):
        super().__init__()
        self.loss_network = vgg if vgg is not None else VGG16Features(math)

    def forward(self, out_images, target_images):
        with torch.no_grad():
            ft = self.loss_network(target_images.contiguous())
        return mse_loss(self.loss_network(out_images), ft)


def perceptual_term(generator_criterion):
    """The ``loss_vgg`` sum of main_new.py:185-192 as a ``train_step(perceptual=...)`` hook on the batched layout
    (m = 2n samples, branch 1 first): per stage the reference adds MSE over branch 1 and MSE over branch 2, which is twice
    the MSE over the batched 2n samples; the stable frames' features are computed once instead of once per stage."""
    net = generator_criterion.loss_network

    def term(fakes, stable_rgb):
        with torch.no_grad():
            ft = net(stable_rgb.contiguous())
        return sum(2.0 * mse_loss(net(f), ft) for f in fakes)
    return term

"""Drop-in replacements for the torch functions the reference drivers call on the hot path
(reference main_new.py:106-118,195,197,706-716), backed by the HIP kernels through the C ABI.

Same names, argument meaning and defaults as ``torch.nn.functional``:
    grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=None)
    affine_grid(theta, size, align_corners=None)
plus ``upsample_bilinear2d(x, size)`` (== ``torch.nn.UpsamplingBilinear2d(size=...)``) and the fused
``upsample_grid_sample(input, field)`` of the 720p path.  torch is used for device memory, streams and
autograd bookkeeping only.
"""
import torch

from . import hipabi as A


def _ac(align_corners):
    # torch >= 1.3 semantics, which is what the goldens were generated with: None -> False
    return 1 if align_corners else 0


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, align_corners):
        A.require_cuda(input, grid)
        if input.dim() != 4 or grid.dim() != 4 or grid.shape[-1] != 2 or grid.shape[0] != input.shape[0]:
            raise RuntimeError("grid_sample: expected input (N,C,H,W) and grid (N,Ho,Wo,2), got %s and %s"
                               % (tuple(input.shape), tuple(grid.shape)))
        input, grid = input.contiguous(), grid.contiguous()
        n, c, h, w = input.shape
        ho, wo = grid.shape[1], grid.shape[2]
        out = torch.empty((n, c, ho, wo), device=input.device, dtype=torch.float32)
        A.check(A.lib().pws_grid_sample_fwd(A.ptr(input), A.ptr(grid), A.ptr(out), n, c, h, w, ho, wo, align_corners,
                                            A.current_stream()), "pws_grid_sample_fwd")
        ctx.save_for_backward(input, grid)
        ctx.align_corners = align_corners
        return out

    @staticmethod
    def backward(ctx, gout):
        input, grid = ctx.saved_tensors
        gout = gout.contiguous()
        n, c, h, w = input.shape
        ho, wo = grid.shape[1], grid.shape[2]
        gi = torch.empty_like(input) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        A.check(A.lib().pws_grid_sample_bwd(A.ptr(gout), A.ptr(input), A.ptr(grid), A.ptr(gi), A.ptr(gg), n, c, h, w, ho,
                                            wo, ctx.align_corners, A.current_stream()), "pws_grid_sample_bwd")
        return gi, gg, None


def grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
    if mode != "bilinear" or padding_mode != "zeros":
        raise NotImplementedError("grid_sample: only mode='bilinear', padding_mode='zeros' (what the reference uses)")
    return _GridSample.apply(input, grid, _ac(align_corners))


class _AffineGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, n, h, w, align_corners):
        A.require_cuda(theta)
        theta = theta.contiguous().view(n, 6)
        grid = torch.empty((n, h, w, 2), device=theta.device, dtype=torch.float32)
        A.check(A.lib().pws_affine_grid(A.ptr(theta), A.ptr(grid), n, h, w, align_corners, A.current_stream()), "pws_affine_grid")
        ctx.dims = (n, h, w, align_corners)
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        n, h, w, ac = ctx.dims
        ggrid = ggrid.contiguous()
        gtheta = torch.empty((n, 2, 3), device=ggrid.device, dtype=torch.float32)
        A.check(A.lib().pws_affine_grid_bwd(A.ptr(ggrid), A.ptr(gtheta), n, h, w, ac, A.current_stream()), "pws_affine_grid_bwd")
        return gtheta, None, None, None, None


def affine_grid(theta, size, align_corners=None):
    """F.affine_grid(theta (N,2,3), size (N,C,H,W)) -> (N,H,W,2) (reference lib/networks_cascading.py:164, main_new.py:195)."""
    A.require_cuda(theta)
    if len(size) != 4:
        raise NotImplementedError("affine_grid: 2-D grids only (size = (N, C, H, W))")
    n, h, w = int(size[0]), int(size[-2]), int(size[-1])
    if theta.numel() != n * 6:
        raise RuntimeError("affine_grid: expected theta of shape (%d, 2, 3), got %s" % (n, tuple(theta.shape)))
    return _AffineGrid.apply(theta, n, h, w, _ac(align_corners))


class _UpsampleBilinearAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        A.require_cuda(x)
        x = x.contiguous()
        n, c, h, w = x.shape
        out = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32)
        A.check(A.lib().pws_upsample_bilinear_ac(A.ptr(x), A.ptr(out), n, c, h, w, ho, wo, A.current_stream()),
                "pws_upsample_bilinear_ac")
        ctx.dims = (n, c, h, w, ho, wo)
        return out

    @staticmethod
    def backward(ctx, gout):
        n, c, h, w, ho, wo = ctx.dims
        gout = gout.contiguous()
        gin = torch.empty((n, c, h, w), device=gout.device, dtype=torch.float32)
        A.check(A.lib().pws_upsample_bilinear_ac_bwd(A.ptr(gout), A.ptr(gin), n, c, h, w, ho, wo, A.current_stream()),
                "pws_upsample_bilinear_ac_bwd")
        return gin, None, None


def upsample_bilinear2d(x, size):
    """torch.nn.UpsamplingBilinear2d(size=size)(x): bilinear, align_corners=True (reference main_new.py:708)."""
    if x.dim() != 4:
        raise RuntimeError("upsample_bilinear2d: expected (N,C,H,W), got %s" % (tuple(x.shape),))
    if isinstance(size, int):
        size = (size, size)
    return _UpsampleBilinearAC.apply(x, int(size[0]), int(size[1]))


class UpsamplingBilinear2d(torch.nn.Module):
    def __init__(self, size=None, scale_factor=None):
        super().__init__()
        if size is None:
            raise NotImplementedError("UpsamplingBilinear2d: only size=(H, W) (what the reference uses)")
        self.size = size

    def forward(self, x):
        return upsample_bilinear2d(x, self.size)


def upsample_grid_sample(input, field, align_corners=None):
    """grid_sample(input, UpsamplingBilinear2d(size=input.shape[-2:])(field)) without materialising the resized field.

    input: (N,C,H,W) frame; field: (N,h,w,2) warp field as returned by ``netG(x, False)``.
    """
    A.require_cuda(input, field)
    input, field = input.contiguous(), field.contiguous()
    n, c, h, w = input.shape
    fh, fw = field.shape[1], field.shape[2]
    out = torch.empty_like(input)
    A.check(A.lib().pws_upsample_grid_sample_fwd(A.ptr(input), A.ptr(field), A.ptr(out), n, c, h, w, fh, fw,
                                                 _ac(align_corners), A.current_stream()), "pws_upsample_grid_sample_fwd")
    return out


def upsample_grid_sample_u8(frames_hwc, field, swap_rb=False, align_corners=None, out=None):
    """The 720p step of the reference's ``process()`` on the frames as OpenCV delivers them (main_new.py:679-721):
    ``frames_hwc`` (N,H,W,3) uint8 -> [optional BGR<->RGB swap] -> float CHW -> grid_sample with the field resized to (H,W)
    (UpsamplingBilinear2d) -> ``astype(uint8)`` HWC, in one kernel and 6 bytes of frame traffic per pixel.  W % 4 == 0."""
    A.require_cuda(field)
    A.require_cuda(frames_hwc, dtype=torch.uint8)
    if frames_hwc.dim() != 4 or frames_hwc.shape[3] != 3:
        raise ValueError("upsample_grid_sample_u8: frames must be uint8 (N,H,W,3), got %s %s" % (frames_hwc.dtype, tuple(frames_hwc.shape)))
    frames_hwc, field = frames_hwc.contiguous(), field.contiguous()
    n, h, w, _ = frames_hwc.shape
    if out is None:
        out = torch.empty_like(frames_hwc)
    A.check(A.lib().pws_upsample_grid_sample_u8(A.ptr(frames_hwc), A.ptr(field), A.ptr(out), n, h, w, field.shape[1], field.shape[2],
                                                int(bool(swap_rb)), _ac(align_corners), A.current_stream()),
            "pws_upsample_grid_sample_u8")
    return out


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step):
    """In-place fused Adam on flat fp32 CUDA buffers."""
    A.require_cuda(p, g, m, v)
    A.check(A.lib().pws_adam_step(A.ptr(p), A.ptr(g), A.ptr(m), A.ptr(v), p.numel(), lr, beta1, beta2, eps, step,
                                  A.current_stream()), "pws_adam_step")

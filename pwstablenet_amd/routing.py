"""Routes the torch entry points the reference's drivers call DIRECTLY on the hot path to the HIP kernels, so that an
unchanged ``main_new.py`` reaches them (reference main_new.py:9 ``import torch.nn.functional as functional``;
``functional.grid_sample`` :106,109,116,118,197,716; ``functional.affine_grid`` :195; ``torch.nn.UpsamplingBilinear2d`` :708).

``install()`` (called when ``dropin/lib/networks_cascading.py`` is imported -- the driver's first import of the path,
main_new.py:5) replaces three module
attributes: ``torch.nn.functional.grid_sample``, ``torch.nn.functional.affine_grid`` and ``torch.nn.UpsamplingBilinear2d``.
A call is routed when it is what the kernels implement -- float32 device tensors, 4-D, bilinear + zeros padding -- and handed
to the original torch function UNTOUCHED otherwise (CPU tensors, other dtypes / modes, 5-D volumes): torch code elsewhere in
the process keeps working, and nothing on a device tensor silently changes meaning.  This is routing, not a fallback: a routed
call that fails in the HIP library raises.  ``uninstall()`` restores torch's own attributes.  ``stats`` counts routed /
passed-through calls per entry point (the tests also check the kernels' own launch records, pws_prof_*).
"""
import torch
import torch.nn
import torch.nn.functional as F

from . import functional as PF

_orig = {}
stats = {"grid_sample": [0, 0], "affine_grid": [0, 0], "upsample": [0, 0]}   # name -> [routed, passed through]


def _dev_f32(t):
    return isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32


def _grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
    if (_dev_f32(input) and _dev_f32(grid) and input.dim() == 4 and grid.dim() == 4 and mode == "bilinear"
            and padding_mode == "zeros" and input.device == grid.device):
        stats["grid_sample"][0] += 1
        with torch.cuda.device(input.device):
            return PF.grid_sample(input, grid, mode, padding_mode, align_corners)
    stats["grid_sample"][1] += 1
    return _orig["grid_sample"](input, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def _affine_grid(theta, size, align_corners=None):
    if _dev_f32(theta) and len(size) == 4 and theta.dim() == 3 and tuple(theta.shape[1:]) == (2, 3):
        stats["affine_grid"][0] += 1
        with torch.cuda.device(theta.device):
            return PF.affine_grid(theta, size, align_corners)
    stats["affine_grid"][1] += 1
    return _orig["affine_grid"](theta, size, align_corners=align_corners)


def _make_upsampling_class(base):
    class UpsamplingBilinear2d(base):
        __doc__ = base.__doc__

        def forward(self, input):
            if _dev_f32(input) and input.dim() == 4 and self.size is not None:
                stats["upsample"][0] += 1
                with torch.cuda.device(input.device):
                    return PF.upsample_bilinear2d(input, self.size)
            stats["upsample"][1] += 1
            return super().forward(input)

    UpsamplingBilinear2d.__module__ = base.__module__
    UpsamplingBilinear2d.__qualname__ = base.__qualname__
    return UpsamplingBilinear2d


def installed():
    return bool(_orig)


def install():
    """Idempotent."""
    if _orig:
        return
    _orig.update(grid_sample=F.grid_sample, affine_grid=F.affine_grid, upsample=torch.nn.UpsamplingBilinear2d)
    F.grid_sample = _grid_sample
    F.affine_grid = _affine_grid
    cls = _make_upsampling_class(_orig["upsample"])
    torch.nn.UpsamplingBilinear2d = cls
    torch.nn.modules.UpsamplingBilinear2d = cls
    torch.nn.modules.upsampling.UpsamplingBilinear2d = cls


def uninstall():
    if not _orig:
        return
    F.grid_sample, F.affine_grid = _orig["grid_sample"], _orig["affine_grid"]
    torch.nn.UpsamplingBilinear2d = _orig["upsample"]
    torch.nn.modules.UpsamplingBilinear2d = _orig["upsample"]
    torch.nn.modules.upsampling.UpsamplingBilinear2d = _orig["upsample"]
    _orig.clear()

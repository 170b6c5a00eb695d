"""CPU oracle for the PWStableNet hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``pwstablenet_amd``) never does, and fails loudly when its HIP library is absent.
"""
from .oracle import (  # noqa: F401
    ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, PROBE_NAMES, adam_step, affine_grid, build, conv2d,
    conv_transpose2d, grid_sample_bwd, grid_sample_fwd, lib, netg_forward, upsample_bilinear_ac,
)

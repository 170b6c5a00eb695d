"""pwstablenet_amd -- MI355X (gfx950) implementation of PWStableNet's hot path.

Layout:
  csrc/            HIP kernels + the C ABI (include/pwstable.h) -> libpwstable_hip.so
  hipabi.py        ctypes binding of that ABI (raw pointers, sizes, stream)
  lib/             host-side mirror of the reference interface: lib.cfg (opt) and lib.networks_cascading (define_G)
  functional.py    grid_sample / affine_grid / UpsamplingBilinear2d / fused 720p warp on the HIP kernels
  spec.py, synth.py  layer table and deterministic synthetic weights / inputs (pure numpy)

Importing this package does not touch the GPU and does not import the oracle.
"""
__version__ = "0.1.0"

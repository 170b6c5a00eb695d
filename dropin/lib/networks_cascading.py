"""Shim: `from lib.networks_cascading import define_G, define_D, GANLoss` resolves to the MI355X build."""
from pwstablenet_amd.lib.networks_cascading import *  # noqa: F401,F403
from pwstablenet_amd.lib.networks_cascading import (GANLoss, UnetGenerator, define_D, define_G, init_net,  # noqa: F401
                                                    init_weights)

# The driver also calls torch.nn.functional.grid_sample / affine_grid and torch.nn.UpsamplingBilinear2d directly
# (main_new.py:106-118,195-197,708,716): route device tensors of those calls to the HIP kernels (CPU tensors pass through).
from pwstablenet_amd import routing as _routing  # noqa: E402

_routing.install()

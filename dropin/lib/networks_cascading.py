"""Shim: `from lib.networks_cascading import define_G, define_D, GANLoss` resolves to the MI355X build."""
from pwstablenet_amd.lib.networks_cascading import *  # noqa: F401,F403
from pwstablenet_amd.lib.networks_cascading import (GANLoss, UnetGenerator, define_D, define_G, init_net,  # noqa: F401
                                                    init_weights)

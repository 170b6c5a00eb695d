"""Shim: `from lib.cfg import opt` / `from lib.cfg import *` resolve to the MI355X build's config mirror."""
from pwstablenet_amd.lib.cfg import *  # noqa: F401,F403
from pwstablenet_amd.lib.cfg import (argparse, cudnn, index_sample, index_sample_discriminator, np, opt, parser,  # noqa: F401
                                     period, test_files, train_files, val_files)

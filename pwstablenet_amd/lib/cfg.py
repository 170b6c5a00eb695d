"""Config surface of the hot path.

Interface mirror of the reference's ``lib/cfg.py`` (reference lib/cfg.py:1-57): the same flag names,
types and defaults, the same module-level names (``opt``, ``period``, ``index_sample``,
``index_sample_discriminator``, ``train_files``, ``val_files``, ``test_files``, ``np``, ``cudnn``,
``parser``, ``argparse``), so ``from lib.cfg import opt`` and ``from lib.cfg import *`` keep working
(tests/test_boundary.py checks every flag against tests/golden/reference_meta.json).

Deliberate differences:
  * the reference parses ``sys.argv`` strictly at import time (lib/cfg.py:43) and therefore crashes under
    any foreign command line (pytest, torchrun, an embedding application); unknown arguments are ignored here;
  * the seven flags only the older driver ``main.py`` reads are accepted as well;
  * ``cudnn.benchmark = True`` is kept for parity of side effects; it does not affect the HIP kernels.
"""
import argparse
import sys

import numpy as np
import torch.backends.cudnn as cudnn

period = 30  # frames either side of the current one are period//2; the generator sees period+1 gray frames

# (flag, type, default, help).  type None = plain string; 'flag' = store_true.  `type=bool` entries keep the
# reference's argparse quirk on purpose (any non-empty string parses as True, reference lib/cfg.py:25,35,37).
_REFERENCE_FLAGS = (
    ("continue_train", int, 5, "epoch whose checkpoint is loaded when training starts"),
    ("checkpoint_dir", None, "unet_256_kalman_backup", "sub-directory of checkpoint/"),
    ("mode", None, "test", "'train' runs the training loop, anything else runs video inference"),
    ("num_layer", int, 3, "cascade stages"),
    ("batchSize", int, 16, "training batch size"),
    ("test_dir", None, "result_shapeloss0.01", "output directory of the video loop"),
    ("nEpochs", int, 80, "epochs"),
    ("input_nc", int, period + 1, "gray frames per window"),
    ("output_nc", int, 2, "warp-field components"),
    ("ngf", int, 64, "generator base width"),
    ("ndf", int, 32, "discriminator base width"),
    ("lr", float, 0.0001, "Adam learning rate"),
    ("beta1", float, 0.5, "Adam beta1"),
    ("cuda", "flag", True, "always on"),
    ("threads", int, 16, "data-loader workers"),
    ("seed", int, 123, "RNG seed"),
    ("lamd", int, 10, "weight of the L1 term"),
    ("input_size", int, 256, "network input height = width"),
    ("use_gan", bool, False, "adversarial training"),
    ("start_gan", int, 40, "first epoch with the GAN loss"),
    ("path_feature", None, "../../feature_add/", "training feature points"),
    ("path_affine", None, "../../affine640_add/", "training affine matrices"),
    ("path_image", None, "../../image256_rgb_blank_add/", "training frames"),
    ("path_adjacent", None, "../../feature_adjacent_add/", "adjacent-frame homographies"),
    ("number_feature", int, 400, "feature points per frame"),
    ("period_D", int, 3, "discriminator window is 2*period_D+1"),
    ("balance_gd", float, 0.1, "generator/discriminator balance"),
    ("block", int, 16, "shape-loss block size"),
    ("shapeloss", bool, True, "use the shape loss"),
    ("shapeloss_weight", float, 1, "shape-loss weight"),
    ("use_BN", bool, False, "BatchNorm after every conv"),
    ("visdom_port", int, 7007, "visdom port"),
    ("decreaselr", int, 8, "epochs per 10x learning-rate decay"),
)
# read by the stale driver main.py only (main.py:28,55,63,222,223,283,678)
_LEGACY_FLAGS = (
    ("dataset", None, "", "legacy"),
    ("train", "flag", True, "legacy"),
    ("dir_logs", None, "logs", "legacy"),
    ("testBatchSize", int, 1, "legacy"),
    ("affine_weight", float, 1.0, "legacy"),
    ("start_loss_affine", int, 0, "legacy"),
)


def _build_parser():
    p = argparse.ArgumentParser(description="PWStableNet hot path on MI355X: configuration")
    for name, typ, default, text in _REFERENCE_FLAGS + _LEGACY_FLAGS:
        if typ == "flag":
            p.add_argument("--" + name, action="store_true", default=default, help=text)
        elif typ is None:
            p.add_argument("--" + name, default=default, help=text)
        else:
            p.add_argument("--" + name, type=typ, default=default, help=text)
    p.add_argument("--gpu_ids", type=int, nargs="*", default=[0], help="legacy; ignored (one process per GPU)")
    # not in the reference: arithmetic of the conv contractions on the MI355X matrix cores (UnetGenerator.set_math)
    p.add_argument("--math", choices=("fp32", "bf16"), default="fp32",
                   help="fp32: exact-fp32 MFMA (parity path); bf16: bf16 MFMA with fp32 accumulation (training configs)")
    # not in the reference: coordinate convention of affine_grid / grid_sample.  The reference is pinned to "pytorch 0.4.0+"
    # (README.md:27), where both behaved as align_corners=True; the torch it can be imported with today (>= 1.3) defaults to
    # False, which is what the goldens and this build default to.  Set it to load checkpoints trained under torch < 1.3.
    p.add_argument("--align_corners", type=int, choices=(0, 1), default=0,
                   help="0: torch >= 1.3 semantics of affine_grid / grid_sample (default); 1: torch < 1.3 (legacy checkpoints)")
    return p


parser = _build_parser()
REFERENCE_FLAGS = [f[0] for f in _REFERENCE_FLAGS]
opt, _unknown_args = parser.parse_known_args(sys.argv[1:] if getattr(sys, "argv", None) else [])

_half = period // 2
index_sample = np.arange(-_half, _half + 1)
index_sample_discriminator = np.arange(-opt.period_D, opt.period_D + 1)

train_files = [91, 92, 93, 94, 95, 96]
val_files = [6]
test_files = [4, 8, 34, 39, 52, 27, 29, 57]

cudnn.benchmark = True

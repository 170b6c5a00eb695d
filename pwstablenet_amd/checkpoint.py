"""Checkpoint files in the reference's format (main_new.py:433-443 ``checkpoint(epoch)`` writes
``{'net': netG.state_dict(), 'epoch': epoch}`` to ``checkpoint/<checkpoint_dir>/netG_model_epoch_<epoch>.pth``;
main_new.py:469-471 reads it back with ``netG.load_state_dict(torch.load(path)['net'])``).  The state dict carries the
``module.`` prefix of the reference's DataParallel wrapper and torch-layout tensors (OIHW / IOHW), which is what
``define_G`` of this package exposes as well (tests/test_boundary.py pins the keys and shapes against the reference), so the
authors' ``netG_model_epoch_*.pth`` load unchanged and files written here load into the reference.  Host logic only."""
import os

import torch


def checkpoint_path(checkpoint_dir, epoch, root="checkpoint"):
    """``checkpoint/{checkpoint_dir}/netG_model_epoch_{epoch}.pth`` (main_new.py:440)."""
    return os.path.join(root, str(checkpoint_dir), "netG_model_epoch_{}.pth".format(epoch))


def save_checkpoint(netG, epoch, checkpoint_dir, root="checkpoint"):
    """main_new.py:433-443.  Tensors are saved from the host copy of the state dict, so the file loads on a box without a GPU."""
    path = checkpoint_path(checkpoint_dir, epoch, root)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    state = {"net": {k: v.detach().cpu() for k, v in netG.state_dict().items()}, "epoch": epoch}
    torch.save(state, path)
    return path


def load_checkpoint(netG, path, map_location="cpu"):
    """main_new.py:469-471: returns the stored epoch.  Accepts the reference's ``{'net', 'epoch'}`` files and bare state
    dicts; a state dict saved without the DataParallel wrapper (no ``module.`` prefix) is re-keyed.  Strict: a missing or
    unexpected key raises, as ``load_state_dict`` does in the reference."""
    obj = torch.load(path, map_location=map_location, weights_only=True)
    state = obj["net"] if isinstance(obj, dict) and "net" in obj else obj
    if not isinstance(state, dict) or not state:
        raise RuntimeError("load_checkpoint: %s holds no state dict" % path)
    want_prefix = next(iter(netG.state_dict())).startswith("module.")
    have_prefix = next(iter(state)).startswith("module.")
    if want_prefix and not have_prefix:
        state = {"module." + k: v for k, v in state.items()}
    elif have_prefix and not want_prefix:
        state = {k[len("module."):]: v for k, v in state.items()}
    netG.load_state_dict(state)
    return obj.get("epoch") if isinstance(obj, dict) else None

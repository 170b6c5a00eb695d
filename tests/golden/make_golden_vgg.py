#!/usr/bin/env python3
"""Golden vectors for the VGG-16 perceptual term (reference lib/utils.py:11-32 ``GeneratorLoss``:
``nn.Sequential(*list(vgg16(pretrained=True).features)[:31])`` + ``MSELoss``), produced by CALLING torchvision ITSELF.
``torchvision`` is NOT in the build image, so this script cannot run there: it is committed so that the day an image has it,
``python tests/golden/make_golden_vgg.py`` writes ``tests/golden/vgg.npz`` and tests/test_objective_golden.py::
test_vgg_restatement_vs_torchvision_fixture / tests/test_hip_perceptual.py::test_features_vs_torchvision_fixture stop saying
"parity unpinned".

What is stored: torchvision's module tree of ``vgg16().features[:31]`` (index, class name, conv shapes, state-dict key list -- what
``pwstablenet_amd.perceptual.VGG16Features.load_state_dict`` must accept), and ONE forward + the MSE loss of two seeded image batches
through that stack with SEEDED weights (pwstablenet_amd/synth.py ``vgg_weights``; pretrained weights need the network).  With
``--pretrained`` (weights already in the torch hub cache) the same forward is stored for the pretrained weights as a checksum and a
strided sample.  Only seeds, key names / shapes and torchvision's OUTPUTS are stored; no reference source is copied.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def seeded_vgg_weights(seed=4):
    """[w, b] per conv of VGG-16 in order; He-scaled so that activations neither die nor explode through 13 layers."""
    cfg = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
    rs = np.random.RandomState(seed)
    out, cin = [], 3
    for cout in cfg:
        out.append((rs.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32))
        out.append((rs.standard_normal(cout) * 0.05).astype(np.float32))
        cin = cout
    return out


def images(seed, n=2, size=64):
    rs = np.random.RandomState(seed)
    return rs.uniform(-1, 1, (n, 3, size, size)).astype(np.float32)


def main():
    try:
        import torch
        from torchvision.models.vgg import vgg16
    except ImportError:
        print("make_golden_vgg.py: torchvision is not importable in this image -- nothing written (the VGG oracle stays 'parity unpinned')")
        return 1
    net = vgg16()
    feats = torch.nn.Sequential(*list(net.features)[:31]).eval()
    out = {"torchvision_version": np.array(__import__("torchvision").__version__),
           "modules": np.array(["%d:%s" % (i, type(m).__name__) for i, m in enumerate(feats)]),
           "state_keys": np.array(list(feats.state_dict().keys())),
           "state_shapes": np.array([",".join(map(str, v.shape)) for v in feats.state_dict().values()])}
    w = seeded_vgg_weights()
    convs = [m for m in feats if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 13
    with torch.no_grad():
        for i, c in enumerate(convs):
            c.weight.copy_(torch.from_numpy(w[2 * i])), c.bias.copy_(torch.from_numpy(w[2 * i + 1]))
        a, b = torch.from_numpy(images(21)), torch.from_numpy(images(22))
        fa, fb = feats(a), feats(b)
        out["features_a"] = fa.numpy()
        out["loss_ab"] = np.array(float(torch.nn.functional.mse_loss(fa, fb)))
    if "--pretrained" in sys.argv:
        net_p = vgg16(pretrained=True)
        fp = torch.nn.Sequential(*list(net_p.features)[:31]).eval()
        with torch.no_grad():
            f = fp(torch.from_numpy(images(21)))
        out["pretrained_features_sum"] = np.array(float(f.double().sum()))
        out["pretrained_features_sample"] = f.numpy()[:, ::37, :, :]
    np.savez_compressed(os.path.join(HERE, "vgg.npz"), **out)
    print("wrote", os.path.join(HERE, "vgg.npz"))
    return 0


if __name__ == "__main__":
    sys.exit(main())

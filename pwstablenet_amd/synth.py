"""Deterministic synthetic weights and inputs (numpy only).

There is no network access for datasets or the authors' checkpoint, and the 194 MB weight set is too
large to commit, so weights are *regenerated* on every side (golden generator, tests, bench, smoke)
by this filler: ``numpy.random.RandomState(seed)`` -> one ``standard_normal`` draw per tensor in
state-dict order, scaled by a documented rule.

Two weight sets are used:
  * ``W1`` ("xavier-like", gain 0.5): moderate activations; residual field |max| ~ 0.2-0.4.
  * ``W2`` ("kaiming-like"): activations O(1) everywhere, residual saturates towards tanh(tanh(.)) =
    0.76 and the field leaves [-1, 1], so grid_sample's out-of-bounds taps are exercised.
In both sets ``linear`` gets bias ``[1,0,0,0,1,0]`` and small weights so the affine part of the field
is identity + perturbation (the reference's default N(0, 0.02) init gives a degenerate field ~ 0).
"""
import numpy as np

from .spec import layer_specs, weight_shape

IDENTITY_THETA = np.array([1, 0, 0, 0, 1, 0], dtype=np.float32)


def _eff_fan_in(ls):
    """Number of input taps contributing to one output element."""
    if ls.kind == "conv":
        return ls.cin * ls.k * ls.k
    return ls.cin * ls.k * ls.k // (ls.s * ls.s)


def make_weights(kind="W1", seed=123, input_nc=31, output_nc=2, ngf=64):
    """Returns an ordered dict-like list [(key, float32 ndarray)] in state-dict order."""
    if kind not in ("W1", "W2"):
        raise ValueError("unknown weight set %r" % (kind,))
    rs = np.random.RandomState(seed)
    gain = {"W1": 0.5, "W2": 2.0}[kind]
    out = []
    for ls in layer_specs(input_nc, output_nc, ngf):
        std = np.sqrt(gain / _eff_fan_in(ls))
        w = rs.standard_normal(weight_shape(ls)).astype(np.float32) * np.float32(std)
        b = rs.standard_normal((ls.cout,)).astype(np.float32) * np.float32(0.05)
        if ls.name.startswith("linear."):
            w *= np.float32(0.1)
            b = IDENTITY_THETA + b * np.float32(0.2)
        out.append((ls.name + ".weight", w))
        out.append((ls.name + ".bias", b.astype(np.float32)))
    return out


def smooth_frames_u8(n, c, h, w, seed=123):
    """uint8 frames made of a ramp + 4 low-frequency sinusoids; consecutive channels are the same scene
    under a small random translation (a shaky 31-frame window).  Smooth on purpose: the warped-frame
    error is |grad(image)| * (W/2) * field error, so noise images would measure image gradient, not us."""
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    out = np.empty((n, c, h, w), dtype=np.uint8)
    for i in range(n):
        fx = rs.uniform(0.5, 3.0, 4) * 2 * np.pi / w
        fy = rs.uniform(0.5, 3.0, 4) * 2 * np.pi / h
        ph = rs.uniform(0, 2 * np.pi, 4)
        amp = rs.uniform(10, 30, 4)
        for j in range(c):
            dx, dy = rs.uniform(-4, 4, 2)
            img = 100.0 + 40.0 * (xx + dx) / w + 30.0 * (yy + dy) / h
            for k in range(4):
                img = img + amp[k] * np.sin(fx[k] * (xx + dx) + fy[k] * (yy + dy) + ph[k])
            out[i, j] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def make_window(n, input_nc=31, size=256, seed=123):
    """Network input as the reference drivers build it (main_new.py:650): uint8/255*2-1, float32."""
    u8 = smooth_frames_u8(n, input_nc, size, size, seed)
    return (u8.astype(np.float32) / np.float32(255) * np.float32(2) - np.float32(1)).astype(np.float32)


def make_frames(n, c=3, h=256, w=256, seed=321):
    """RGB frames as float32 in 0..255 (main_new.py:679-684)."""
    return smooth_frames_u8(n, 1, h, w, seed).repeat(c, axis=1).astype(np.float32) * \
        np.linspace(1.0, 0.8, c, dtype=np.float32).reshape(1, c, 1, 1)


def noise_window(n, input_nc=31, size=256, seed=123):
    """BASELINE.md throughput input: RandomState(seed).randint(0,256,...)/255*2-1."""
    rs = np.random.RandomState(seed)
    u8 = rs.randint(0, 256, (n, input_nc, size, size)).astype(np.uint8)
    return (u8.astype(np.float32) / np.float32(255) * np.float32(2) - np.float32(1)).astype(np.float32)


def make_bn_state(seed=321, input_nc=31, output_nc=2, ngf=64):
    """BatchNorm2d tensors of the ``use_BN=True`` variant (reference lib/networks_cascading.py:253-341: one BatchNorm2d after
    every conv, ``<block>.<seq>.1.*``), in state-dict order per layer: weight (gamma), bias (beta), running_mean,
    running_var, num_batches_tracked.  Non-trivial on purpose so that a wrong fold shows up."""
    rs = np.random.RandomState(seed)
    out = []
    for ls in layer_specs(input_nc, output_nc, ngf):
        base = ls.name[:-1] + "1"   # "<block>.<seq>.0" -> "<block>.<seq>.1"
        c = ls.cout
        out.append((base + ".weight", rs.uniform(0.6, 1.4, c).astype(np.float32)))
        out.append((base + ".bias", (rs.standard_normal(c) * 0.1).astype(np.float32)))
        out.append((base + ".running_mean", (rs.standard_normal(c) * 0.1).astype(np.float32)))
        out.append((base + ".running_var", rs.uniform(0.5, 1.5, c).astype(np.float32)))
        out.append((base + ".num_batches_tracked", np.array(7, dtype=np.int64)))
    return out


def make_train_batch(n, seed=123, size=256, period=30, number_feature=400, stable_extra=0):
    """One item batch as the reference's ``customData.__getitem__`` collates it (lib/utils.py:154-244), synthetic:
      images1, images2     uint8 (n, period+1 + 3 + 3 + stable_extra, size, size): the gray window, the unstable RGB frame,
                           the stable RGB frame (+ the discriminator's gray frames when GAN training is on);
      features1, features2 float64 (n, number_feature, 6): [stable x, y, 1, unstable x, y, 1] in normalised coordinates;
      affine1, affine2     float64 (n, 2, 3) (crop boxes for the discriminator; unused without GAN);
      feature_adjacent     float64 (n, 2, 3): affine map between the two consecutive stable frames.
    The second item is the first one a frame later (a small extra shift), as in the reference."""
    rs = np.random.RandomState(seed)
    c = period + 1 + 3 + 3 + stable_extra
    base = smooth_frames_u8(n, c + 1, size, size, seed + 1)
    images1, images2 = base[:, :c].copy(), base[:, 1:c + 1].copy()

    def feats():
        stable = rs.uniform(-0.96, 0.96, (n, number_feature, 2))
        unstable = stable + rs.normal(0, 0.03, (n, number_feature, 2))
        one = np.ones((n, number_feature, 1))
        return np.concatenate([stable, one, unstable, one], axis=2)

    def near_identity(scale):
        return np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float64)[None] + rs.normal(0, scale, (n, 2, 3))

    features1, features2 = feats(), feats()
    affine1, affine2 = near_identity(0.01), near_identity(0.01)
    feature_adjacent = near_identity(0.02)
    return images1, features1, affine1, images2, features2, affine2, feature_adjacent

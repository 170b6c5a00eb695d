"""PyTorch-CPU restatement of the hot path -- TEST INFRASTRUCTURE ONLY (same rules as pws_oracle.c).

The reference's arithmetic is PyTorch's own ATen CPU kernels (oneDNN convolutions); the reference's Python
cannot travel to the GPU box, so this module restates the same graph with ``torch.nn.functional`` calls on
plain tensors.  It serves two purposes:
  * a second, independent oracle (different summation order than pws_oracle.c) pinned to the same goldens;
  * the strongest honest host-CPU baseline for bench.py's ``cpu_baseline`` leg: it executes the ops the
    reference would execute on this box's cores.

Follows reference lib/networks_cascading.py:152-237 (forward), :245-350 (blocks) and the driver's
warp step main_new.py:106,716.
"""
import torch
import torch.nn.functional as F

# state-dict order (see pwstablenet_amd/spec.py): index of each layer's weight; bias is +1
_TRANSFER, _DOWN1, _UP7, _OUT, _DB1, _UB7, _FLATTEN, _LINEAR = 0, 2, 16, 30, 32, 60, 88, 90


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


def netg_forward(params, x, is_training=True, align_corners=False):
    """params: 92 tensors in state-dict order; x: N,31,256,256.  Returns like the reference's forward."""
    P = params

    def conv(i, t, s, p):
        return F.conv2d(t, P[i], P[i + 1], stride=s, padding=p)

    def convT(i, t, s, p):
        return F.conv_transpose2d(t, P[i], P[i + 1], stride=s, padding=p)

    def down(i, t):
        return _lrelu(conv(_DOWN1 + 2 * (i - 1), t, 2, 1))

    def down_bottom(k, left, up_in):
        cs, mp = _DB1 + 4 * (k - 1), _DB1 + 4 * (k - 1) + 2
        c = _lrelu(conv(cs, up_in, 1, 1))
        return _lrelu(conv(mp, c if left is None else torch.cat([left, c], 1), 2, 1))

    def up(level, a, skip):
        u = F.relu(convT(_UP7 + 2 * (7 - level), a, 2, 1))
        return u if skip is None else torch.cat([u, skip], 1)

    def up_bottom(level, x_up, x_left, x_before):
        mp = _UB7 + 4 * (7 - level)
        e = F.relu(convT(mp + 2, x_up, 1, 1))
        v = F.relu(convT(mp, torch.cat([e, x_left], 1), 2, 1))
        return v if x_before is None else torch.cat([v, x_before], 1)

    def affine(t):
        theta = _lrelu(conv(_LINEAR, _lrelu(conv(_FLATTEN, t, 1, 0)), 1, 0)).view(-1, 2, 3)
        return F.affine_grid(theta, torch.Size((theta.shape[0], 3, 256, 256)), align_corners=align_corners)

    def resid(t):
        return torch.tanh(torch.tanh(conv(_OUT, t, 1, 1))).permute(0, 2, 3, 1)

    x11 = _lrelu(conv(_TRANSFER, x, 1, 2))
    x12 = down(1, x11); x13 = down(2, x12); x14 = down(3, x13); x15 = down(4, x14)  # noqa: E702
    x16 = down(5, x15); x17 = down(6, x16); x18 = down(7, x17)  # noqa: E702
    a1 = affine(x18)
    x177 = up(7, x18, x17); x166 = up(6, x177, x16); x155 = up(5, x166, x15)  # noqa: E702
    x144 = up(4, x155, x14); x133 = up(3, x144, x13); x122 = up(2, x133, x12)  # noqa: E702
    r1 = resid(up(1, x122, None)) if is_training else None

    x22 = down_bottom(1, None, x11)
    x23 = down_bottom(2, x22, x12); x24 = down_bottom(3, x23, x13); x25 = down_bottom(4, x24, x14)  # noqa: E702
    x26 = down_bottom(5, x25, x15); x27 = down_bottom(6, x26, x16); x28 = down_bottom(7, x27, x17)  # noqa: E702
    a2 = affine(x28)
    x277 = up_bottom(7, x18, x28, x27); x266 = up_bottom(6, x177, x277, x26)  # noqa: E702
    x255 = up_bottom(5, x166, x266, x25); x244 = up_bottom(4, x155, x255, x24)  # noqa: E702
    x233 = up_bottom(3, x144, x244, x23); x222 = up_bottom(2, x133, x233, x22)  # noqa: E702
    r2 = resid(up_bottom(1, x122, x222, None)) if is_training else None

    x32 = down_bottom(1, None, x11)  # recomputed, as the reference does (:200)
    x33 = down_bottom(2, x32, x22); x34 = down_bottom(3, x33, x23); x35 = down_bottom(4, x34, x24)  # noqa: E702
    x36 = down_bottom(5, x35, x25); x37 = down_bottom(6, x36, x26); x38 = down_bottom(7, x37, x27)  # noqa: E702
    a3 = affine(x38)
    x377 = up_bottom(7, x28, x38, x37); x366 = up_bottom(6, x277, x377, x36)  # noqa: E702
    x355 = up_bottom(5, x266, x366, x35); x344 = up_bottom(4, x255, x355, x34)  # noqa: E702
    x333 = up_bottom(3, x244, x344, x33); x322 = up_bottom(2, x233, x333, x32)  # noqa: E702
    r3 = resid(up_bottom(1, x222, x322, None))
    if is_training:
        return [r1 + a1, r2 + a2, r3 + a3], [r1, r2, r3]
    return r3 + a3


def stabilize_step(params, window, frames):
    """One unit of the BASELINE metric on CPU: field = netG(window, False); warped = grid_sample(frame, field)."""
    with torch.no_grad():
        grid = netg_forward(params, window, is_training=False)
        return F.grid_sample(frames, grid, mode="bilinear", padding_mode="zeros", align_corners=False)

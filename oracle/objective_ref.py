"""PyTorch-CPU restatement of the training objective that wraps the hot path -- TEST INFRASTRUCTURE ONLY
(same rules as pws_oracle.c / torch_ref.py: imported by tests/, smoke() and bench.py's cpu_baseline leg only).

Restates, on plain CPU tensors and with the reference line cited per function:
  * ``pre_processing``   lib/utils.py:246-255   uint8 -> [-1,1], split into unstable / stable / feature halves
  * ``loss_calculate``   lib/utils.py:339-362   feature-point loss, L1 pixel loss, field smoothness (monitor only)
  * ``shape_basis``      lib/utils.py:427-447   bilinear-corner basis of a block (generate_affine_matrix)
  * ``loss_shape``       lib/utils.py:405-425   fp64 least-squares block projection residual (loss_pixel1)
  * ``objective``        main_new.py:101-118,184-212   how train() composes them (no GAN; the VGG perceptual term
                         ``generator_criterion`` needs torchvision's pretrained VGG-16 and is left to the caller)
Pinned by tests/golden/objective.npz, which tests/golden/make_golden_objective.py produced by calling the
reference's own ``pre_propossing`` / ``loss_calulate`` / ``loss_pixel1`` / ``generate_affine_matrix`` / ``netG``.
"""
import numpy as np
import torch
import torch.nn.functional as F


def pre_processing(images, features, period=30):
    """lib/utils.py:246-255."""
    images = images.float() * (1. / 255) * 2 - 1
    images_unstable = images[:, 0:period + 1 + 3]
    images_stable = images[:, period + 1 + 3:]
    feature_stable = features[:, :, 0:3].permute(0, 2, 1)
    feature_unstable = features[:, :, 3:6].permute(0, 2, 1)
    return images_stable, images_unstable, feature_stable, feature_unstable


def loss_calculate(grid, feature_stable, feature_unstable, fake, real, batch, size=256, number_feature=400,
                   cfg_batch=None):
    """lib/utils.py:339-362.  ``batch`` is the loop bound the caller passes, ``cfg_batch`` the divisor (opt.batchSize)."""
    cfg_batch = batch if cfg_batch is None else cfg_batch
    feature_loss = 0
    for i in range(batch):
        iy = ((feature_stable[i, 1, :] + 1) * size / 2).int().long()
        ix = ((feature_stable[i, 0, :] + 1) * size / 2).int().long()
        grid_pos = grid[i, iy, ix, :]                                     # (F, 2)
        d = feature_unstable[i, 0:2, :] - grid_pos.t()
        feature_loss = feature_loss + (d * d).sum() / number_feature       # == pow(dist(a, b), 2) / number_feature
    feature_loss = feature_loss / cfg_batch
    mse_loss = torch.mean(torch.abs(real[:, 0:3] - fake))
    delta_x = torch.abs(grid[:, :, 0:size - 1, :] - grid[:, :, 1:size, :])
    delta_y = torch.abs(grid[:, 0:size - 1, :, :] - grid[:, 1:size, :, :])
    delta = (torch.mean(delta_x) + torch.mean(delta_y)) / 2
    return mse_loss, delta, feature_loss


def shape_basis(block):
    """lib/utils.py:427-447 for one block: (block*block, 4) bilinear weights of the block's four corners, pixel-major
    (y * block + x); the reference tiles this same matrix over every block."""
    x2 = y2 = block - 1
    y, x = np.meshgrid(np.arange(block), np.arange(block), indexing="ij")
    x, y = x.reshape(-1).astype(np.float64), y.reshape(-1).astype(np.float64)
    q = np.stack([(x2 - x) * (y2 - y), x * (y2 - y), (x2 - x) * y, x * y], axis=1) / (x2 * y2)
    return q


def loss_shape(resid, block=16, size=256):
    """lib/utils.py:405-425 (loss_pixel1): every (size/block)^2-pixel block of the residual field is projected, in
    fp64, onto the span of the bilinear corner basis; the loss is the L1 distance to the projection, summed (no mean)."""
    n = resid.shape[0]
    bs = size // block
    a = torch.from_numpy(shape_basis(block))                               # (bs*bs, 4) -- the reference requires bs == block
    assert a.shape[0] == bs * bs, "the reference's basis has block^2 rows and its blocks (size/block)^2 pixels"
    b = resid.to(torch.float64).reshape(n, block, bs, block, bs, 2).permute(1, 3, 0, 2, 4, 5).reshape(-1, bs * bs, 2)
    proj = a @ (torch.inverse(a.t() @ a) @ a.t())
    ab = proj @ b
    return torch.dist(ab, b, 1).to(torch.float32)


def objective(grids1, resid1, grids2, resid2, image_unstable1, image_stable1, feature_stable1, feature_unstable1,
              image_unstable2, image_stable2, feature_stable2, feature_unstable2, feature_adjacent, batch, size=256,
              period=30, number_feature=400, num_layer=3, lamd=10, shapeloss=True, shapeloss_weight=1.0, block=16):
    """main_new.py:101-118 (warps) and :184-212 (loss composition), generator part without GAN and without the VGG
    term.  Returns a dict of the scalar losses and the warped frames."""
    def warps(iu, grids):
        rgb = (iu[:, period + 1:period + 1 + 3] + 1) * 127.5
        return [F.grid_sample(rgb, g, align_corners=False) / 127.5 - 1 for g in grids]
    fake1, fake2 = warps(image_unstable1, grids1), warps(image_unstable2, grids2)
    loss_mse = loss_feature = loss_delta = loss_g2 = 0
    loss_pixel = torch.zeros(())
    theta = feature_adjacent.view(-1, 2, 3).float()
    for nl in range(num_layer):
        m1, d1, f1 = loss_calculate(grids1[nl], feature_stable1, feature_unstable1, fake1[nl], image_stable1, batch,
                                    size, number_feature)
        m2, d2, f2 = loss_calculate(grids2[nl], feature_stable2, feature_unstable2, fake2[nl], image_stable2, batch,
                                    size, number_feature)
        loss_mse = loss_mse + m1 + m2
        loss_feature = loss_feature + f1 + f2
        loss_delta = loss_delta + d1 + d2
        grid = F.affine_grid(theta, fake1[nl].size(), align_corners=False)
        out2_to_1 = F.grid_sample(fake2[nl], grid, align_corners=False)
        loss_g2 = loss_g2 + torch.mean(torch.abs(out2_to_1 - fake1[nl]))
        if shapeloss:   # assigned, not accumulated: only the last stage's term survives the loop (main_new.py:202-203)
            loss_pixel = loss_shape(resid1[nl], block, size) * shapeloss_weight + \
                loss_shape(resid2[nl], block, size) * shapeloss_weight
    loss_g1 = loss_feature + loss_mse + (loss_pixel if shapeloss else 0)
    loss_g = loss_g1 + loss_g2 * lamd
    return {"loss_g": loss_g, "loss_mse": loss_mse, "loss_feature": loss_feature, "loss_delta": loss_delta,
            "loss_g2": loss_g2, "loss_pixel": loss_pixel, "fake1": fake1, "fake2": fake2}


VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


def vgg16_features(params, x, store_bf16=False):
    """``nn.Sequential(*list(vgg16.features)[:31])`` (lib/utils.py:14-15) as functional calls; params: [w, b] per conv in
    order.  PARITY UNPINNED against the reference for this function: torchvision and its pretrained weights are absent here,
    so tests run it with seeded random weights (same arithmetic, arbitrary weights).
    ``store_bf16``: model of the HIP path's bf16 mode -- weights, the input and every layer's output rounded to bf16 (fp32
    accumulation, fp32 bias); the casts' backward rounds the activation gradients to bf16 the same way."""
    def rnd(t):
        return t.bfloat16().float() if store_bf16 else t
    x = rnd(x)
    i = 0
    for v in VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
        else:
            x = rnd(F.relu(F.conv2d(x, rnd(params[i]), params[i + 1], padding=1)))
            i += 2
    return x


def generator_loss(params, out_images, target_images):
    """lib/utils.py:22-32: MSELoss between the VGG features of the two image batches."""
    return F.mse_loss(vgg16_features(params, out_images), vgg16_features(params, target_images))

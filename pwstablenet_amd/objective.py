"""The generator objective the reference's ``train()`` wraps round the hot path, on the HIP kernels of
``csrc/objective.hip`` (reference main_new.py:101-118 warps, :184-212 loss composition; lib/utils.py:246-255
``pre_propossing``, :339-362 ``loss_calulate``, :405-447 ``loss_pixel1`` / ``generate_affine_matrix``).

    loss_g = loss_feature + loss_mse + loss_pixel * [shapeloss] + lamd * loss_g2        (+ the caller's VGG term)

Generator part without GAN (the reference default, lib/cfg.py:25).  The VGG perceptual term (``generator_criterion``,
lib/utils.py:11-32) needs torchvision's pretrained VGG-16 and stays with the caller: the warped frames are returned WITH
autograd attached, so ``loss = out.loss_g + my_vgg(out.fake[nl][:n], stable1) ...`` back-propagates through the same
kernels.

The reference runs the generator twice per step (an item and the item one frame later); without BatchNorm that is one
batch of m = 2n samples -- branch 1 = samples [0, n), branch 2 = [n, 2n) -- and every tensor here is laid out that way.
torch is plumbing only (memory, streams, autograd bookkeeping); there is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import hipabi as A

LOSS_NAMES = ("loss_g", "loss_mse", "loss_feature", "loss_delta", "loss_g2", "loss_pixel")
_NQ = 16   # slot quantities: L1[3], temporal[3], feature[3], (dx, dy)[3], shape


def _planes(t, h, w, what):
    """(data_ptr, sample stride in elements) of an (m, 3, h, w) view whose 3 planes are contiguous per sample."""
    if t.dim() != 4 or t.shape[1] < 3 or t.shape[2] != h or t.shape[3] != w or t.stride()[1:] != (h * w, w, 1):
        raise RuntimeError("objective: %s must be an (m, >=3, %d, %d) view with contiguous planes, got %s strides %s"
                           % (what, h, w, tuple(t.shape), t.stride()))
    return ctypes.c_void_p(t.data_ptr()), t.stride(0)


def u8_normalize(src, out=None):
    """``images.float() * (1. / 255) * 2 - 1`` (lib/utils.py:247) of a uint8 (m, C, H, W) tensor or channel-slice view."""
    A.require_cuda(src, dtype=torch.uint8)
    m = src.shape[0]
    per = int(np.prod(src.shape[1:]))
    if src.stride()[1:] != tuple(int(np.prod(src.shape[k + 1:])) for k in range(1, src.dim())):
        raise RuntimeError("u8_normalize: each sample of the view must be contiguous (a channel range of an NCHW tensor)")
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    A.require_cuda(out)
    if out.shape != src.shape or not out[0:1].is_contiguous():
        raise RuntimeError("u8_normalize: out must have the source's shape with contiguous samples")
    A.check(A.lib().pws_u8_normalize(A.ptr(src), src.stride(0) if m > 1 else per, A.ptr(out), out.stride(0) if m > 1 else per, m,
                                     per, A.current_stream()), "pws_u8_normalize")
    return out


def pre_propossing(images, features, period=30):
    """Same name, arguments and return as lib/utils.py:246-255: ``images`` uint8 (m, C, H, W) -> [-1, 1];
    returns (images_stable, images_unstable, feature_stable, feature_unstable)."""
    images = u8_normalize(images.contiguous())
    feature_stable = features[:, :, 0:3].permute(0, 2, 1)
    feature_unstable = features[:, :, 3:6].permute(0, 2, 1)
    return images[:, period + 1 + 3:], images[:, 0:period + 1 + 3], feature_stable, feature_unstable


class _Objective(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, rgb, stable, features, theta, *fields):
        """fields = grids[0..L-1] + (resid_last,).  Returns (losses[6], fake[0..L-1])."""
        L_ = cfg["num_layer"]
        grids, resid = [g.contiguous() for g in fields[:L_]], fields[L_].contiguous()
        m, h, w = grids[0].shape[0], cfg["size"], cfg["size"]
        n = m // 2
        A.require_cuda(rgb, stable, features, theta, resid, *grids)
        lib, st = A.lib(), A.current_stream()
        dev = rgb.device
        rgb_p, rgb_s = _planes(rgb, h, w, "rgb")
        stb_p, stb_s = _planes(stable, h, w, "stable")
        slots = torch.zeros((_NQ, A.OBJ_SLOTS), device=dev, dtype=torch.float64)
        sp = slots.data_ptr()

        def slot(q):
            return ctypes.c_void_p(sp + q * A.OBJ_SLOTS * 8)
        fakes = []
        for nl in range(L_):
            fake = torch.empty((m, 3, h, w), device=dev, dtype=torch.float32)
            A.check(lib.pws_warp_norm_fwd(rgb_p, rgb_s, A.ptr(grids[nl]), A.ptr(fake), stb_p, stb_s, slot(nl), m, h, w, st),
                    "pws_warp_norm_fwd")
            A.check(lib.pws_temporal_l1_fwd(A.ptr(fake[:n]), A.ptr(fake[n:]), A.ptr(theta), slot(3 + nl), n, h, w, st),
                    "pws_temporal_l1_fwd")
            A.check(lib.pws_feature_loss_fwd(A.ptr(grids[nl]), A.ptr(features), slot(6 + nl), m, features.shape[1], h, w, st),
                    "pws_feature_loss_fwd")
            A.check(lib.pws_field_smoothness(A.ptr(grids[nl]), slot(9 + 2 * nl), slot(10 + 2 * nl), m, h, w, st),
                    "pws_field_smoothness")
            fakes.append(fake)
        if cfg["shapeloss"]:
            A.check(lib.pws_shape_loss_fwd(A.ptr(resid), slot(15), m, cfg["size"], cfg["block"], st), "pws_shape_loss_fwd")
        losses = torch.empty(len(LOSS_NAMES), device=dev, dtype=torch.float32)
        coef = cfg["coef"](dev)
        A.check(lib.pws_objective_finalize(A.ptr(slots), _NQ, A.ptr(coef), len(LOSS_NAMES), A.ptr(losses), st),
                "pws_objective_finalize")
        ctx.cfg = cfg
        ctx.save_for_backward(rgb, stable, features, theta, resid, *grids, *fakes)
        # loss_g is the objective; the full vector (loss_g, loss_mse, ...) is a report and carries no gradient: asking for the
        # gradient of, say, loss_mse alone raises in autograd instead of silently returning zeros
        loss_g = losses[0:1].clone()
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)   # outputs nobody differentiates arrive as None in backward, not as zero tensors (50 MB each)
        return (loss_g, losses) + tuple(fakes)

    @staticmethod
    def backward(ctx, g_loss, _g_report, *g_fakes):
        cfg = ctx.cfg
        L_ = cfg["num_layer"]
        sv = ctx.saved_tensors
        rgb, stable, features, theta, resid = sv[:5]
        grids, fakes = sv[5:5 + L_], sv[5 + L_:5 + 2 * L_]
        m, h, w = grids[0].shape[0], cfg["size"], cfg["size"]
        n = m // 2
        lib, st = A.lib(), A.current_stream()
        rgb_p, rgb_s = _planes(rgb, h, w, "rgb")
        stb_p, stb_s = _planes(stable, h, w, "stable")
        # d loss_g / d(.) times the upstream gradient of loss_g, which stays on the device (no host sync): the kernels
        # read it through `scale`.
        if g_loss is None:
            scale = torch.zeros(1, device=rgb.device, dtype=torch.float32)
        else:
            scale = g_loss.reshape(1).to(torch.float32).contiguous()
        cnt = float(n) * 3 * h * w
        c_l1, c_t = 1.0 / cnt, cfg["lamd"] / cnt
        c_f = 1.0 / (cfg["number_feature"] * cfg["batch"])
        ggrids = []
        for nl in range(L_):
            gx = g_fakes[nl]
            gextra = torch.zeros_like(fakes[nl]) if gx is None else gx.contiguous().clone()
            if cfg.get("deterministic"):   # ordered gather instead of the atomics' scatter (bit-identical runs)
                scratch = torch.empty((n, 3, h, w), device=rgb.device, dtype=torch.float32)
                A.check(lib.pws_temporal_l1_bwd_det(A.ptr(fakes[nl][:n]), A.ptr(fakes[nl][n:]), A.ptr(theta), c_t, A.ptr(scale),
                                                    A.ptr(gextra[:n]), A.ptr(gextra[n:]), A.ptr(scratch), n, h, w, st), "pws_temporal_l1_bwd_det")
            else:
                A.check(lib.pws_temporal_l1_bwd(A.ptr(fakes[nl][:n]), A.ptr(fakes[nl][n:]), A.ptr(theta), c_t, A.ptr(scale),
                                                A.ptr(gextra[:n]), A.ptr(gextra[n:]), n, h, w, st), "pws_temporal_l1_bwd")
            gg = torch.empty_like(grids[nl])
            A.check(lib.pws_warp_norm_bwd(rgb_p, rgb_s, A.ptr(grids[nl]), stb_p, stb_s, c_l1, A.ptr(scale), A.ptr(gextra),
                                          A.ptr(gg), 0, m, h, w, st), "pws_warp_norm_bwd")
            det = bool(cfg.get("deterministic"))
            feat_bwd = lib.pws_feature_loss_bwd_det if det else lib.pws_feature_loss_bwd
            A.check(feat_bwd(A.ptr(grids[nl]), A.ptr(features), c_f, A.ptr(scale), A.ptr(gg), m, features.shape[1], h, w, st),
                    "pws_feature_loss_bwd_det" if det else "pws_feature_loss_bwd")
            ggrids.append(gg)
        gresid = None
        if cfg["shapeloss"]:
            gresid = torch.empty_like(resid)
            dp_world = cfg.get("dp_world") or 1
            A.check(lib.pws_shape_loss_bwd(A.ptr(resid), float(cfg["shapeloss_weight"]) * float(dp_world), A.ptr(scale), A.ptr(gresid), m,
                                           cfg["size"], cfg["block"], st), "pws_shape_loss_bwd")
        return (None, None, None, None, None) + tuple(ggrids) + (gresid,)


class ObjectiveResult(dict):
    """dict with attribute access: loss_g (differentiable), loss_mse, loss_feature, loss_delta, loss_g2, loss_pixel
    (0-dim device tensors, no host sync) and fake: list of the warped frames per stage, (m, 3, H, W) in [-1, 1]."""
    __getattr__ = dict.__getitem__


class StabObjective:
    """Holds the configuration (``opt`` of lib/cfg.py, or keyword overrides) and evaluates the objective."""

    def __init__(self, opt=None, **kw):
        def get(name, default):
            return kw.get(name, getattr(opt, name, default) if opt is not None else default)
        self.size = int(get("input_size", 256))
        self.number_feature = int(get("number_feature", 400))
        self.batch = int(get("batchSize", 16))      # the divisor of the feature term (lib/utils.py:347 uses opt.batchSize)
        self.lamd = float(get("lamd", 10))
        self.shapeloss = bool(get("shapeloss", True))
        self.shapeloss_weight = float(get("shapeloss_weight", 1))
        self.block = int(get("block", 16))
        self.num_layer = int(get("num_layer", 3))
        self.period = int(kw.get("period", 30))
        # Data parallelism (one process per GPU, gradients AVERAGED over ranks -- distributed.allreduce_tensors): every term of
        # loss_g is a mean over the batch except the shape term, which the reference SUMS over items and pixels
        # (lib/utils.py:421 torch.dist(AB, B, 1)).  The reference's nn.DataParallel evaluates the loss on the gathered outputs of the
        # whole batch, so its gradient of that term is the sum over ALL items; averaging the ranks' gradients would deliver 1/world
        # of it.  grad_average_world multiplies the GRADIENT of the sum-type term so that the averaged gradient equals the
        # reference's; the reported loss values stay local (loss_pixel of the job = SUM of the ranks', the other terms = their
        # MEAN).  It is EXPLICIT: None means 1 here (no scaling) -- ``train_step`` passes the world size for the calls whose
        # gradients it averages (``sync_gradients`` given, or an exchange attached to the generator); a caller that composes the
        # step by hand sets it (or passes ``grad_average_world=`` to the call).  A process group that only shards inference, or a
        # job that SUMS its gradients, therefore never gets the factor; an unset value under an initialised group of several
        # ranks warns once.
        self.grad_average_world = kw.get("grad_average_world", None)
        # True: the two scatter gradients of the objective (the temporal term's warp of fake2, feature points sharing a pixel) run
        # without atomics (pws_temporal_l1_bwd_det / pws_feature_loss_bwd_det); together with UnetGenerator.deterministic a whole
        # train_step then gives bit-identical gradients and weights run to run.  train_step asks for it PER CALL from the
        # generator's flag (``deterministic=`` of the call); this attribute is the objective's own default and is never written.
        self.deterministic = bool(kw.get("deterministic", False))
        self._warned_world = False
        if self.num_layer != 3:
            raise NotImplementedError("StabObjective: the generator has 3 cascaded stages (num_layer=%d)" % self.num_layer)
        if bool(get("use_gan", False)):
            raise NotImplementedError("StabObjective: the adversarial terms (use_gan=True) are outside the accelerated path")
        self._coef = {}

    def _coef_matrix(self, n, device):
        key = (n, str(device))
        if key not in self._coef:
            s = self.size
            cnt = float(n) * 3 * s * s
            c = np.zeros((len(LOSS_NAMES), _NQ), np.float64)
            c[1, 0:3] = 1.0 / cnt                                        # loss_mse   = sum_nl (mean|.|_1 + mean|.|_2)
            c[4, 3:6] = 1.0 / cnt                                        # loss_g2
            c[2, 6:9] = 1.0 / (self.number_feature * self.batch)         # loss_feature
            for nl in range(3):                                          # loss_delta = sum (mean dx + mean dy) / 2
                c[3, 9 + 2 * nl] = 0.5 / (float(n) * s * (s - 1) * 2)
                c[3, 10 + 2 * nl] = 0.5 / (float(n) * (s - 1) * s * 2)
            if self.shapeloss:
                c[5, 15] = self.shapeloss_weight                         # loss_pixel (last stage only, main_new.py:202-203)
            c[0] = c[2] + c[1] + c[5] + self.lamd * c[4]                 # loss_g (main_new.py:205-211, no VGG / GAN)
            self._coef[key] = torch.from_numpy(c).to(device)
        return self._coef[key]

    def __call__(self, grids, resid, rgb_unstable, image_stable, features, feature_adjacent, grad_average_world=None, deterministic=None):
        """grids, resid: the two lists ``netG(x)`` returns (m = 2n samples); rgb_unstable: (m, 3, H, W) view of the
        unstable RGB frames in [-1, 1] (``image_unstable[:, period+1:period+4]``); image_stable: (m, >=3, H, W);
        features: (m, nf, 6) float32 as the loader collates them; feature_adjacent: (n, 2, 3) / (n, 6).
        grad_average_world / deterministic: per-call overrides of the attributes of the same name (None: the attribute)."""
        dp_world = grad_average_world if grad_average_world is not None else self.grad_average_world
        if dp_world is None:
            dp_world = 1
            if self.shapeloss:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    # the shape term is a SUM over the batch in the reference (DataParallel adds the replicas' gradients), every
                    # other term a mean: whether this rank's share must be scaled by the world size depends on whether the caller
                    # averages or sums the ranks' gradients, and a silent default is a silent numerical change -- say which
                    raise ValueError("StabObjective: a process group of %d ranks is initialised, shapeloss is on and grad_average_world is "
                                     "unset: pass grad_average_world = world size if the ranks' gradients are AVERAGED (train_step does), "
                                     "or grad_average_world = 1 if they are summed / not exchanged" % dist.get_world_size())
        det = self.deterministic if deterministic is None else bool(deterministic)
        m = grids[0].shape[0]
        if m % 2 != 0:
            raise ValueError("StabObjective: expects the two branches batched (m = 2n samples), got m = %d" % m)
        n = m // 2
        if features.shape[0] != m or features.shape[2] != 6 or features.shape[1] != self.number_feature:
            raise ValueError("StabObjective: features must be (m, number_feature, 6), got %s" % (tuple(features.shape),))
        theta = feature_adjacent.reshape(-1, 6).to(dtype=torch.float32).contiguous()
        if theta.shape[0] != n:
            raise ValueError("StabObjective: feature_adjacent must hold n = %d affine maps, got %s" % (n, tuple(feature_adjacent.shape)))
        cfg = {"size": self.size, "number_feature": self.number_feature, "batch": self.batch, "lamd": self.lamd,
               "shapeloss": self.shapeloss, "shapeloss_weight": self.shapeloss_weight, "block": self.block,
               "num_layer": self.num_layer, "coef": lambda dev: self._coef_matrix(n, dev), "dp_world": dp_world,
               "deterministic": det}
        out = _Objective.apply(cfg, rgb_unstable, image_stable, features.to(dtype=torch.float32).contiguous(), theta,
                               *grids, resid[self.num_layer - 1])
        loss_g, losses, fakes = out[0], out[1], list(out[2:])
        res = ObjectiveResult({name: losses[i] for i, name in enumerate(LOSS_NAMES)})
        res["loss_g"] = loss_g[0]      # the differentiable one; the other five are reports (no gradient)
        res["fake"] = fakes
        return res


def check_features_host(features, size=256):
    """The reference indexes the field with ``int((coord + 1) * size / 2)`` and raises IndexError outside [-size, size);
    the kernel cannot raise, so a host-side batch can be validated here (cheap; the loader's tensors are on the host)."""
    idx = ((features[..., 0:2].double() + 1) * size / 2).trunc()
    if bool(((idx < -size) | (idx >= size)).any()):
        raise IndexError("feature coordinates index the %dx%d field out of bounds" % (size, size))


def _dist_world():
    """World size of the initialised default process group, or None without one."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else None


def resolve_grad_world(netG, objective_world, sync_gradients, dist_world):
    """Over how many ranks the gradients of a ``train_step`` are AVERAGED after its backward (StabObjective.grad_average_world: the
    sum-type shape term's gradient is scaled by it).  The objective's own setting wins; an exchange this step knows of
    (``sync_gradients``, a ``grad_sync`` attached to the generator, torch's DistributedDataParallel round it) averages over the group;
    no group, or a group of one: 1.  A multi-rank group and NO known exchange returns None, so that the objective refuses to guess
    (somebody else -- a wrapper this step cannot see -- may be averaging)."""
    if objective_world is not None:
        return objective_world
    world = dist_world if dist_world is not None else 1
    target = getattr(netG, "module", netG)
    if sync_gradients is not None or getattr(target, "grad_sync", None) is not None:
        return world
    if isinstance(netG, torch.nn.parallel.DistributedDataParallel):
        return world
    return 1 if world <= 1 else None


def train_step(netG, optimizerG, batch, objective, perceptual=None, period=30, sync_gradients=None):
    """One generator step of the reference's ``train()`` (main_new.py:84-118,184-216) on device tensors.

    batch: (images1, features1, affine1, images2, features2, affine2, feature_adjacent) as ``customData`` collates them
    (images uint8 (n, period+1+3+3[+..], 256, 256); affine1/2 are only used by the discriminator).
    The two generator forwards of the reference run as ONE batch of 2n windows.  ``sync_gradients(params)``: called
    between backward and the optimizer step (data-parallel training: ``distributed.allreduce_gradients``).
    Returns the ObjectiveResult."""
    images1, features1, _a1, images2, features2, _a2, feature_adjacent = batch
    n, c = images1.shape[0], images1.shape[1]
    h, w = images1.shape[2], images1.shape[3]
    dev = images1.device
    win = torch.empty((2 * n, period + 1, h, w), device=dev, dtype=torch.float32)
    rest = torch.empty((2 * n, c - period - 1, h, w), device=dev, dtype=torch.float32)
    for half, img in enumerate((images1, images2)):
        u8_normalize(img[:, :period + 1], win[half * n:(half + 1) * n])
        u8_normalize(img[:, period + 1:], rest[half * n:(half + 1) * n])
    features = torch.cat([features1, features2], 0).to(device=dev, dtype=torch.float32)
    grids, resid = netG(win)
    target = getattr(netG, "module", netG)
    # the generator asks for bit-reproducible gradients: so does the objective round it, for THIS call (the objective may be shared)
    det = bool(getattr(target, "deterministic", False)) or objective.deterministic
    # gradients averaged over ranks after this backward (by sync_gradients or by the exchange attached to the generator): the
    # sum-type shape term's gradient is scaled by the world size (StabObjective.grad_average_world)
    dp_world = resolve_grad_world(netG, objective.grad_average_world, sync_gradients, _dist_world())
    out = objective(grids, resid, rest[:, 0:3], rest[:, 3:], features, feature_adjacent.to(dev), grad_average_world=dp_world,
                    deterministic=det)
    loss = out.loss_g if perceptual is None else out.loss_g + perceptual(out.fake, rest[:, 3:6])
    optimizerG.zero_grad()
    loss.backward()
    if sync_gradients is not None:
        sync_gradients(list(netG.parameters()))
    optimizerG.step()
    return out

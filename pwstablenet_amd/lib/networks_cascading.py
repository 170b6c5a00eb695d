"""Host-side mirror of the reference's generator interface (reference lib/networks_cascading.py).

What is kept identical (the drop-in boundary):
  * ``define_G(input_nc, output_nc, ngf, init_type='normal', init_gain=0.02)`` (an optional trailing
    ``gpu_ids`` of the older driver is accepted and ignored), ``init_weights`` rules, ``NotImplementedError``
    for unknown init types (reference :25-60);
  * ``netG(x)`` -> ``([g1,g2,g3], [r1,r2,r3])`` and ``netG(x, False)`` -> ``g3`` with N,256,256,2 fields
    (reference :152-237); ``is_training`` is a call argument, independent of ``.train()/.eval()``;
  * ``state_dict()`` keys, order, shapes and layouts, including the ``module.`` prefix of the reference's
    DataParallel wrapper (reference :51-52; checkpoints are saved through it, main_new.py:438).

What is different: the module tree only *holds* the parameters (torch = plumbing); the arithmetic is the
HIP executor behind the C ABI (``pws_netg_forward``).  There is no CPU fallback: calling the generator
without a GPU or without libpwstable_hip.so raises.  Data parallelism is one process per GPU (RCCL through
``torch.distributed``), so the wrapper returned by ``define_G`` never scatters over devices.
"""
import ctypes
import threading

import torch
import torch.nn as nn
from torch.nn import init

from .. import hipabi as A
from ..spec import layer_specs
from .cfg import opt


# ------------------------------------------------------------------------------------------------ init
def init_weights(net, init_type='normal', gain=0.02):
    """Same rules as the reference (:25-46): conv / transposed-conv weights by ``init_type``, biases 0,
    BatchNorm weight ~ N(1, gain), bias 0."""
    fillers = {
        'normal': lambda w: init.normal_(w, 0.0, gain),
        'xavier': lambda w: init.xavier_normal_(w, gain=gain),
        'kaiming': lambda w: init.kaiming_normal_(w, a=0, mode='fan_in'),
        'orthogonal': lambda w: init.orthogonal_(w, gain=gain),
    }
    if init_type not in fillers:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            fillers[init_type](m.weight.data)
            if m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif isinstance(m, nn.BatchNorm2d):
            init.normal_(m.weight.data, 1.0, gain)
            init.constant_(m.bias.data, 0.0)
    print('initialize network with %s' % init_type)


class SingleDeviceParallel(nn.Module):
    """Stands where the reference has ``torch.nn.DataParallel`` (:51-52): same ``.module`` attribute and the
    same ``module.``-prefixed state-dict keys, but never scatters -- this build runs one process per GPU and
    all-reduces gradients over RCCL (pwstablenet_amd.distributed)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def init_net(net, init_type='normal', init_gain=0.02, parallel=False):
    if parallel:
        net = SingleDeviceParallel(net)
    init_weights(net, init_type, gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, init_type='normal', init_gain=0.02, gpu_ids=None):
    return init_net(UnetGenerator(input_nc, output_nc, ngf), init_type, init_gain, parallel=True)


def define_D(*args, **kwargs):
    raise NotImplementedError("define_D: the GAN discriminators are outside the accelerated hot path "
                              "(off by default in the reference, lib/cfg.py:25)")


class GANLoss(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("GANLoss: the GAN branch is outside the accelerated hot path")


# ------------------------------------------------------------------------------------------------ model
class _Block(nn.Module):
    """Parameter container for one reference block (down / down_bottom / up / up_bottom)."""


class down(_Block):
    pass


class down_bottom(_Block):
    pass


class up(_Block):
    pass


class up_bottom(_Block):
    pass


def _block_class(name):
    for prefix, cls in (("down_bottom", down_bottom), ("up_bottom", up_bottom), ("up", up)):
        if name.startswith(prefix):
            return cls
    return down


class UnetGenerator(nn.Module):
    """3-stage cascading encoder/decoder regressing an H x W x 2 warp field from ``input_nc`` gray frames."""

    def __init__(self, input_nc, output_nc, ngf=64, use_BN=opt.use_BN):
        super().__init__()
        if output_nc != 2:
            raise ValueError("UnetGenerator: the warp field has 2 components (output_nc=%r)" % (output_nc,))
        if ngf % 16 != 0:
            raise ValueError("UnetGenerator: ngf must be a multiple of 16 for the gfx950 kernels (got %r)" % (ngf,))
        self.input_nc, self.output_nc, self.ngf, self.use_BN = input_nc, output_nc, ngf, bool(use_BN)
        self._specs = layer_specs(input_nc, output_nc, ngf)
        for ls in self._specs:
            block_name, seq_name, _ = ls.name.split(".")
            if not hasattr(self, block_name):
                setattr(self, block_name, _block_class(block_name)())
            if ls.kind == "conv":
                conv = nn.Conv2d(ls.cin, ls.cout, kernel_size=ls.k, stride=ls.s, padding=ls.p, bias=True)
            else:
                conv = nn.ConvTranspose2d(ls.cin, ls.cout, kernel_size=ls.k, stride=ls.s, padding=ls.p, bias=True)
            if block_name == "out":
                act = nn.Tanh()
            elif ls.kind == "conv":
                act = nn.LeakyReLU(0.2, True)
            else:
                act = nn.ReLU(True)
            mods = [conv] + ([nn.BatchNorm2d(ls.cout)] if self.use_BN else []) + [act]
            setattr(getattr(self, block_name), seq_name, nn.Sequential(*mods))
        self._packed = None
        self._packed_key = None
        self._packed_dgrad = None
        self._packed_dgrad_key = None
        self._ws = {}
        self._train_ws = {}
        self._graph_mode = False
        self._graph_alias = False
        self._graph = None     # the graph entry of the stream the last call ran on (tests read it)
        self._graphs = {}      # one graph + arena per STREAM (hip stream handle -> entry): two batches in flight on two streams (bench.py --in-flight 2)
        self.math, self.store = "fp32", "fp32"
        # coordinate convention of the fused affine_grid (and of the functional.grid_sample default the drivers should pass):
        # False = torch >= 1.3 defaults (goldens); True = torch < 1.3, what checkpoints trained on the reference's pinned
        # "pytorch 0.4.0+" expect
        self.align_corners = bool(getattr(opt, "align_corners", 0))
        self.two_queues = None   # None: the process default (PWS_OPT_TWO_QUEUES); True / False: per generator
        # True: the backward adds every weight / bias gradient element with ONE fp32 atomic per launch (PWS_NETG_DETERMINISTIC):
        # bit-identical gradients run to run, at the price of the weight-gradient kernels' parallelism over pixels
        self.deterministic = False
        # True: ``netG(x, False)`` does not compute stage 1's ``up2`` (PWS_NETG_PRUNE_DEAD): its output x122 (reference :171) is read by
        # ``up1`` (:173) and stage 2's ``up_bottom1`` (:196) only, both under ``if is_training`` -- same field bit for bit, 2.3 % fewer flops.
        # Off by default: the reference executes the layer.  (Changing it re-captures an enabled graph.)
        self.prune_dead = False
        if getattr(opt, "math", "fp32") == "bf16":
            self.set_math("bf16")

    def set_math(self, math, store=None):
        """'fp32' (default; exact-fp32 matrix cores, the parity path) or 'bf16' (BASELINE configs 3/4: conv contractions on the
        bf16 matrix cores with fp32 accumulation; master weights, biases, fields and weight gradients stay fp32).
        ``store``: element type of the activations / activation gradients inside the arena -- 'bf16' (default with bf16 math
        when ngf % 32 == 0: half the HBM traffic of every layer) or 'fp32' (operands are rounded while they are staged).
        Applies to the forward and to the backward of forwards run after the call."""
        if math not in ("fp32", "bf16"):
            raise ValueError("UnetGenerator.set_math: expected 'fp32' or 'bf16', got %r" % (math,))
        if store is None:
            store = "bf16" if (math == "bf16" and self.ngf % 32 == 0) else "fp32"
        if store not in ("fp32", "bf16") or (store == "bf16" and (math != "bf16" or self.ngf % 32 != 0)):
            raise ValueError("UnetGenerator.set_math: store=%r needs math='bf16' and ngf %% 32 == 0" % (store,))
        if (math, store) != (self.math, self.store):
            self._graph, self._graphs = None, {}
        self.math, self.store = math, store
        return self

    def _opts(self, math=None, store=None, x_sample_stride=0):
        """The mode of one executor call as the C ABI takes it (pws_netg_opts): carried in the call's arguments, no
        process-wide state is written."""
        math, store = math or self.math, store or self.store
        tq = -1 if self.two_queues is None else int(bool(self.two_queues))
        flags = (A.NETG_DETERMINISTIC if self.deterministic else 0) | (A.NETG_PRUNE_DEAD if getattr(self, "prune_dead", False) else 0)
        return A.PwsNetgOpts(A.MATH_BF16 if math == "bf16" else A.MATH_FP32, A.STORE_BF16 if store == "bf16" else A.STORE_FP32, tq, flags,
                             int(x_sample_stride))

    def enable_graph(self, on=True, alias_output=False, per_stream=False):
        """Opt-in hipGraph replay of the inference forward (``netG(x, False)`` under ``no_grad``): the ~75 launches of a
        forward are captured once per (batch, weight version, mode) and replayed as one graph launch.  The graph owns its
        activation arena and its output buffer; every call returns a fresh tensor (a 0.5 MB/frame device copy) unless
        ``alias_output=True``, in which case the returned field is the graph's own buffer and is OVERWRITTEN by the next call."""
        self._graph_mode = bool(on)
        self._graph_alias = bool(alias_output)
        # per_stream: one graph + arena per calling STREAM instead of one in all, for callers that keep several batches in flight by issuing
        # consecutive calls on different streams (bench.py --in-flight 2).  Default: one graph, replayed on whatever stream the caller is on --
        # safe as long as the caller's calls are ordered (one stream, or streams that wait for each other).  Switching between the two keeps what
        # was captured (the shared graph lives in slot 0, a stream's graph in the slot of its handle: separate arenas).
        self._graph_per_stream = bool(per_stream)
        if not on:
            self._graph, self._graphs = None, {}

    # -- parameters in state-dict order: (weight, bias) per layer
    def _ordered_params(self):
        out = []
        for ls in self._specs:
            block_name, seq_name, _ = ls.name.split(".")
            conv = getattr(getattr(self, block_name), seq_name)[0]
            out += [conv.weight, conv.bias]
        return out

    def _bn_layers(self):
        return [getattr(getattr(self, ls.name.split(".")[0]), ls.name.split(".")[1])[1] for ls in self._specs]

    def _effective_params(self):
        """(weight, bias) per layer as the kernels see them.  ``use_BN`` in eval() mode: BatchNorm2d with running statistics
        is a per-output-channel affine map directly behind the conv (reference lib/networks_cascading.py:253-341), folded
        here:  w' = w * g / sqrt(var + eps),  b' = (b - mean) * g / sqrt(var + eps) + beta."""
        params = self._ordered_params()
        if not self.use_BN:
            return params
        out = []
        with torch.no_grad():
            for i, (ls, bn) in enumerate(zip(self._specs, self._bn_layers())):
                w, b = params[2 * i], params[2 * i + 1]
                s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                shape = (-1, 1, 1, 1) if ls.kind == "conv" else (1, -1, 1, 1)   # OIHW / IOHW: the OUTPUT channel axis
                out += [(w * s.view(shape)).contiguous(), ((b - bn.running_mean) * s + bn.bias).contiguous()]
        return out

    def _weights_key(self):
        ts = list(self._ordered_params())
        if self.use_BN:
            for bn in self._bn_layers():
                ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return tuple((p.data_ptr(), p._version) for p in ts)

    def packed_weights(self, raw=False, train=False):
        """Device buffer with every layer in the kernels' layout; re-packed when a parameter changed.  raw: the conv weights as
        they are (training-mode BatchNorm normalises with batch statistics), not folded with the running statistics.
        train: a backward will follow -- the data-gradient buffer is packed in the same go (pws_netg_pack_weights_train: in bf16 mode
        ONE pass over the weights makes both bf16 copies and skips the fp32 packed copies nobody reads there)."""
        key = (("raw",) + tuple((p.data_ptr(), p._version) for p in self._ordered_params())) if raw else (("eff",) + self._weights_key())
        # the buffer serves ONE math mode (pws_netg_pack_weights_for: a bf16 training loop does not re-make the Winograd copies after
        # every optimizer step, an fp32 one not the bf16 copies); set_math() to the other mode re-packs
        key = (self.math,) + key
        params = None
        if self._packed is None or key != self._packed_key:
            params = self._ordered_params() if raw else self._effective_params()
        if params is not None:
            A.require_cuda(*params)
            dev = params[0].device
            nfl = A.lib().pws_netg_packed_floats(self.input_nc, self.ngf)
            if self._packed is None or self._packed.numel() != nfl or self._packed.device != dev:
                self._packed = torch.empty(nfl, device=dev, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
            math = A.MATH_BF16 if self.math == "bf16" else A.MATH_FP32
            if train and self.math == "bf16" and not self.use_BN:
                dkey = tuple((p.data_ptr(), p._version) for p in params)
                nfd = A.lib().pws_netg_packed_dgrad_floats(self.input_nc, self.ngf)
                if self._packed_dgrad is None or self._packed_dgrad.numel() != nfd or self._packed_dgrad.device != dev:
                    self._packed_dgrad = torch.empty(nfd, device=dev, dtype=torch.float32)
                A.check(A.lib().pws_netg_pack_weights_train(ptrs, A.ptr(self._packed), A.ptr(self._packed_dgrad), self.input_nc, self.ngf, math,
                                                            A.current_stream()), "pws_netg_pack_weights_train")
                self._packed_dgrad_key = dkey
            else:
                A.check(A.lib().pws_netg_pack_weights_for(ptrs, A.ptr(self._packed), self.input_nc, self.ngf, math, A.current_stream()),
                        "pws_netg_pack_weights_for")
            self._packed_key = key
        return self._packed

    def packed_dgrad_weights(self):
        """Second device buffer: the weights in the data-gradient kernels' layout (training only)."""
        params = self._ordered_params()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed_dgrad is None or key != self._packed_dgrad_key:
            A.require_cuda(*params)
            nfl = A.lib().pws_netg_packed_dgrad_floats(self.input_nc, self.ngf)
            if self._packed_dgrad is None or self._packed_dgrad.numel() != nfl or self._packed_dgrad.device != params[0].device:
                self._packed_dgrad = torch.empty(nfl, device=params[0].device, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
            A.check(A.lib().pws_netg_pack_weights_dgrad(ptrs, A.ptr(self._packed_dgrad), self.input_nc, self.ngf,
                                                        A.current_stream()), "pws_netg_pack_weights_dgrad")
            self._packed_dgrad_key = key
        return self._packed_dgrad

    def _grad_layout(self):
        """(first_float, floats) per layer of the gradient slab the backward fills (pws_netg_grad_layout)."""
        if getattr(self, "_grad_layout_cache", None) is None:
            nl = len(self._specs)
            first, count = (ctypes.c_size_t * nl)(), (ctypes.c_size_t * nl)()
            A.check(A.lib().pws_netg_grad_layout(self.input_nc, self.ngf, first, count), "pws_netg_grad_layout")
            self._grad_layout_cache = (list(first), list(count))
        return self._grad_layout_cache

    def _workspace(self, n, is_training, device):
        """The inference arena of the CURRENT stream: kernels of two streams run concurrently, so host threads that share this
        generator on their own streams (the reference's nn.DataParallel runs one thread per replica, lib/networks_cascading.py:51-52)
        must not share an arena.  One arena per stream is kept (a new batch size replaces that stream's arena)."""
        sid = torch.cuda.current_stream(device).cuda_stream
        k = (n, bool(is_training), device)
        ent = self._ws.get(sid)
        if ent is None or ent[0] != k:
            nbytes = A.lib().pws_netg_workspace_bytes(n, self.input_nc, self.ngf, int(is_training))
            ent = (k, torch.empty(nbytes + 256, device=device, dtype=torch.uint8))
            self._ws[sid] = ent
        return ent[1]

    # The training arena (activations + their gradients: 23 GB at 64 samples) is private to a forward until its backward has run.
    # It is handed back to the GENERATOR, not to torch's caching allocator: there a freed 23 GB block is the best fit for whatever
    # is allocated next while it lies free (the optimizer's state on the first step did exactly that), the next forward then finds
    # it split and allocates a second one -- 23 GB more reserved, and a hipMalloc of that size inside a step (1 ms, but 0.5 s when
    # the driver is still reclaiming a previous process's memory: configs[2] timed 110 ms per step instead of 29 in one process of
    # six, round 4).  One arena per stream is kept (work on a stream is ordered, so the next forward there may overwrite it);
    # eval() / release_training_arena() drop it.
    def _take_train_arena(self, nbytes, device):
        sid = torch.cuda.current_stream(device).cuda_stream
        ws = self._train_ws.pop(sid, None)
        if ws is not None and ws.numel() == nbytes and ws.device == device:
            return ws
        del ws
        return torch.empty(nbytes, device=device, dtype=torch.uint8)

    def _give_train_arena(self, ws):
        if ws is not None and self.training:
            self._train_ws[torch.cuda.current_stream(ws.device).cuda_stream] = ws

    def release_training_arena(self):
        self._train_ws.clear()

    def train(self, mode=True):
        if not mode:
            self._train_ws.clear()
        return super().train(mode)

    def forward(self, input1, is_training=True):
        A.require_cuda(input1)
        if input1.dim() != 4 or input1.shape[1] != self.input_nc or input1.shape[2] != 256 or input1.shape[3] != 256:
            raise RuntimeError("UnetGenerator: expected input (N, %d, 256, 256) -- 7 stride-2 levels and the 2x2 flatten "
                               "conv fix the size -- got %s" % (self.input_nc, tuple(input1.shape)))
        needs_grad = torch.is_grad_enabled() and (input1.requires_grad or any(p.requires_grad for p in self.parameters()))
        if self.use_BN and self.training:
            # train() mode: batch statistics, running statistics updated, as nn.BatchNorm2d (the reference's forward always
            # returns the six training outputs here; netG(x, False) in train() mode still normalises with batch statistics)
            # (set_math('bf16'): the conv contractions run on the bf16 matrix cores; activations, BatchNorm statistics and gradients
            # stay fp32 -- the BatchNorm kernels are fp32 -- whatever the storage set_math chose for the BatchNorm-free path)
            from ..autograd import netg_apply_bn
            grids, resid = netg_apply_bn(self, input1)
            return (grids, resid) if is_training else grids[2]
        if self.use_BN and needs_grad:
            raise NotImplementedError(
                "UnetGenerator(use_BN=True).eval(): gradients through the folded (running-statistics) BatchNorm are not "
                "provided -- call .train() to train, or run inference under torch.no_grad()")
        if needs_grad:
            return _netg_autograd(self, input1, is_training)
        if self._graph_mode and not is_training:
            return self._run_graph(input1)
        return self._run(input1, is_training)

    def _capture(self, x, key, static_input):
        """Captures one forward on ``x`` (or on a private dense copy of it: ``static_input``) with an arena private to the graph.
        The caller must hold no reference to the previous graph's entry (it is dropped first; its arena is reused when it fits).
        Captures of different host threads are serialised, preamble included (_capture_lock): on this runtime a device-wide
        synchronisation -- the one below, or the one torch.cuda.graph makes on entry -- from a thread that is about to capture
        invalidates the capture another thread has open (hipErrorStreamCaptureInvalidated / ...Unjoined, seen in
        tests/test_hip_threads.py), whatever the capture mode.  A capture happens once per (batch, mode); eager launches and
        stream-level synchronisation of other threads go on beside it."""
        with _capture_lock:
            xg = x.clone(memory_format=torch.contiguous_format) if static_input else x
            nbytes = A.lib().pws_netg_workspace_bytes(xg.shape[0], self.input_nc, self.ngf, 0)
            sid = self._graph_slot(xg.device)
            old, self._graph = self._graphs.pop(sid, None), None
            ws = old["ws"] if old is not None and old["ws"].numel() == nbytes + 256 and old["ws"].device == xg.device else None
            del old   # the previous hipGraphExec, its output and its input buffer go before the new graph is made
            if ws is None:
                ws = torch.empty(nbytes + 256, device=xg.device, dtype=torch.uint8)
            self._run(xg, False, ws=ws)  # eager warm-up: one-time kernel attribute calls must not happen during capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are held to the capture rules; its own capture stream: torch.cuda.graph's
            # default one is shared by every capture of the process
            with torch.cuda.graph(g, stream=_capture_stream(xg.device), capture_error_mode="thread_local"):
                out = self._run(xg, False, ws=ws)
            # one graph is kept; it holds its input tensor (address baked in), its arena and its output
            self._graph = self._graphs[sid] = dict(key=key, g=g, out=out, x=xg, ws=ws, static=static_input)
            if len(self._graphs) > 8:   # (streams come and go: keep the most recent few)
                self._graphs.pop(next(iter(self._graphs)))
            return self._graph

    def _graph_slot(self, device):
        return torch.cuda.current_stream(device).cuda_stream if getattr(self, "_graph_per_stream", False) else 0

    def _run_graph(self, input1):
        x = input1   # may be a strided view (the video loop's overlapping windows): copied ONCE, straight into the graph's input
        self.packed_weights()
        # the packed buffer is re-packed IN PLACE at a fixed address when a parameter changes (packed_weights() above, outside the
        # graph), so a weight update needs no re-capture: the key holds the buffer's address, not the parameter versions
        key = (tuple(x.shape), str(x.device), self._packed.data_ptr(), self.math, self.store, self.align_corners, self.two_queues, bool(self.prune_dead))
        # one graph (and arena) per stream: a forward replayed on stream B while the previous one still runs on stream A must not share its arena
        ent = self._graphs.get(self._graph_slot(x.device))
        if ent is None or ent["key"] != key:
            ent = None   # (no reference to the old entry while the new graph is captured)
            ent = self._capture(x, key, static_input=not x.is_contiguous())
        elif not x.is_contiguous() or x.data_ptr() != ent["x"].data_ptr():
            # a different input buffer: from now on the graph reads a private static buffer that every call copies into
            if not ent["static"]:
                ent = None
                ent = self._capture(x, key, static_input=True)   # (the capture's copy of x IS this call's input)
            else:
                ent["x"].copy_(x)
        self._graph = ent
        ent["g"].replay()
        return ent["out"] if self._graph_alias else ent["out"].clone()

    def _run(self, input1, is_training, train_ctx=None, ws=None):
        """train_ctx: dict filled with what backward needs; the arena is then private to this call (the reference's
        training loop runs two forwards before one backward, main_new.py:101,112,214).  ws: caller-owned arena (graph capture)."""
        # overlapping sliding windows (stream._windows: an as_strided view whose samples start one plane apart) are read in place
        # by the first layer -- pws_netg_opts.x_sample_stride -- instead of from a gathered copy (inference only)
        x, sstride = input1, 0
        if not x.is_contiguous():
            st = x.stride()
            if (train_ctx is None and not is_training and x.dim() == 4 and st[1:] == (256 * 256, 256, 1) and st[0] > 0 and st[0] % 4 == 0
                    and st[0] < (1 << 30)):
                sstride = st[0]
            else:
                x = x.contiguous()
        n = x.shape[0]
        S = 256
        packed = self.packed_weights(train=train_ctx is not None)
        if train_ctx is not None:
            ws = self._take_train_arena(A.lib().pws_netg_train_workspace_bytes(n, self.input_nc, self.ngf) + 256, x.device)
        elif ws is None:
            ws = self._workspace(n, is_training, x.device)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        ws_bytes = ws.numel() - (ws_ptr - ws.data_ptr())
        ng = 3 if is_training else 1
        grids = torch.empty((ng, n, S, S, 2), device=x.device, dtype=torch.float32)
        resid = torch.empty((3, n, S, S, 2), device=x.device, dtype=torch.float32) if is_training else None
        thetas = torch.empty((3, n, 6), device=x.device, dtype=torch.float32)
        ac = int(self.align_corners)
        opts = self._opts(x_sample_stride=sstride)
        A.check(A.lib().pws_netg_forward_opts(A.ptr(packed), A.ptr(x), n, self.input_nc, self.ngf, int(bool(is_training)),
                                              ac, ctypes.c_void_p(ws_ptr), ws_bytes, A.ptr(grids), A.ptr(resid), A.ptr(thetas),
                                              ctypes.byref(opts), A.current_stream()), "pws_netg_forward_opts")
        self.last_thetas = thetas
        if train_ctx is not None:
            train_ctx.update(x=x, ws=ws, ws_ptr=ws_ptr, ws_bytes=ws_bytes, grids=grids, resid=resid, thetas=thetas, packed=packed,
                             math=self.math, store=self.store, ac=ac, weights_key=self._weights_key())
        if is_training:
            return [grids[0], grids[1], grids[2]], [resid[0], resid[1], resid[2]]
        return grids[0]


_capture_tls = threading.local()
_capture_lock = threading.Lock()


def _capture_stream(device):
    """One graph-capture stream per host thread and device."""
    streams = getattr(_capture_tls, "streams", None)
    if streams is None:
        streams = _capture_tls.streams = {}
    key = str(device)
    if key not in streams:
        streams[key] = torch.cuda.Stream(device)
    return streams[key]


def _netg_autograd(net, input1, is_training):
    """Training entry: forward through the HIP executor with the activations kept in the arena, backward through
    the HIP backward kernels (pwstablenet_amd.autograd)."""
    from ..autograd import netg_apply
    return netg_apply(net, input1, is_training)

"""ctypes binding of the C ABI in include/pwstable.h (libpwstable_hip.so, gfx950).

This is the ONLY way the Python host reaches the kernels, and it passes raw device pointers, sizes and
the current HIP stream -- exactly what any other host language would pass.  There is no CPU fallback:
if the library is missing or the call fails, a RuntimeError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PWS_LIB_PATH") or os.path.join(_HERE, "libpwstable_hip.so")   # PWS_LIB_PATH: A/B builds (tools only)

c_f32p = ctypes.c_void_p  # device pointers are passed as integers
c_stream = ctypes.c_void_p

ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
OPT_TWO_QUEUES = 1
OPT_MATH, MATH_FP32, MATH_BF16 = 2, 0, 1
OPT_STORE, STORE_FP32, STORE_BF16 = 3, 0, 1
ABI_VERSION = 5
NETG_DETERMINISTIC = 1
NETG_PRUNE_DEAD = 2
OPT_EXPERIMENT = 100   # measured kernel variants (tools, per-path tests); 0 = product default
OBJ_SLOTS = 64
CONV_K3S1, CONV_K3S2, CONV_K5S1, CONVT_K3S1, CONVT_K4S2, CONV_K2S1P0, CONV_K1, CONV_K3S1_OUT = range(8)


class PwsSrc(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("channels", ctypes.c_int), ("ld", ctypes.c_int)]


class PwsConvArgs(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int),
                ("nsrc", ctypes.c_int), ("src", PwsSrc * 4), ("src_nchw", ctypes.c_int), ("cout", ctypes.c_int),
                ("w_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("act", ctypes.c_int),
                ("out", ctypes.c_void_p), ("out_ld", ctypes.c_int), ("w_wino", ctypes.c_void_p), ("ws", ctypes.c_void_p),
                ("ws_bytes", ctypes.c_size_t), ("math", ctypes.c_int), ("w_bf16", ctypes.c_void_p), ("store", ctypes.c_int),
                ("w_wring", ctypes.c_void_p), ("out_sign", ctypes.c_void_p), ("out_sign_ld", ctypes.c_int)]


class PwsDst(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("channels", ctypes.c_int), ("ld", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("act_y", ctypes.c_void_p), ("act_y_ld", ctypes.c_int), ("act", ctypes.c_int),
                ("act_sign", ctypes.c_void_p), ("act_sign_ld", ctypes.c_int)]


class PwsConvBwdDataArgs(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int), ("cout", ctypes.c_int),
                ("gout", ctypes.c_void_p), ("gout_ld", ctypes.c_int), ("w_dgrad", ctypes.c_void_p), ("ndst", ctypes.c_int),
                ("dst", PwsDst * 4), ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_size_t), ("math", ctypes.c_int),
                ("w_dgrad_bf16", ctypes.c_void_p), ("store", ctypes.c_int)]


class PwsConvBwdWeightArgs(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int), ("nsrc", ctypes.c_int),
                ("src", PwsSrc * 4), ("src_nchw", ctypes.c_int), ("cout", ctypes.c_int), ("gout", ctypes.c_void_p),
                ("gout_ld", ctypes.c_int), ("dw_packed", ctypes.c_void_p), ("math", ctypes.c_int), ("store", ctypes.c_int),
                ("dbias", ctypes.c_void_p), ("deterministic", ctypes.c_int), ("src2_ptr", ctypes.c_void_p * 4), ("gout2", ctypes.c_void_p)]


class PwsNetgOpts(ctypes.Structure):
    _fields_ = [("math", ctypes.c_int), ("store", ctypes.c_int), ("two_queues", ctypes.c_int), ("flags", ctypes.c_int),
                ("x_sample_stride", ctypes.c_size_t)]


class PwsProfRecord(ctypes.Structure):
    _fields_ = [("kernel_id", ctypes.c_int), ("tag", ctypes.c_int), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double), ("ms", ctypes.c_float)]


_I, _S, _P = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p
_F = ctypes.c_float
# name -> (restype, argtypes); must list every symbol declared in include/pwstable.h
SIGNATURES = {
    "pws_version": (_I, []),
    "pws_last_error": (ctypes.c_char_p, []),
    "pws_set_option": (_I, [_I, _I]),
    "pws_get_option": (_I, [_I]),
    "pws_packed_bf16_floats": (_S, [_I, _I, _I]),
    "pws_pack_weight_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_nchw_to_nhwc_pad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "pws_nchw_to_nhwc_pad_s": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pws_cvt_bf16_to_f32": (_I, [_P, _P, _S, _P]),
    "pws_cvt_f32_to_bf16": (_I, [_P, _P, _S, _I, _P]),
    "pws_device_info": (_I, [ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "pws_packed_weight_floats": (_S, [_I, _I, _I]),
    "pws_pack_conv_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_packed_wino_floats": (_S, [_I, _I]),
    "pws_pack_conv_weight_wino": (_I, [_P, _P, _I, _I, _P]),
    "pws_packed_wring_floats": (_S, [_I, _I, _I]),
    "pws_pack_conv_weight_wring": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_packed_wino_ct4_floats": (_S, [_I, _I]),
    "pws_pack_conv_weight_wino_ct4": (_I, [_P, _P, _I, _I, _P]),
    "pws_conv2d_fwd": (_I, [ctypes.POINTER(PwsConvArgs), _P]),
    "pws_act_bwd_bias": (_I, [_P, _P, _S, _I, _I, _P, _P]),
    "pws_act_bwd_bias_s": (_I, [_P, _P, _S, _I, _I, _P, _I, _P, _S, _P]),
    "pws_act_bwd_bias_ws_bytes": (_S, [_I]),
    "pws_packed_dgrad_floats": (_S, [_I, _I, _I]),
    "pws_pack_conv_weight_dgrad": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_conv2d_bwd_data": (_I, [ctypes.POINTER(PwsConvBwdDataArgs), _P]),
    "pws_conv2d_bwd_weight": (_I, [ctypes.POINTER(PwsConvBwdWeightArgs), _P]),
    "pws_unpack_conv_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_theta_head_fwd_save": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pws_theta_head_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "pws_field_head_bwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "pws_field_head_bwd_s": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P]),
    "pws_field_head_bwd_act": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    "pws_netg_packed_dgrad_floats": (_S, [_I, _I]),
    "pws_netg_pack_weights_dgrad": (_I, [ctypes.POINTER(_P), _P, _I, _I, _P]),
    "pws_netg_train_workspace_bytes": (_S, [_I, _I, _I]),
    "pws_netg_backward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P, _P, _P]),
    "pws_netg_backward_part": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "pws_netg_bn_floats": (_S, [_I, _I]),
    "pws_netg_train_workspace_bytes_bn": (_S, [_I, _I, _I]),
    "pws_netg_forward_bn": (_I, [_P, _P, _P, _F, _F, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P]),
    "pws_netg_backward_bn": (_I, [_P, _P, _P, _F, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P, _P, _P, _P]),
    "pws_netg_forward_bn_opts": (_I, [_P, _P, _P, _F, _F, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, ctypes.POINTER(PwsNetgOpts), _P]),
    "pws_netg_backward_bn_opts": (_I, [_P, _P, _P, _F, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P, _P, _P, ctypes.POINTER(PwsNetgOpts), _P]),
    "pws_netg_unpack_grads": (_I, [_P, ctypes.POINTER(_P), _I, _I, _P]),
    "pws_theta_head_ws_floats": (_S, [_I, _I, _I]),
    "pws_theta_head_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pws_field_head_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "pws_field_head_fwd_s": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P]),
    "pws_affine_grid": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pws_grid_sample_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pws_grid_sample_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pws_upsample_bilinear_ac": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pws_upsample_bilinear_ac_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pws_affine_grid_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pws_upsample_grid_sample_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pws_upsample_grid_sample_u8": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pws_adam_step": (_I, [_P, _P, _P, _P, _S, _F, _F, _F, _F, _I, _P]),
    "pws_adam_step_multi": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _I, _P]),
    "pws_netg_packed_floats": (_S, [_I, _I]),
    "pws_netg_pack_weights": (_I, [ctypes.POINTER(_P), _P, _I, _I, _P]),
    "pws_netg_pack_weights_for": (_I, [ctypes.POINTER(_P), _P, _I, _I, _I, _P]),
    "pws_netg_pack_weights_train": (_I, [ctypes.POINTER(_P), _P, _P, _I, _I, _I, _P]),
    "pws_netg_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "pws_netg_forward": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P]),
    "pws_netg_forward_opts": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _S, _P, _P, _P, ctypes.POINTER(PwsNetgOpts), _P]),
    "pws_netg_backward_opts": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _S, _P, _P, _P, _P, _P, _I, _I, _P, ctypes.POINTER(PwsNetgOpts), _P]),
    "pws_netg_backward_lists": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _S, _P, _P, ctypes.POINTER(_P), ctypes.POINTER(_P), _P, _I, _I, _P,
                                     ctypes.POINTER(PwsNetgOpts), _P]),
    "pws_netg_grad_floats": (_S, [_I, _I]),
    "pws_netg_grad_layout": (_I, [_I, _I, ctypes.POINTER(_S), ctypes.POINTER(_S)]),
    "pws_netg_backward_plan": (_I, [_I, _I, _I, ctypes.POINTER(ctypes.c_ubyte)]),
    "pws_u8_normalize": (_I, [_P, _S, _P, _S, _I, _S, _P]),
    "pws_warp_norm_fwd": (_I, [_P, _S, _P, _P, _P, _S, _P, _I, _I, _I, _P]),
    "pws_warp_norm_bwd": (_I, [_P, _S, _P, _P, _S, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "pws_temporal_l1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "pws_temporal_l1_bwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "pws_temporal_l1_bwd_det": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pws_feature_loss_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "pws_feature_loss_bwd": (_I, [_P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "pws_feature_loss_bwd_det": (_I, [_P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "pws_field_smoothness": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "pws_shape_loss_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "pws_shape_loss_bwd": (_I, [_P, ctypes.c_double, _P, _P, _I, _I, _I, _P]),
    "pws_objective_finalize": (_I, [_P, _I, _P, _I, _P, _P]),
    "pws_maxpool2x2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pws_maxpool2x2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "pws_maxpool2x2_fwd_s": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "pws_maxpool2x2_bwd_s": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pws_sqdiff_sum": (_I, [_P, _P, _S, _P, _P]),
    "pws_sqdiff_bwd": (_I, [_P, _P, _S, _F, _P, _P, _P]),
    "pws_gray_area_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pws_area_half_u8": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pws_area_resize_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pws_bn_ws_bytes": (_S, [_I]),
    "pws_bn_train_fwd": (_I, [_P, _S, _I, _P, _P, _I, _P, _P, _P, _P, _F, _F, _I, _P, _S, _P]),
    "pws_bn_train_bwd": (_I, [_P, _P, _P, _P, _P, _I, _S, _I, _P, _P, _P, _S, _P]),
    "pws_prof_enable": (_I, [_I]),
    "pws_prof_collect": (_I, [ctypes.POINTER(PwsProfRecord), _I]),
    "pws_prof_kernel_name": (ctypes.c_char_p, [_I]),
}

_lib = None


def lib():
    """Loads libpwstable_hip.so (built by ``python -m pwstablenet_amd.build``).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pwstablenet_amd: %s is missing -- build it with `python -m pwstablenet_amd.build` "
                "(there is no CPU fallback for the HIP hot path)" % LIB_PATH)
        # torch bundles its own HIP runtime (libamdhip64.so.7); it must be in the process BEFORE this library is
        # opened so that both share ONE runtime -- two runtimes in one process cannot both own the device
        # ("no ROCm-capable device is detected" from whichever comes second).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.pws_version() != ABI_VERSION:
            raise RuntimeError("pwstablenet_amd: ABI version mismatch (%d)" % L.pws_version())
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("pwstable HIP call %s failed (rc=%d): %s" % (what, rc, lib().pws_last_error().decode()))


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cu_masked_streams(shares, device=None):
    """torch streams that may only use DISJOINT sets of the device's compute units: ``shares`` = CUs per stream, dealt in order
    (``hipExtStreamCreateWithCUMask``).  For work of several tenants on one GPU: while a kernel that leaves room on its CUs executes
    ``v_mfma_f32_32x32x16_bf16`` (the first-generation bf16 kernels of this library), kernels of OTHER streams on the SAME CUs compute
    other values in a few lanes; on disjoint CUs they do not (tools/probes/kernel_victim_probe.py PROBE_CU_MASK).
    A generator run on such a stream should keep to it: ``net.module.two_queues = False`` (the side queue is not masked).
    The streams live until the process ends (torch.cuda.ExternalStream does not own them), and every one of them takes a hardware queue
    of its own: the same ``shares`` give the same streams again (a few hundred of them made in a loop crashed the runtime)."""
    import torch
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (dev, tuple(int(n) for n in shares))
    if key in _cu_streams:
        return list(_cu_streams[key])
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if sum(shares) > ncu or min(shares) < 1:
        raise ValueError("cu_masked_streams: %r does not fit %d compute units" % (list(shares), ncu))
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ncu + 31) // 32
    out, first = [], 0
    with torch.cuda.device(dev):
        for n in shares:
            bits = ((1 << n) - 1) << first
            mask = (ctypes.c_uint32 * words)(*[(bits >> (32 * w)) & 0xffffffff for w in range(words)])
            st = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
            if rc != 0 or not st.value:
                raise RuntimeError("hipExtStreamCreateWithCUMask failed (rc=%d)" % rc)
            out.append(torch.cuda.ExternalStream(st.value, device=dev))
            first += n
    _cu_streams[key] = tuple(out)
    return out


_cu_streams = {}


def require_cuda(*tensors, dtype=None):
    """The hot path runs on the GPU only; refuse anything else loudly.  dtype: expected dtype (default torch.float32)."""
    import torch
    want = dtype or torch.float32
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("pwstablenet_amd: the HIP hot path needs CUDA/HIP tensors (got %s); there is no CPU "
                               "fallback -- move the model and inputs to the GPU" %
                               (t.device if isinstance(t, torch.Tensor) else type(t)))
        if t.dtype != want:
            raise RuntimeError("pwstablenet_amd: %s tensors expected, got %s" % (want, t.dtype))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def prof_collect(max_records=4096):
    """Returns [(kernel_name, tag, flops, bytes, ms)] of the launches recorded since pws_prof_enable(1)."""
    buf = (PwsProfRecord * max_records)()
    n = lib().pws_prof_collect(buf, max_records)
    if n < 0:
        check(n, "pws_prof_collect")
    L = lib()
    return [(L.pws_prof_kernel_name(r.kernel_id).decode(), r.tag, r.flops, r.bytes, r.ms) for r in buf[:min(n, max_records)]]

"""ctypes binding of oracle/pws_oracle.c (TEST INFRASTRUCTURE ONLY -- see the C file's header).

Every function takes / returns contiguous float32 numpy arrays.  Activations are NCHW, warp fields are
N,H,W,2, exactly as in the reference (lib/networks_cascading.py, main_new.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpws_oracle.so")

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
PROBE_NAMES = ["x11", "x14", "x18", "x177", "x122", "x22", "x25", "x28", "x277", "x222", "x38", "x322"]

_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the C oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "pws_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_version.restype = ctypes.c_int
        _lib.orc_netg_forward.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def conv2d(x, w, b, stride, pad, act=ACT_NONE):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d(_p(x), _p(w), _p(b), _p(out), N, Cin, H, W, Cout, k, stride, pad, act)
    return out


def conv_transpose2d(x, w, b, stride, pad, act=ACT_NONE):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, Cin, H, W = x.shape
    _, Cout, k, _ = w.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv_transpose2d(_p(x), _p(w), _p(b), _p(out), N, Cin, H, W, Cout, k, stride, pad, act)
    return out


def affine_grid(theta, H, W, align_corners=False):
    theta = _c(theta).reshape(-1, 6)
    N = theta.shape[0]
    out = np.empty((N, H, W, 2), np.float32)
    lib().orc_affine_grid(_p(theta), _p(out), N, H, W, int(align_corners))
    return out


def grid_sample_fwd(inp, grid, align_corners=False):
    inp, grid = _c(inp), _c(grid)
    N, C, H, W = inp.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_grid_sample_fwd(_p(inp), _p(grid), _p(out), N, C, H, W, Ho, Wo, int(align_corners))
    return out


def grid_sample_bwd(gout, inp, grid, align_corners=False, want_input=True, want_grid=True):
    gout, inp, grid = _c(gout), _c(inp), _c(grid)
    N, C, H, W = inp.shape
    _, Ho, Wo, _ = grid.shape
    gi = np.empty_like(inp) if want_input else None
    gg = np.empty_like(grid) if want_grid else None
    lib().orc_grid_sample_bwd(_p(gout), _p(inp), _p(grid), _p(gi), _p(gg), N, C, H, W, Ho, Wo,
                              int(align_corners))
    return gi, gg


def upsample_bilinear_ac(x, Ho, Wo):
    x = _c(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_upsample_bilinear_ac(_p(x), _p(out), N, C, H, W, Ho, Wo)
    return out


def adam_step(p, g, m, v, lr, b1, b2, eps, step):
    """In place on p, m, v (float32 contiguous)."""
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    lib().orc_adam_step(_p(p), _p(g), _p(m), _p(v), ctypes.c_size_t(p.size), ctypes.c_float(lr),
                        ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(eps), int(step))


class _Probe(ctypes.Structure):
    _fields_ = [("sum", ctypes.c_double), ("abssum", ctypes.c_double)]


def netg_forward(params, x, is_training=True, ngf=64, align_corners=False, probes=False):
    """params: list of 92 float32 arrays in state-dict order (torch layouts).

    Returns dict(grids=[...], resid=[...]|None, thetas=(3,N,6), probes={name:(sum,abssum)}|None);
    ``grids`` has 3 entries when is_training else 1 (stage 3), each N,S,S,2.
    """
    x = _c(x)
    N, input_nc, S, _ = x.shape
    ps = [_c(p) for p in params]
    assert len(ps) == 92
    arr = (_f32p * 92)(*[_p(p) for p in ps])
    ng = 3 if is_training else 1
    grids = np.empty((ng, N, S, S, 2), np.float32)
    resid = np.empty((3, N, S, S, 2), np.float32) if is_training else None
    thetas = np.empty((3, N, 6), np.float32)
    pr = (_Probe * len(PROBE_NAMES))() if probes else None
    rc = lib().orc_netg_forward(arr, _p(x), N, input_nc, ngf, S, int(is_training), int(align_corners),
                                _p(grids), _p(resid), _p(thetas), pr)
    if rc != 0:
        raise ValueError("orc_netg_forward failed rc=%d (input must be N x C x 256 x 256)" % rc)
    return {
        "grids": [grids[i] for i in range(ng)],
        "resid": [resid[i] for i in range(3)] if is_training else None,
        "thetas": thetas,
        "probes": {n: (pr[i].sum, pr[i].abssum) for i, n in enumerate(PROBE_NAMES)} if probes else None,
    }

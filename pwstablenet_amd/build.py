"""Builds libpwstable_hip.so (gfx950 only) in-tree with hipcc.  `python -m pwstablenet_amd.build [--force]`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpwstable_hip.so")
SOURCES = ["abi.cpp", "netg.cpp", "pack.hip", "conv_mfma.hip", "conv_bf16.hip", "conv_wgrad.hip", "wgrad_bf16.hip", "conv_wino.hip", "head.hip", "head_bwd.hip", "grid_sample.hip",
           "adam.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _deps():
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(HERE, "..", "include", "pwstable.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src + ".o")
    if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), _deps()):
        return o, False
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
    subprocess.check_call(cmd)
    return o, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

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
SOURCES = ["abi.cpp", "netg.cpp", "netg_pack.hip", "pack.hip", "conv_mfma.hip", "conv_bf16.hip", "conv_ring.hip", "conv_ring_f32.hip", "conv_first.hip", "conv_first_wino.hip", "conv_wgrad.hip", "wgrad_bf16.hip", "wgrad_ring.hip", "conv_wino.hip", "conv_wring.hip", "conv_skinny.hip", "conv_skinny16.hip", "head.hip", "head_bwd.hip", "grid_sample.hip",
           "adam.hip", "objective.hip", "pool.hip", "frameio.hip", "bnorm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers the library is built from: profile summaries under
    profiles/ carry it (`_meta.source_hash`), and bench.py only quotes counter-derived figures whose hash matches the build."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(SOURCES + ["common.h", "conv_common.h", "netg_pack.h"])
    for f in files + [os.path.join("..", "..", "include", "pwstable.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


def _deps():
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "netg_pack.h"), os.path.join(HERE, "..", "include", "pwstable.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(h) for h in hdrs)


def _resources(stderr_text):
    """Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks: {kernel: {vgprs, scratch, occupancy, lds}}."""
    import re
    out, cur = {}, None
    for line in stderr_text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def _compile(src, force):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src + ".o")
    if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), _deps()):
        return o, False
    hip = src.endswith(".hip")
    cmd = [HIPCC] + FLAGS + (["-x", "hip", "-Rpass-analysis=kernel-resource-usage"] if hip else []) + ["-c", s, "-o", o]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if hip:
        # per-kernel register / scratch / occupancy report next to the object (tests/test_boundary.py reads these:
        # a refactor that pushes accumulators into scratch must not go unnoticed)
        import json
        with open(o + ".resources.json", "w") as f:
            json.dump(_resources(r.stderr), f, indent=1, sort_keys=True)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" not in ln and not ln.lstrip().startswith(("|", "^"))
                        and not ln.strip()[:1].isdigit())
    else:
        err = r.stderr
    if err.strip():
        sys.stderr.write(err + "\n")
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    return o, True


def kernel_resources():
    """{source: {kernel: {...}}} from the last build of each HIP source."""
    import json
    out = {}
    for src in SOURCES:
        f = os.path.join(OBJ, src + ".o.resources.json")
        if os.path.exists(f):
            with open(f) as fh:
                out[src] = json.load(fh)
    return out


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

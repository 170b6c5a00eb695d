#!/usr/bin/env python3
"""Which objective kernel gives run-to-run different results while bf16 inference of another generator runs on another stream?
Each candidate is launched over and over on FIXED inputs (its own stream); every output is compared with the first.
usage: python tools/probes/kernel_victim_probe.py [seconds per candidate] [aggressor math: bf16|fp32]"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_hip_threads as T  # noqa: E402
from pwstablenet_amd import hipabi as A  # noqa: E402
from pwstablenet_amd import synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
amath = sys.argv[2] if len(sys.argv) > 2 else "bf16"
L = A.lib()
L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
m, n, h, w = 4, 2, 256, 256
g = torch.Generator(device="cuda").manual_seed(3)
base = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 256, device="cuda"), torch.linspace(-1, 1, 256, device="cuda"), indexing="ij")[::-1], -1)
grid = (base[None] + 0.05 * torch.randn(m, h, w, 2, device="cuda", generator=g)).contiguous()
rgb = torch.rand(m, 3, h, w, device="cuda", generator=g) * 2 - 1
stable = torch.rand(m, 3, h, w, device="cuda", generator=g) * 2 - 1
gextra0 = torch.randn(m, 3, h, w, device="cuda", generator=g) * 1e-3
fake = torch.rand(m, 3, h, w, device="cuda", generator=g) * 2 - 1
theta = torch.tensor([[1.01, 0.02, 0.01, -0.015, 0.99, 0.02]] * n, device="cuda")
features = torch.rand(m, 400, 6, device="cuda", generator=g) * 1.8 - 0.9
scale = torch.ones(1, device="cuda")
stop, started = threading.Event(), threading.Event()


def masked_stream(which):
    """PROBE_CU_MASK=1: victim and aggressor on DISJOINT halves of the chip's CUs (hipExtStreamCreateWithCUMask): is the effect local to a CU?
    PROBE_CU_MASK=2: both on the SAME half (control)."""
    mode = os.environ.get("PROBE_CU_MASK")
    if not mode:
        return torch.cuda.Stream()
    hip = ctypes.CDLL("libamdhip64.so")
    words = 8   # 256 CUs
    lo = (ctypes.c_uint32 * words)(*([0xffffffff] * 4 + [0] * 4))
    hi = (ctypes.c_uint32 * words)(*([0] * 4 + [0xffffffff] * 4))
    mask = lo if (which == "victim" or mode == "2") else hi
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0 and st.value, rc
    return torch.cuda.ExternalStream(st.value)


def cand_warp(st):
    gg = torch.empty_like(grid)
    A.check(L.pws_warp_norm_bwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), A.ptr(stable), 3 * h * w, 1e-6, A.ptr(scale), A.ptr(gextra0), A.ptr(gg), 0, m, h, w, st), "w")
    return gg


big_scale = torch.ones(1 << 20, device="cuda")


def cand_warp_noscale(st):
    gg = torch.empty_like(grid)
    A.check(L.pws_warp_norm_bwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), A.ptr(stable), 3 * h * w, 1e-6, None, A.ptr(gextra0), A.ptr(gg), 0, m, h, w, st), "w")
    return gg


def cand_warp_bigscale(st):
    gg = torch.empty_like(grid)
    A.check(L.pws_warp_norm_bwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), A.ptr(stable), 3 * h * w, 1e-6, A.ptr(big_scale[4096:4097]), A.ptr(gextra0), A.ptr(gg), 0, m, h, w, st), "w")
    return gg


def cand_warp_noextra(st):
    gg = torch.empty_like(grid)
    A.check(L.pws_warp_norm_bwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), A.ptr(stable), 3 * h * w, 1e-6, A.ptr(scale), None, A.ptr(gg), 0, m, h, w, st), "w")
    return gg


def cand_warp_notarget(st):
    gg = torch.empty_like(grid)
    A.check(L.pws_warp_norm_bwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), None, 0, 1e-6, A.ptr(scale), A.ptr(gextra0), A.ptr(gg), 0, m, h, w, st), "w")
    return gg


def cand_temporal(st):
    ge = gextra0.clone()
    scratch = torch.empty((n, 3, h, w), device="cuda")
    A.check(L.pws_temporal_l1_bwd_det(A.ptr(fake[:n]), A.ptr(fake[n:]), A.ptr(theta), 1e-5, A.ptr(scale), A.ptr(ge[:n]), A.ptr(ge[n:]), A.ptr(scratch), n, h, w, st), "t")
    return ge


def cand_feature(st):
    gg = torch.zeros_like(grid)
    A.check(L.pws_feature_loss_bwd_det(A.ptr(grid), A.ptr(features), 1e-3, A.ptr(scale), A.ptr(gg), m, 400, h, w, st), "f")
    return gg


from pwstablenet_amd import functional as PF  # noqa: E402
frames256 = torch.rand(8, 3, 256, 256, device="cuda", generator=g) * 2 - 1
grid8 = (base[None] + 0.05 * torch.randn(8, h, w, 2, device="cuda", generator=g)).contiguous()
frames720 = torch.randint(0, 256, (4, 720, 1280, 3), device="cuda", dtype=torch.uint8, generator=g)


def cand_grid_sample(st):   # the inference output: F.grid_sample of 256 x 256 frames
    return PF.grid_sample(frames256, grid8)


def cand_warp720_u8(st):    # ... and the fused 720p uint8 warp of the video loop
    return PF.upsample_grid_sample_u8(frames720, grid8[:4])


def cand_clone(st):
    return (grid.clone() * 1.5).contiguous()


def cand_warp_fwd(st):
    out = torch.empty(m, 3, h, w, device="cuda")
    slots = torch.zeros(A.OBJ_SLOTS, device="cuda", dtype=torch.float64)
    A.check(L.pws_warp_norm_fwd(A.ptr(rgb), 3 * h * w, A.ptr(grid), A.ptr(out), A.ptr(stable), 3 * h * w, A.ptr(slots), m, h, w, st), "wf")
    return torch.cat([out.reshape(-1), slots.float()])


CLEAN = {}


def victim():
    torch.cuda.set_device(0)
    s = masked_stream("victim")
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.default_stream())
        if os.environ.get("PROBE_DETAIL") == "2":
            # round 5: WHICH of the deterministic temporal backward's three outputs differs (gfake1 and the staged s values are written by
            # temporal_l1_kernel<true>, gfake2 by temporal_gather_kernel from the staged values), does a second read of the same memory
            # still differ (wrong data vs a wrong read), and what do the wrong values look like
            def run_t():
                ge = gextra0.clone()
                scratch = torch.full((n, 3, h, w), float("nan"), device="cuda")
                A.check(L.pws_temporal_l1_bwd_det(A.ptr(fake[:n]), A.ptr(fake[n:]), A.ptr(theta), 1e-5, A.ptr(scale), A.ptr(ge[:n]), A.ptr(ge[n:]),
                                                  A.ptr(scratch), n, h, w, A.current_stream()), "t")
                return ge, scratch
            ref_ge, ref_sc = run_t()
            ref_ge, ref_sc = ref_ge.clone(), ref_sc.clone()
            s.synchronize()
            started.wait(120)
            shown = tot = 0
            kinds = {"gfake1": 0, "staged": 0, "gfake2": 0, "second read equal": 0}
            for it in range(int(os.environ.get("PROBE_LAUNCHES", "20000"))):
                ge, sc = run_t()
                d1, ds, d2 = (ge[:n] != ref_ge[:n]), (sc != ref_sc), (ge[n:] != ref_ge[n:])
                n1, ns, n2 = int(d1.sum()), int(ds.sum()), int(d2.sum())
                if n1 or ns or n2:
                    tot += 1
                    kinds["gfake1"] += 1 if n1 else 0
                    kinds["staged"] += 1 if ns else 0
                    kinds["gfake2"] += 1 if n2 else 0
                    s.synchronize()
                    again = int((ge != ref_ge).sum()) + int((sc != ref_sc).sum())
                    kinds["second read equal"] += 1 if again == 0 else 0
                    if shown < 8:
                        shown += 1
                        which = d2 if n2 else (ds if ns else d1)
                        cur = ge[n:] if n2 else (sc if ns else ge[:n])
                        ref = ref_ge[n:] if n2 else (ref_sc if ns else ref_ge[:n])
                        idx = which.reshape(-1).nonzero().reshape(-1)
                        rows = torch.unique((idx % (h * w)) // w)
                        print("launch %d: gfake1 %d, staged %d, gfake2 %d elements differ (second read: %d); rows %s; got %s want %s" % (
                            it, n1, ns, n2, again, rows[:8].tolist(), ["%.6g" % v for v in cur.reshape(-1)[idx[:5]].tolist()],
                            ["%.6g" % v for v in ref.reshape(-1)[idx[:5]].tolist()]), flush=True)
            print("temporal_l1_bwd_det: %d launches with wrong elements: %s" % (tot, kinds), flush=True)
            stop.set()
            return
        if os.environ.get("PROBE_DETAIL") == "1":
            clean = cand_warp(A.current_stream()).clone()   # before the aggressor starts
            s.synchronize()
            started.wait(120)
            shown = 0
            tot = 0
            for it in range(int(os.environ.get("PROBE_LAUNCHES", "20000"))):
                cur = cand_warp(A.current_stream())
                d = (cur != clean).reshape(-1)
                nd = int(d.sum())
                tot += 1 if nd else 0
                if nd and shown < 6:
                    shown += 1
                    idx = d.nonzero().reshape(-1)
                    blocks = torch.unique(idx // 2048)   # a workgroup writes 256 lanes x 4 pixels x 2 floats
                    waves = torch.unique(idx // 512)
                    c, r = cur.reshape(-1)[idx[:6]].tolist(), clean.reshape(-1)[idx[:6]].tolist()
                    print("launch %d: %d elements differ, in %d workgroups %s, %d waves; first values got %s want %s" % (
                        it, nd, blocks.numel(), blocks[:8].tolist(), waves.numel(), ["%.4g" % v for v in c], ["%.4g" % v for v in r]), flush=True)
            print("warp_norm_bwd: %d launches with wrong elements [PWS_EXPERIMENT %s]" % (tot, os.environ.get("PWS_EXPERIMENT", "0")), flush=True)
            stop.set()
            return
        started.wait(120)
        for name, fn in (("warp_norm_bwd", cand_warp), ("warp_norm_bwd, scale NULL", cand_warp_noscale), ("warp_norm_bwd, scale in 4 MB", cand_warp_bigscale),
                         ("warp_norm_bwd, no gextra", cand_warp_noextra), ("warp_norm_bwd, no target", cand_warp_notarget), ("temporal_l1_bwd_det", cand_temporal), ("feature_loss_bwd_det", cand_feature),
                         ("warp_norm_fwd + slots", cand_warp_fwd), ("grid_sample 256^2 (inference output)", cand_grid_sample),
                         ("upsample_grid_sample_u8 720p", cand_warp720_u8), ("torch clone * 1.5", cand_clone)):
            ref = fn(A.current_stream()).clone()
            s.synchronize()
            t_end, it, bad, worst = time.perf_counter() + secs, 0, 0, 0
            while time.perf_counter() < t_end:
                cur = fn(A.current_stream())
                if not torch.equal(cur, ref):
                    bad += 1
                    worst = max(worst, int((cur != ref).sum()))
                it += 1
            print("%-38s %5d of %6d launches differ from the first (at most %d elements)" % (name, bad, it, worst), flush=True)
    stop.set()


def aggressor():
    torch.cuda.set_device(0)
    s = masked_stream("aggressor")
    if amath.startswith("poison"):
        # round 5: NO matrix instruction, NO memory traffic -- waves that fill their registers (512 per lane: "poison"; 128: "poison128";
        # "poisonlds": + 64 KB of LDS; "poisonhold": sleep a while first) with a NaN pattern and exit, back to back.  A victim that reads a
        # register / LDS word it never wrote now reads NaN.  tools/probes/reg_poison.hip, built into tools/_bin/reg_poison.so.
        so = os.path.join(ROOT, "tools", "_bin", "reg_poison.so")
        if not os.path.exists(so):
            import subprocess
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                                   os.path.join(ROOT, "tools", "probes", "reg_poison.hip"), "-o", so])
        P = ctypes.CDLL(so)
        P.reg_poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lds_too = 128 if amath == "poison128" else (1 if amath == "poisonlds" else 0)
        hold = 64 if amath in ("poisonhold", "poison128") else 0
        nblk = int(os.environ.get("PROBE_POISON_BLOCKS", "1024"))
        with torch.cuda.stream(s):
            stq = A.current_stream()
            k = 0
            while not stop.is_set():
                rc = P.reg_poison_launch(stq, nblk, lds_too, hold)
                assert rc == 0, rc
                k += 1
                if k == 4:
                    started.set()
                if k % 64 == 0:
                    s.synchronize()
        print("poison launches:", k, flush=True)
        return
    if amath.startswith("gemm"):   # NOT this library: torch matmuls (hipBLASLt / rocBLAS) in bf16 or fp32, back to back
        dt = torch.bfloat16 if amath == "gemm_bf16" else torch.float32
        with torch.cuda.stream(s):
            a_ = torch.randn(8192, 8192, device="cuda", dtype=dt)
            b_ = torch.randn(8192, 8192, device="cuda", dtype=dt)
            k = 0
            while not stop.is_set():
                c_ = a_ @ b_
                k += 1
                if k == 4:
                    started.set()
                if k % 8 == 0:
                    s.synchronize()
        return
    if amath.startswith("conv"):   # ONE bf16 (or fp32: "convf32") 3x3 layer of this library, 128 -> 128 channels on 128 x 128 maps x 8, back to back
        bf16 = amath != "convf32"
        with torch.cuda.stream(s):
            stq = A.current_stream()
            cin = cout = 128
            kind = A.CONV_K3S1
            wt = torch.randn((cout, cin, 3, 3), device="cuda") / (cin * 3) ** 0.5
            wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
            A.check(L.pws_pack_conv_weight(A.ptr(wt), A.ptr(wp), kind, cin, cout, stq), "pack")
            x = torch.randn((8, 128, 128, cin), device="cuda")
            out = torch.empty((8, 128, 128, cout), device="cuda")
            bb = torch.randn(cout, device="cuda")
            ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
            a = A.PwsConvArgs()
            a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, 8, 128, 128, 1, cout, 1
            a.src[0].ptr, a.src[0].channels, a.src[0].ld = x.data_ptr(), cin, cin
            a.w_packed, a.bias, a.out, a.out_ld, a.ws, a.ws_bytes = wp.data_ptr(), bb.data_ptr(), out.data_ptr(), cout, ws.data_ptr(), ws.numel()
            if bf16:
                if os.environ.get("PROBE_STORE_FP32") != "1":   # (bf16 math on fp32 tensors otherwise)
                    x, out = x.bfloat16(), out.bfloat16()
                    a.src[0].ptr, a.out, a.store = x.data_ptr(), out.data_ptr(), 1
                wb = torch.empty(L.pws_packed_bf16_floats(9, cin, cout), device="cuda")
                A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), 9, cin, cout, stq), "bf16 pack")
                a.math, a.w_bf16 = A.MATH_BF16, wb.data_ptr()
            else:
                ww = torch.empty(L.pws_packed_wino_floats(cin, cout), device="cuda")
                A.check(L.pws_pack_conv_weight_wino(A.ptr(wp), A.ptr(ww), cin, cout, stq), "wino pack")
                a.w_wino = ww.data_ptr()
                if L.pws_packed_wring_floats(kind, cin, cout):
                    wr = torch.empty(L.pws_packed_wring_floats(kind, cin, cout), device="cuda")
                    A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), kind, cin, cout, stq), "wring pack")
                    a.w_wring = wr.data_ptr()
            k = 0
            L.pws_prof_enable(1)
            A.check(L.pws_conv2d_fwd(ctypes.byref(a), stq), "conv")
            L.pws_prof_enable(0)
            print("aggressor kernel:", [r[0] for r in A.prof_collect()], flush=True)
            while not stop.is_set():
                A.check(L.pws_conv2d_fwd(ctypes.byref(a), stq), "conv")
                k += 1
                if k == 4:
                    started.set()
                if k % 32 == 0:
                    s.synchronize()
        return
    with torch.cuda.stream(s):
        net = T.make_net("W1", 5)
        net.module.set_math(amath)
        xs = [torch.from_numpy(synth.noise_window(4, 31, 256, seed=33 + r)).cuda() for r in range(2)]
        k = 0
        with torch.no_grad():
            while not stop.is_set():
                net(xs[k & 1], False)
                k += 1
                if k == 4:
                    started.set()
                if k % 16 == 0:
                    s.synchronize()


if os.environ.get("PROBE_AGGRESSOR_ONLY"):   # this process only keeps the GPU busy (the victim runs in ANOTHER process: separate address spaces)
    t = threading.Thread(target=aggressor)
    t.start()
    time.sleep(float(os.environ["PROBE_AGGRESSOR_ONLY"]))
    stop.set()
    t.join()
    sys.exit(0)
if amath == "none":
    started.set()
    ths = [threading.Thread(target=victim)]
else:
    ths = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print("aggressor: %s inference" % amath)

#!/usr/bin/env python3
"""Single-layer conv micro-benchmark through pws_conv2d_fwd: python tools/conv_bench.py KIND N H W CIN COUT [wino|bf16]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwstablenet_amd import hipabi as A  # noqa: E402


def bench_dgrad(kname, n, h, w, cin, cout):
    """Data gradient of the layer (forward input n x h x w x cin, forward cout) with bf16 math and storage, one destination."""
    L, st = A.lib(), A.current_stream()
    L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
    kind = getattr(A, kname)
    k = {"CONV_K3S1": 3, "CONV_K3S2": 3, "CONVT_K3S1": 3, "CONVT_K4S2": 4}[kname]
    wt = torch.randn((cout, cin, k, k) if not kname.startswith("CONVT") else (cin, cout, k, k), device="cuda") / (cin * k) ** 0.5
    wdg = torch.empty(L.pws_packed_dgrad_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight_dgrad(A.ptr(wt), A.ptr(wdg), kind, cin, cout, st), "pack_dgrad")
    planes = 9 if "S1" in kname else 16
    wdb = torch.empty(L.pws_packed_bf16_floats(planes, cout, cin), device="cuda")
    A.check(L.pws_pack_weight_bf16(A.ptr(wdg), A.ptr(wdb), planes, cout, cin, st), "pack_bf16")
    oh, ow = (h, w) if "S1" in kname else ((h // 2, w // 2) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    rot = int(os.environ.get("CONV_BENCH_ROTATE", "1"))
    dys = [torch.randn((n, oh, ow, cout), device="cuda").bfloat16() for _ in range(rot)]
    dxs = [torch.empty((n, h, w, cin), device="cuda", dtype=torch.bfloat16) for _ in range(rot)]
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    da = A.PwsConvBwdDataArgs()
    da.kind, da.n, da.h, da.w, da.cout = kind, n, h, w, cout
    da.gout_ld, da.w_dgrad, da.ndst = cout, wdg.data_ptr(), 1
    da.math, da.w_dgrad_bf16, da.store = A.MATH_BF16, wdb.data_ptr(), A.STORE_BF16
    da.dst[0].channels, da.dst[0].ld, da.dst[0].accumulate = cin, cin, int(os.environ.get("CONV_BENCH_ACC", "0"))   # 1: read-modify-write of dx (a second consumer's gradient)
    da.ws, da.ws_bytes = ws.data_ptr(), ws.numel()

    def launch(i):
        da.gout, da.dst[0].ptr = dys[i % rot].data_ptr(), dxs[i % rot].data_ptr()
        A.check(L.pws_conv2d_bwd_data(ctypes.byref(da), st), "bwd_data")
    for i in range(3):
        launch(i)
    torch.cuda.synchronize()
    L.pws_prof_enable(1)
    for i in range(12):
        launch(i)
    L.pws_prof_enable(0)
    r = A.prof_collect()
    ms = sorted(x_[4] for x_ in r)[len(r) // 2]
    print("%-11s n=%d %dx%d %d->%d dgrad exp=%s %-18s: %8.1f us  %6.1f TFLOP/s (algorithmic)" % (
        kname, n, h, w, cin, cout, os.environ.get("PWS_EXPERIMENT", "0"), r[0][0], ms * 1e3, r[0][2] / (ms * 1e-3) / 1e12))


def bench(kname, n, h, w, cin, cout, wino, bf16=False):
    L, st = A.lib(), A.current_stream()
    L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
    kind = getattr(A, kname)
    k = {"CONV_K3S1": 3, "CONV_K3S2": 3, "CONVT_K3S1": 3, "CONVT_K4S2": 4, "CONV_K5S1": 5}[kname]
    wt = torch.randn((cout, cin, k, k) if not kname.startswith("CONVT") else (cin, cout, k, k), device="cuda") / (cin * k) ** 0.5
    wp = torch.empty(L.pws_packed_weight_floats(kind, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight(A.ptr(wt), A.ptr(wp), kind, cin, cout, st), "pack")
    x = torch.randn((n, h, w, cin), device="cuda")
    oh, ow = (h, w) if "S1" in kname else ((h // 2, w // 2) if kname == "CONV_K3S2" else (2 * h, 2 * w))
    out = torch.empty((n, oh, ow, cout), device="cuda")
    b = torch.randn(cout, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act = kind, n, h, w, 1, cout, 1
    a.src[0].ptr, a.src[0].channels, a.src[0].ld = x.data_ptr(), cin, cin
    a.w_packed, a.bias, a.out, a.out_ld, a.ws, a.ws_bytes = wp.data_ptr(), b.data_ptr(), out.data_ptr(), cout, ws.data_ptr(), ws.numel()
    if wino and kname == "CONVT_K4S2":
        ww = torch.empty(L.pws_packed_wino_ct4_floats(cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wino_ct4(A.ptr(wp), A.ptr(ww), cin, cout, st), "wino pack")
        a.w_wino = ww.data_ptr()
    elif wino:
        ww = torch.empty(L.pws_packed_wino_floats(cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wino(A.ptr(wp), A.ptr(ww), cin, cout, st), "wino pack")
        a.w_wino = ww.data_ptr()
    if wino and L.pws_packed_wring_floats(kind, cin, cout):
        wr = torch.empty(L.pws_packed_wring_floats(kind, cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), kind, cin, cout, st), "wring pack")
        a.w_wring = wr.data_ptr()
    store16 = bool(int(os.environ.get("CONV_BENCH_STORE16", "0")))
    if bf16 and store16:   # bf16 activation storage: sources and output hold bf16 elements
        x = x.bfloat16()
        out = out.bfloat16()
        a.src[0].ptr, a.out, a.store = x.data_ptr(), out.data_ptr(), 1
    if bf16:
        planes = 16 if kname == "CONVT_K4S2" else k * k
        cin_pad = (cin + 15) // 16 * 16
        wb = torch.empty(L.pws_packed_bf16_floats(planes, cin_pad, cout), device="cuda")
        A.check(L.pws_pack_weight_bf16(A.ptr(wp), A.ptr(wb), planes, cin_pad, cout, st), "bf16 pack")
        a.math, a.w_bf16 = A.MATH_BF16, wb.data_ptr()
    # CONV_BENCH_ROTATE=k: cycle over k input/output buffer pairs so that repeated launches do not find their
    # activations in the 256 MB MALL / L2 (what a layer sees inside the network)
    rot = int(os.environ.get("CONV_BENCH_ROTATE", "1"))
    xs = [x] + [torch.randn_like(x.float()).to(x.dtype) for _ in range(rot - 1)]
    outs = [out] + [torch.empty_like(out) for _ in range(rot - 1)]

    def launch(i):
        a.src[0].ptr, a.out = xs[i % rot].data_ptr(), outs[i % rot].data_ptr()
        A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv")
    for i in range(3):
        launch(i)
    torch.cuda.synchronize()
    L.pws_prof_enable(1)
    for i in range(12):
        launch(i)
    L.pws_prof_enable(0)
    r = A.prof_collect()
    ms = sorted(x_[4] for x_ in r)[len(r) // 2]
    print("%-11s n=%d %dx%d %d->%d %-5s exp=%s %-18s: %8.1f us  %6.1f TFLOP/s (algorithmic)" % (
        kname, n, h, w, cin, cout, "bf16" if bf16 else ("wino" if wino else "direct"), os.environ.get("PWS_EXPERIMENT", "0"), r[0][0], ms * 1e3,
        r[0][2] / (ms * 1e-3) / 1e12))
    secs = float(os.environ.get("CONV_BENCH_SECONDS", "0"))
    if secs > 0:
        # sustained: the same launch back to back for `secs` seconds, the engine clock and the socket power sampled beside it
        # (a 12-launch measurement is over before the power management has settled)
        import re
        import subprocess
        import threading
        import time
        stop, samples = threading.Event(), []

        def sampler():
            while not stop.is_set():
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                pw = re.search(r"Power \(W\): ([0-9.]+)", o)
                samples.append((int(c.group(1)) if c else -1, float(pw.group(1)) if pw else -1.0))
                time.sleep(0.3)
        th = threading.Thread(target=sampler)
        th.start()
        t_end, per = time.perf_counter() + secs, []
        i = 0
        while time.perf_counter() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                launch(i)
                i += 1
            e1.record()
            e1.synchronize()
            per.append(e0.elapsed_time(e1) / 300 * 1e3)
        stop.set()
        th.join()
        tail = samples[len(samples) // 2:]
        print("   sustained %.0f s: %.1f us per launch at the start, %.1f at the end (%.1f TFLOP/s); sclk %s MHz, power %s W (second half of the samples)" % (
            secs, per[0], per[-1], r[0][2] / (per[-1] * 1e-6) / 1e12, sorted(set(c for c, _ in tail)), sorted(set(round(q) for _, q in tail))[::max(1, len(tail) // 6)]))


def bench_first(n=8, h=256, w=256, cin=31, cout=64):
    """The generator's first layer on the NCHW window (conv_first_kernel; PWS_EXPERIMENT 25 = conv_mfma_kernel<k5s1, NCHW>)."""
    L, st = A.lib(), A.current_stream()
    L.pws_set_option(100, int(os.environ.get("PWS_EXPERIMENT", "0")))
    wt = torch.randn((cout, cin, 5, 5), device="cuda") / (cin * 25) ** 0.5
    wp = torch.empty(L.pws_packed_weight_floats(A.CONV_K5S1, cin, cout), device="cuda")
    A.check(L.pws_pack_conv_weight(A.ptr(wt), A.ptr(wp), A.CONV_K5S1, cin, cout, st), "pack")
    xs = [torch.randn((n, cin, h, w), device="cuda") for _ in range(3)]
    outs = [torch.empty((n, h, w, cout), device="cuda") for _ in range(3)]
    b = torch.randn(cout, device="cuda")
    a = A.PwsConvArgs()
    a.kind, a.n, a.h, a.w, a.nsrc, a.cout, a.act, a.src_nchw = A.CONV_K5S1, n, h, w, 1, cout, 1, 1
    a.src[0].channels, a.src[0].ld = cin, 0
    a.w_packed, a.bias, a.out_ld = wp.data_ptr(), b.data_ptr(), cout
    if L.pws_packed_wring_floats(A.CONV_K5S1, cin, cout):   # F(2x2,5x5) weights: wino5_first_kernel (PWS_EXPERIMENT 26 = conv_first_kernel)
        wr = torch.empty(L.pws_packed_wring_floats(A.CONV_K5S1, cin, cout), device="cuda")
        A.check(L.pws_pack_conv_weight_wring(A.ptr(wp), A.ptr(wr), A.CONV_K5S1, cin, cout, st), "pack_wring")
        a.w_wring = wr.data_ptr()

    def launch(i):
        a.src[0].ptr, a.out = xs[i % 3].data_ptr(), outs[i % 3].data_ptr()
        A.check(L.pws_conv2d_fwd(ctypes.byref(a), st), "conv")
    for i in range(3):
        launch(i)
    torch.cuda.synchronize()
    L.pws_prof_enable(1)
    for i in range(12):
        launch(i)
    L.pws_prof_enable(0)
    r = A.prof_collect()
    if not r:   # PWS_EXPERIMENT 1200: the kernel reports its own clock on stderr instead
        return
    ms = sorted(x_[4] for x_ in r)[len(r) // 2]
    print("first layer n=%d %dx%d %d->%d exp=%s %-30s: %8.1f us  %6.1f TFLOP/s (algorithmic)" % (
        n, h, w, cin, cout, os.environ.get("PWS_EXPERIMENT", "0"), r[0][0], ms * 1e3, r[0][2] / (ms * 1e-3) / 1e12))
    L.pws_set_option(100, 0)


if __name__ == "__main__":
    v = sys.argv[1:]
    if v and v[0] == "first":
        bench_first(*[int(t) for t in v[1:]])
        sys.exit(0)
    if len(v) > 7 and v[7] == "dgrad":
        bench_dgrad(v[0], int(v[1]), int(v[2]), int(v[3]), int(v[4]), int(v[5]))
        sys.exit(0)
    bench(v[0], int(v[1]), int(v[2]), int(v[3]), int(v[4]), int(v[5]), len(v) > 6 and v[6] == "wino", len(v) > 6 and v[6] == "bf16")

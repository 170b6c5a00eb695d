#!/usr/bin/env python3
"""Benchmark of the hot path: stabilized frames/s = frames for which ``field = netG(window, False)`` AND
``warped = grid_sample(frame, field)`` completed (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
N=1 runs in this process.  N>1: either already under torch.distributed.run (RANK / WORLD_SIZE in the environment, one rank per
GPU over RCCL), or -- when WORLD_SIZE is unset -- this process starts `python -m torch.distributed.run --nproc-per-node N` on
itself BEFORE anything touches the GPU, relays the ranks' output and exits with their code (it never measures one GPU and
calls it N).  WORLD_SIZE != --gpus is an error.

Workload at every N: BASELINE.json configs[1] per GPU -- batch=8 windows of 31x256x256 (fp32) through the HIP
generator + HIP grid_sample of 8 RGB 256x256 frames; inputs are synthetic and resident in HBM before the timed
region.  Frames are independent units, so ranks shard them with NO collective (weak scaling: 8 frames per GPU
per step); the only collectives are the timing barrier and the MAX over ranks of the elapsed time.

One JSON line on rank 0.  Besides the driver's contract it carries
  roofline      -- the dominant kernel (most GPU time in the timed region), from hipEvents recorded around every
                   launch of the K timed steps on the launch stream (pws_prof_* hooks of the C ABI);
  roofline_grid_sample -- same for grid_sample_fwd_kernel (HBM-bound, north_star's 40 % target), measured on a
                   larger batch so that the launch is not latency-bound;
  cpu_baseline  -- the hot path on this box's host cores (PyTorch-CPU restatement of the reference graph, i.e. the
                   ops the reference executes; oracle/torch_ref.py), bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak (no TF32 on gfx950)
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak
GFLOP_PER_FRAME_INFER = 94.48  # de-duplicated inference forward
GFLOP_PER_SAMPLE_TRAIN = 331.0  # training forward 112.50 + backward (no dgrad for `transfer`)
PEAK_BF16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak


def self_launch(a):
    """--gpus N > 1 without a launcher: start N ranks of this script under torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will: re-exec'ing or forking a process that initialised HIP takes the box down),
    pass their stdout / stderr through and return their exit code."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["PWS_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (configs[1]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the instrumented (hipEvent) repetition of the K steps")
    ap.add_argument("--no-graph", action="store_true", help="launch the forward eagerly instead of replaying a hipGraph")
    ap.add_argument("--serial", action="store_true", help="single queue for the whole run (clean per-kernel durations under rocprof)")
    ap.add_argument("--gs-batch", type=int, default=256, help="frames in the grid_sample roofline launch")
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32",
                    help="conv arithmetic of the TIMED region (default fp32 = configs[1], the headline); bf16 is for profiling the "
                         "bf16 path: metric/dtype fields say so and vs_baseline stays null")
    ap.add_argument("--no-extra", action="store_true", help="skip the bf16 inference / training-step legs (extra JSON fields)")
    ap.add_argument("--experiment", type=int, default=0, help="PWS_OPT_EXPERIMENT for the whole run (A/B switches of the library; 0 = product)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight in the inference legs: consecutive steps alternate over this many streams (one hipGraph + arena per stream, one "
                         "queue per forward), so that one batch's large layers fill the CUs the other batch's chain of short launches leaves idle; "
                         "1 = one step at a time on two queues (rounds 1-5; always with --serial)")
    ap.add_argument("--ddp-items", type=int, default=32, help="item pairs per GPU per step of the N>1 training leg (configs[3]: 32)")
    ap.add_argument("--stream-frames", type=int, default=128, help="720p frames per GPU of the N>1 streaming leg (configs[4])")
    ap.add_argument("--force-collectives", action="store_true",
                    help="--gpus 1 only: initialise a ONE-rank RCCL group and run the training_ddp leg with its collectives forced "
                         "(first contact with RCCL on a single GPU; rccl_ranks = 1 in the line)")
    return ap.parse_args()


# bench kernel name -> substrings of the rocprofv3 kernel names (tools/summarize_prof.py) that make it up
_PMC_NAMES = {
    "conv_mfma_kernel<convT4,16x16>": ["convT4,tile 1x16x16", "convT4,tile 1x8x16"],
    "conv_mfma_kernel<k3s1,16x16>": ["k3s1,tile 1x16x16", "k3s1,tile 1x8x16"],
    "wino_k3s1_kernel<F(2x2,3x3)>": ["wino_k3s1_kernel", "wino_kernel<0>"],
    "wino_ring_kernel<F(2x2,3x3)>": ["wino_ring_kernel<F(2x2,3x3)"],          # 32-wide and 16-wide map geometries
    "wino_ring_kernel<convT4,F(2x2,2x2)>": ["wino_ring_kernel<convT4"],        # 1 or 2 classes per unit, both geometries
    "conv_bf16_kernel": ["conv_bf16_kernel", "conv_bf16_k5_kernel"],
    "conv_mfma_kernel<k5s1,16x16>": ["k5s1,tile"],
    "conv_first_kernel": ["conv_first_kernel"],
    "wino5_first_kernel": ["wino5_first_kernel"],
    "grid_sample_fwd_kernel@roofline": ["grid_sample_fwd2_kernel<4, true"],   # the 256-frame launch (non-temporal variant: > 256 MB)
    "upsample_grid_sample_fwd_kernel": ["upsample_grid_sample_fwd_kernel"],
    "grid_sample_fwd_kernel": ["grid_sample_fwd2_kernel"],
}


# multiplies of the direct convolution per multiply the Winograd kernels execute (conv_wino.hip / conv_wring.hip)
WINOGRAD_REDUCTION = {"wino_k3s1_kernel<F(2x2,3x3)>": 2.25, "wino_ct4_kernel<F(3x3,2x2)>": 2.25, "wino_ring_kernel<F(2x2,3x3)>": 2.25,
                      "wino_ring_kernel<convT4,F(2x2,2x2)>": 16.0 / 9.0, "wino5_first_kernel": 100.0 / 36.0}


def pmc_traffic(kernel):
    """HBM bytes per launch from a committed PMC summary (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    tools/profile_round.sh): (2*FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE counts 128-B requests as 64 B on gfx950
    (MI355X_MICROARCH.md, HBM section).  Only a summary whose `_meta.source_hash` equals the hash of the kernel sources this
    library was built from counts (a stale summary would silently describe other kernels): else None."""
    import glob
    if kernel not in _PMC_NAMES:
        return None
    try:
        from pwstablenet_amd.build import source_hash
        want = source_hash()
    except Exception:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                pmc = json.load(f)
        except Exception:
            continue
        if pmc.get("_meta", {}).get("source_hash") != want:
            continue
        tot, n = 0.0, 0
        for k, v in pmc.items():
            if k != "_meta" and any(sub in k for sub in _PMC_NAMES[kernel]) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                d = v["FETCH_SIZE"]["dispatches"]
                tot += (2.0 * v["FETCH_SIZE"]["mean_per_dispatch"] + v["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0 * d
                n += d
        if n:
            return round(tot / n)
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()))))
        except Exception:
            pass
    return n


def cpu_provenance():
    """Why `cores` is what it is, so that the baseline compares across boxes: logical CPUs of the host, the affinity mask this process
    was given, the cgroup CPU quota (v2 cpu.max / v1 cfs quota) in CPUs, what limits, and the load average when the baseline started."""
    out = {"logical_cpus": os.cpu_count()}
    try:
        out["affinity_mask_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        out["affinity_mask_cpus"] = None
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(per), 2)
        out["cgroup"] = "v2 cpu.max = %s %s" % (q, per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            quota = None if q <= 0 else round(q / per, 2)
            out["cgroup"] = "v1 cfs_quota_us / cfs_period_us = %d / %d" % (q, per)
        except Exception:
            out["cgroup"] = "no cpu controller visible"
    out["cgroup_quota_cpus"] = quota
    aff = out["affinity_mask_cpus"] or out["logical_cpus"] or 1
    out["limited_by"] = ("cgroup quota" if quota is not None and quota < aff else
                         "affinity mask" if aff < (out["logical_cpus"] or aff) else "nothing (all logical CPUs)")
    try:
        out["loadavg_1m"] = round(os.getloadavg()[0], 2)
    except Exception:
        pass
    return out


def cpu_baseline(timed=None):
    """Hot path on the host cores: netG(window, False) + grid_sample, N=1, min over a bounded number of runs.
    timed = (x, frames, field, warped) as CPU tensors: the inputs of the TIMED region (batch 8) and what its launch path --
    hipGraph replay, two queues -- produced for them; the CPU path runs on the same inputs for the metric's error figures."""
    import torch
    from oracle import torch_ref
    from pwstablenet_amd import synth
    prov = cpu_provenance()
    cores = min(usable_cores(), 64)  # oneDNN convs of one 256x256 frame stop scaling long before 64 threads
    torch.set_num_threads(cores)
    params = [torch.from_numpy(v) for _, v in synth.make_weights("W1", seed=123, ngf=64)]
    x = torch.from_numpy(synth.noise_window(1, 31, 256, seed=123))
    fr = torch.from_numpy(synth.make_frames(1, 3, 256, 256, seed=321))
    torch_ref.stabilize_step(params, x, fr)  # warm-up
    best, runs, t_all = 1e9, 0, time.time()
    while runs < 10 and (time.time() - t_all) < 20.0:
        t0 = time.time()
        torch_ref.stabilize_step(params, x, fr)
        best = min(best, time.time() - t0)
        runs += 1
    res = {"value": round(1.0 / best, 3), "unit": "frames/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
           "provenance": dict(prov, torch_num_threads=torch.get_num_threads(), torch_num_interop_threads=torch.get_num_interop_threads()),
           "sample": "N=1 frame: PyTorch-CPU restatement of netG(x,False)+grid_sample (oracle/torch_ref.py), "
                     "min of %d runs after 1 warm-up, %d torch threads" % (runs, torch.get_num_threads()),
           # the reference's own Python cannot travel to this box: its numbers measured in the build container
           "reference_cpu_container": {"fps": 4.5, "fps_batch8": 4.3, "cores": 8,
                                       "what": "reference netG(x, False) imported unmodified, torch 2.10 CPU (oneDNN, AVX-512)",
                                       "source": "BASELINE.md section 2 / SURVEY.md section 6"}}
    # the split SURVEY 8(d) asks for, bounded (~8 s): generator only / grid_sample only at N=1, end to end at N=8
    def best_of(fn, max_runs, budget_s):
        fn()
        b, t_start = 1e9, time.time()
        for _ in range(max_runs):
            t0 = time.time()
            fn()
            b = min(b, time.time() - t0)
            if time.time() - t_start > budget_s:
                break
        return b
    with torch.no_grad():
        field1 = torch_ref.netg_forward(params, x, is_training=False)
        t_net = best_of(lambda: torch_ref.netg_forward(params, x, is_training=False), 5, 3.0)
        t_gs = best_of(lambda: torch.nn.functional.grid_sample(fr, field1, mode="bilinear", padding_mode="zeros", align_corners=False), 20, 1.0)
        x8 = torch.from_numpy(synth.noise_window(8, 31, 256, seed=123))
        fr8 = torch.from_numpy(synth.make_frames(8, 3, 256, 256, seed=321))
        t_n8 = best_of(lambda: torch_ref.stabilize_step(params, x8, fr8), 2, 4.0)
    res["detail"] = {"netg_only_fps_n1": round(1.0 / t_net, 3), "grid_sample_only_fps_n1": round(1.0 / t_gs, 1),
                     "end_to_end_fps_n8": round(8.0 / t_n8, 3)}
    if timed is not None:
        xt, frt, field, warped = timed
        with torch.no_grad():
            ref_field = torch_ref.netg_forward(params, xt, is_training=False)
            ref_warp = torch.nn.functional.grid_sample(frt, ref_field, mode="bilinear", padding_mode="zeros", align_corners=False)
        res["parity_vs_cpu_path"] = {"what": "the TIMED launch path (as config.launch says) on the timed region's own batch of %d "
                                             "windows / frames vs the CPU path on the same inputs" % xt.shape[0],
                                     "warp_field_max_abs_err": float((field - ref_field).abs().max()),
                                     "warped_frame_max_abs_err_on_pm1_scale": float((warped - ref_warp).abs().max() / 127.5),
                                     "bound": "1e-3 (north_star, fp32)"}
    return res


def sample_clock_and_power(run_once, seconds=1.5):
    """Engine clock (MHz) and socket power (W) from rocm-smi while ``run_once`` is repeated for ``seconds`` -- OUTSIDE any timed region.
    The bf16 matrix kernels pull the chip to its power cap (1.4 kW on MI355X): the dense peak that applies inside them is the spec peak
    times clock / 2400 (tools/clock_probe.sh).  Returns {} when rocm-smi is not there or says nothing."""
    import re
    import shutil
    import subprocess
    import threading
    if shutil.which("rocm-smi") is None:
        return {}
    stop, samples = threading.Event(), []

    def sampler():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\): ([0-9.]+)", o)
            if c and pw:
                samples.append((int(c.group(1)), float(pw.group(1))))
            time.sleep(0.15)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    import torch
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        run_once()
        torch.cuda.synchronize()
    stop.set()
    th.join(6)
    samples = samples[1:] if len(samples) > 2 else samples   # the first sample may predate the load
    if not samples:
        return {}
    clk = sorted(c for c, _ in samples)[len(samples) // 2]
    pw = sorted(p_ for _, p_ in samples)[len(samples) // 2]
    return {"sclk_mhz_median": clk, "socket_power_w_median": round(pw), "samples": len(samples)}


def configs2_step_leg(net, dev, n_items, math, reps, sync=None, seed=500, perceptual=None, per_rep=None, clock=None, warmup=1):
    """BASELINE configs[2] / [3] step, per GPU: n_items item pairs -> 2*n_items generator forwards (is_training), 6 fused
    warp+L1 launches, temporal / feature / smoothness / fp64 shape terms, backward through all of it, [gradient
    all-reduce,] fused Adam (reference main_new.py:84-216 without GAN and without the VGG term, which needs torchvision
    weights).  ``pwstablenet_amd.objective.train_step``; uint8 item tensors resident in HBM."""
    import torch
    from pwstablenet_amd import synth
    from pwstablenet_amd.objective import StabObjective, train_step
    from pwstablenet_amd.optim import Adam
    net.module.enable_graph(False)
    net.module.set_math(math)
    small = synth.make_train_batch(4, seed=seed)     # 4 distinct items tiled to n_items (host-side synthesis is slow)
    rep = (n_items + 3) // 4
    batch = [torch.from_numpy(t).repeat((rep,) + (1,) * (t.ndim - 1))[:n_items].to(dev) for t in small]
    # a leg without an exchange (no sync_gradients, no grad_sync on the generator) says so: its gradients stay this rank's own, whatever
    # process group the bench has initialised (train_step refuses to guess in a multi-rank group)
    exchanged = sync is not None or getattr(net.module, "grad_sync", None) is not None
    obj = StabObjective(batchSize=n_items, grad_average_world=None if exchanged else 1)
    opt = Adam(net.parameters(), lr=1e-6, betas=(0.5, 0.999))
    for _ in range(max(1, warmup)):   # warm-up (allocations, weight re-pack; at the power cap the first steps run at a higher clock)
        out = train_step(net, opt, batch, obj, sync_gradients=sync, perceptual=perceptual)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = train_step(net, opt, batch, obj, sync_gradients=sync, perceptual=perceptual)
        if per_rep is not None:   # diagnostics (tools/configs2_step.py --per-rep): a host sync per step, host and device time apart
            th = time.perf_counter()
            torch.cuda.synchronize()
            per_rep.append((th, time.perf_counter()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    loss = float(out.loss_g.detach())
    assert loss == loss and abs(loss) != float("inf"), loss
    if clock is not None:   # after the timed steps: what clock / power the chip holds under this step
        try:
            clock.update(sample_clock_and_power(lambda: train_step(net, opt, batch, obj, sync_gradients=sync, perceptual=perceptual)))
        except Exception as e:  # never at the cost of the leg
            clock["error"] = str(e)[:120]
    del opt, batch
    net.zero_grad(set_to_none=True)
    return dt, loss


def bf16_legs(net, x, frames, out_fp32, a, PF, A, synth):
    """Extra fields (not `value`): the same inference step with bf16 conv math and bf16 activation storage (PWS_MATH_BF16 +
    PWS_STORE_BF16: bf16 matrix cores, fp32 accumulation; weights, biases, fields fp32), its error against the fp32 step, and one configs[2]-shaped training step
    (forward is_training + 3 grid_sample + L1 + backward + fused Adam) per math mode at this batch size."""
    import torch
    from pwstablenet_amd.optim import Adam
    B = x.shape[0]
    res = {}
    with torch.no_grad():
        net.module.enable_graph(False)
        f32 = net(x, False).clone()
        net.module.set_math("bf16")
        f16 = net(x, False).clone()
        net.module.enable_graph(not a.no_graph)

        def step():
            return PF.grid_sample(frames, net(x, False))
        # one step at a time on two queues (rounds 1-5), then D batches in flight as the headline leg (see there)
        D = 1 if a.serial else max(1, a.in_flight)
        for _ in range(max(a.warmup, 2)):
            o = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            o = step()
        torch.cuda.synchronize()
        dt = dt_one = time.perf_counter() - t0
        if D > 1:
            lanes = [torch.cuda.Stream(x.device) for _ in range(D)]
            net.module.two_queues = False
            net.module.enable_graph(not a.no_graph, per_stream=True)
            try:
                def step_on(i):
                    with torch.cuda.stream(lanes[i % D]):
                        return step()
                for i in range(max(a.warmup, 2 * D)):
                    o = step_on(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(a.steps):
                    o = step_on(i)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                net.module.two_queues = None
                net.module.enable_graph(not a.no_graph)
                torch.cuda.synchronize()
    res["inference"] = {"value": round(B * a.steps / dt, 2), "unit": "frames/s", "n_gpus": 1, "dtype": "bf16 operands + activation storage, f32 accumulate",
                        "ms_per_step": round(1e3 * dt / a.steps, 4), "batches_in_flight": D,
                        "value_one_in_flight": round(B * a.steps / dt_one, 2),
                        "field_max_abs_err_vs_fp32": float((f16 - f32).abs().max()),
                        "warped_max_abs_err_vs_fp32_over_255": float((o - out_fp32).abs().max() / 255.0),
                        "netg_tflops": round(B * a.steps / dt * GFLOP_PER_FRAME_INFER / 1e3, 1),
                        "netg_frac_bf16_peak": round(B * a.steps / dt * GFLOP_PER_FRAME_INFER / 1e3 / PEAK_BF16_TFLOPS, 4)}
    net.module.enable_graph(False)
    tg = torch.roll(frames, shifts=(2, -3), dims=(2, 3))
    train = {}
    for math in ("fp32", "bf16"):
        net.module.set_math(math)
        opt = Adam(net.parameters(), lr=1e-6, betas=(0.5, 0.999))

        def tstep():
            opt.zero_grad()
            grids, _ = net(x)
            loss = sum(torch.nn.functional.l1_loss(PF.grid_sample(frames, g) / 127.5 - 1, tg / 127.5 - 1) for g in grids)
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            tstep()
        torch.cuda.synchronize()
        reps = max(3, a.steps // 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = tstep()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(loss)
        train[math] = {"samples_per_s": round(B * reps / dt, 1), "ms_per_step": round(1e3 * dt / reps, 2),
                       "tflops": round(B * reps / dt * GFLOP_PER_SAMPLE_TRAIN / 1e3, 1)}
        del opt
    # configs[2] itself: batch=32 ITEM PAIRS per step = 64 generator forwards + the objective, bf16 math
    try:
        clk = {}
        dt, loss = configs2_step_leg(net, x.device, 32, "bf16", 3, clock=clk)
        train["configs2_bf16_batch32"] = {
            "workload": "configs[2]: 32 item pairs per step = 64 netG forwards (is_training) + fused warp/L1 x6, temporal, feature, "
                        "smoothness, fp64 shape terms + backward + fused Adam (train() of main_new.py:84-216, no GAN, no VGG term)",
            "items_per_s": round(32 / dt, 1), "forwards_per_s": round(64 / dt, 1), "ms_per_step": round(1e3 * dt, 2),
            "tflops": round(64 / dt * GFLOP_PER_SAMPLE_TRAIN / 1e3, 1),
            "frac_bf16_peak": round(64 / dt * GFLOP_PER_SAMPLE_TRAIN / 1e3 / PEAK_BF16_TFLOPS, 4), "loss_g": round(loss, 4)}
        if clk.get("sclk_mhz_median"):
            # the step runs at the socket's power cap: the peak the matrix cores can reach at the clock the chip holds (spec peak x clock / 2400 MHz)
            c2 = train["configs2_bf16_batch32"]
            c2["under_this_step"] = clk
            c2["frac_bf16_peak_at_held_clock"] = round(c2["frac_bf16_peak"] * 2400.0 / clk["sclk_mhz_median"], 4)
    except Exception as e:  # an extra leg must never cost the headline line
        train["configs2_bf16_batch32"] = {"error": str(e)[:200]}
    # the same step with the VGG-16 perceptual term of train() (main_new.py:191-192) on top: 4 x 64 VGG forwards (3 stages of
    # warped frames + the stable frames) and the data-gradient backward of the 3 x 64, bf16 math, random VGG weights
    try:
        from pwstablenet_amd.perceptual import GeneratorLoss, VGG16Features, perceptual_term
        crit = GeneratorLoss(VGG16Features("bf16").init_random(0)).to(x.device)
        dt, loss = configs2_step_leg(net, x.device, 32, "bf16", 2, perceptual=perceptual_term(crit))
        vgg_gf = 64 * (4 * 40.1 + 3 * 40.1)   # fwd of 4 x 64 images + dgrad of 3 x 64 (conv1_1's input gradient included)
        train["configs2_bf16_batch32_with_vgg"] = {
            "workload": "configs2_bf16_batch32 + VGG-16 features[:31] perceptual MSE per stage and branch (frozen random weights)",
            "items_per_s": round(32 / dt, 1), "ms_per_step": round(1e3 * dt, 2),
            "tflops": round((64 * GFLOP_PER_SAMPLE_TRAIN + vgg_gf) / dt / 1e3, 1)}
        del crit
    except Exception as e:
        train["configs2_bf16_batch32_with_vgg"] = {"error": str(e)[:200]}
    net.module.set_math("fp32")
    net.zero_grad(set_to_none=True)
    res["training_step"] = {"workload": "batch=%d: netG(x) is_training + 3 grid_sample + L1 + backward + fused Adam, one netG "
                                        "forward per sample (configs[2] runs two per item); fp32 master weights" % B,
                            "flops_per_sample_gf": GFLOP_PER_SAMPLE_TRAIN, **train}
    return res


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line: native libraries write to file descriptor 1 as well (RCCL prints its version banner there when
    # the first communicator is made), so descriptor 1 is pointed at stderr for the rest of the process and the line goes out
    # through a private duplicate of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-GPU number as a %d-GPU one" % (a.gpus, world, world, a.gpus),
              file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist
    ctl_device = None  # device of the control-plane tensors (timing barrier / MAX-reduce); the data path has no collective
    control_plane = "none (single process)"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # test hooks (one-GPU boxes): PWS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, PWS_BENCH_BACKEND=gloo routes the
        # collectives through the host -- exercises the multi-rank control flow where RCCL cannot run
        if os.environ.get("PWS_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        launch_check = os.environ.get("PWS_BENCH_LAUNCH_CHECK") == "1"   # CPU test of the launcher + control plane: no GPU work
        if not launch_check:
            torch.cuda.set_device(local_rank)
        if os.environ.get("PWS_BENCH_BACKEND", "nccl") != "nccl":
            # explicit test hook only; a FAILED RCCL init is fatal (a gloo run must never pass for an RCCL one)
            dist.init_process_group("gloo")
            ctl_device = torch.device("cpu")
            control_plane = "gloo (PWS_BENCH_BACKEND test hook: RCCL not exercised)"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
            ctl_device = torch.device("cuda", local_rank)
            control_plane = "rccl"
        if launch_check:
            t = torch.tensor([1.0 + rank], dtype=torch.float64, device=ctl_device)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if rank == 0:
                print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_arg": a.gpus, "max_over_ranks": float(t.item()),
                                  "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                                  "control_plane": control_plane,
                                  "self_launched": os.environ.get("PWS_BENCH_SELF_LAUNCHED") == "1"}), file=json_out, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            return
    else:
        torch.cuda.set_device(0)
        if a.force_collectives:
            # a group of ONE rank over RCCL, made exactly as the N > 1 branch makes its group
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29547")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
            ctl_device = torch.device("cuda", 0)
            control_plane = "rccl (one-rank group: --force-collectives)"
    force = bool(a.force_collectives) and world == 1
    dev = torch.device("cuda", torch.cuda.current_device())

    from pwstablenet_amd import functional as PF
    from pwstablenet_amd import hipabi as A
    from pwstablenet_amd import synth
    from pwstablenet_amd.lib.networks_cascading import define_G

    B = a.batch
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # define_G prints like the reference's init_weights; stdout carries ONE JSON line
        net = define_G(31, 2, 64, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W1", seed=123, ngf=64)})
    net = net.to(dev)
    net.module.set_math(a.math)
    x = torch.from_numpy(synth.noise_window(B, 31, 256, seed=123 + rank)).to(dev)
    frames = torch.from_numpy(synth.make_frames(B, 3, 256, 256, seed=321 + rank)).to(dev)

    def step():
        with torch.no_grad():
            grid = net(x, False)
            return PF.grid_sample(frames, grid)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.serial:
        A.lib().pws_set_option(A.OPT_TWO_QUEUES, 0)
    if a.experiment:
        A.lib().pws_set_option(A.OPT_EXPERIMENT, a.experiment)   # (captured graphs keep the kernel selection of their capture)
    net.module.enable_graph(not a.no_graph)  # the ~75 launches of a forward replay as one hipGraph launch
    # Batches in flight (round 6): the K steps are independent batches, so step i is issued on stream i % D without waiting for step i - 1
    # (one graph + arena per stream: UnetGenerator keeps them per stream; one queue per forward -- two forwards that both fork into the
    # device's side queue do not overlap).  While one batch walks the chain of its <= 8x8 levels (one 5-25 us launch at a time, most CUs
    # idle), the other batch's large layers run.  D = 1: one step at a time on two queues, as rounds 1-5 timed it (value_one_in_flight).
    D = 1 if a.serial else max(1, a.in_flight)
    one_fps = None
    lanes = None
    if D > 1:
        if rank == 0 and world == 1 and not a.no_extra:
            for _ in range(a.warmup):
                step()
            torch.cuda.synchronize()
            t1f = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            one_fps = B * a.steps / (time.perf_counter() - t1f)
        lanes = [torch.cuda.Stream(dev) for _ in range(D)]

    def set_in_flight(on):
        """D > 1: one queue per forward and one graph + arena per stream while steps alternate over the lanes; off: the process defaults again."""
        if lanes is None:
            return
        torch.cuda.synchronize()
        net.module.two_queues = False if on else None
        net.module.enable_graph(not a.no_graph, per_stream=bool(on))
        if not on:
            step()   # (the default graph is captured here, if it was not yet, and not inside a later instrumented loop)
            torch.cuda.synchronize()

    def on_lane(i, fn):
        if lanes is None:
            return fn()
        with torch.cuda.stream(lanes[i % D]):
            return fn()

    def step_on(i):
        return on_lane(i, step)
    set_in_flight(True)
    for i in range(max(a.warmup, 2 * D)):
        step_on(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step_on(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    # NOT the headline: the same step with the one dead layer of the inference forward pruned (UnetGenerator.prune_dead /
    # PWS_NETG_PRUNE_DEAD: stage 1's up2, whose output the reference reads only under `if is_training`, lib/networks_cascading.py:171,173,196).
    # Same field bit for bit (tests/test_hip_netg.py); the headline keeps the layer because the reference executes it.
    pruned_fps = None
    if rank == 0 and world == 1 and not a.no_extra:
        try:
            net.module.prune_dead = True
            for i in range(2 * D):
                step_on(i)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(a.steps):
                step_on(i)
            torch.cuda.synchronize()
            pruned_fps = B * a.steps / (time.perf_counter() - tp)
        except Exception:
            pruned_fps = None
        finally:
            net.module.prune_dead = False
            for i in range(D):
                step_on(i)   # (the default path's graphs again, for everything below)
            torch.cuda.synchronize()
    # what the timed launch path produced for its inputs (one more replay of the same path), for parity_vs_cpu_path
    timed_pair = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        with torch.no_grad(), torch.cuda.stream(lanes[0] if lanes else torch.cuda.current_stream()):
            tp_field, tp_warp = net(x, False), step()
        torch.cuda.synchronize()
        timed_pair = (tp_field.cpu(), tp_warp.cpu())
    # 720p leg of the metric on EVERY rank (configs[4] shape, frame-sharded, no collective): netG on 256x256 windows + fused
    # field-resize+warp of 1280x720 RGB frames (reference main_new.py:697-716), frames resident in HBM
    f720 = torch.rand((B, 3, 720, 1280), device=dev) * 255

    def step720():
        with torch.no_grad():
            return PF.upsample_grid_sample(f720, net(x, False))
    for i in range(2 * D):
        on_lane(i, step720)
    barrier()
    t1 = time.perf_counter()
    for i in range(a.steps):
        o720 = on_lane(i, step720)
    torch.cuda.synchronize()
    dt720 = time.perf_counter() - t1
    barrier()
    assert torch.isfinite(o720).all()
    del o720
    set_in_flight(False)   # the legs below run one step at a time: the process default (two queues, one graph) again
    if world > 1:
        t = torch.tensor([dt720], device=ctl_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt720 = float(t.item())
    # Per-kernel events cannot be recorded inside a graph replay, and in the timed region two queues overlap kernels
    # (stage k+1's encoder beside stage k's decoder), so the per-kernel roofline comes from an instrumented repetition of
    # the same K steps right after the timed region: eager launches on ONE queue, each bracketed by hipEvents.
    recs = []
    if not a.no_prof and rank == 0:
        net.module.enable_graph(False)
        A.lib().pws_set_option(A.OPT_TWO_QUEUES, 0)
        step()
        torch.cuda.synchronize()
        A.lib().pws_prof_enable(1)
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        A.lib().pws_prof_enable(0)
        recs = A.prof_collect(1 << 16)
        A.lib().pws_set_option(A.OPT_TWO_QUEUES, 0 if a.serial else 1)
    if world > 1:
        t = torch.tensor([elapsed], device=ctl_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        fps = world * B * a.steps / elapsed
        line = {
            "metric": "stabilized frames/sec at 256x256 (netG %s + grid_sample), whole job" % a.math,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0,
            "control_plane": control_plane, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.math == "fp32" else "bf16 operands, f32 accumulate", "data": "synthetic",
            "config": {"workload": "configs[1]: batch=8 256x256 inference per GPU, fp32 HIP conv + grid_sample; "
                                   "frame-sharded, no collective", "frames_per_gpu_per_step": B,
                       "launch": ("eager" if a.no_graph else "hipGraph replay of the forward + 1 grid_sample launch") +
                                 (", single queue" if a.serial else (", two queues" if D == 1 else ", %d batches in flight (step i on stream i %% %d: one graph + arena per stream, one queue per forward)" % (D, D)) +
                                  "; stages 2 and 3 in lockstep (their shared layers as one launch of batch 2n: 45 launches per forward)"),
                       "batches_in_flight": D,
                       "window": "31x256x256", "frame": "3x256x256", "weights": "synthetic W1 (pwstablenet_amd.synth)"},
            "netg_tflops_per_gpu": round(fps / world * GFLOP_PER_FRAME_INFER / 1e3, 2),
            **({"value_one_in_flight": {"value": round(one_fps, 2), "unit": "frames/s",
                                        "what": "the same K steps one at a time (each step waits for the one before), two queues: how rounds 1-5 timed `value`"}} if one_fps else {}),
            **({"value_dead_layer_pruned": {
                "value": round(pruned_fps, 2), "unit": "frames/s", "what": "NOT the headline: the same step with stage 1's up2 pruned from the inference forward "
                "(pws_netg_opts.flags PWS_NETG_PRUNE_DEAD): its output x122 is read only under `if is_training` in the reference (lib/networks_cascading.py:171,173,196), "
                "so the returned field is bit-identical; 92.33 instead of 94.48 GFLOP per frame"}} if pruned_fps else {}),
            "netg_frac_fp32_peak": round(fps / world * GFLOP_PER_FRAME_INFER / 1e3 / PEAK_FP32_TFLOPS, 4),
        }
        if recs:
            agg = {}
            for name, tag, fl, by, ms in recs:
                e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                e[0] += 1; e[1] += fl; e[2] += by; e[3] += ms  # noqa: E702
            dom = max(agg.items(), key=lambda kv: kv[1][3])
            name, (cnt, fl, by, ms) = dom
            # Winograd kernels: `fl` counts the DIRECT convolution's flops (the algorithmic work of the layer); the matrix
            # cores execute 1 / reduction of them, and the roofline fraction is taken on the EXECUTED flops (utilisation), with the
            # direct-equivalent rate quoted beside it
            red = WINOGRAD_REDUCTION.get(name, 1.0)
            ach = fl / red / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_TFLOPS if "bf16" in name else PEAK_FP32_TFLOPS
            line["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": pmc_traffic(name),
                                "algorithmic_bytes_per_launch": round(by / cnt),
                                "launches": cnt, "avg_launch_us": round(1e3 * ms / cnt, 2),
                                "flops_per_launch": fl / red / cnt}
            if red != 1.0:
                line["roofline"]["winograd_reduction"] = round(red, 4)
                line["roofline"]["direct_equivalent_tflops"] = round(fl / (ms * 1e-3) / 1e12, 2)
                line["roofline"]["what"] = ("achieved / frac = matrix-core flops actually executed (direct flops / reduction) over the "
                                            "launch time: utilisation of the fp32 MFMA peak; direct_equivalent_tflops = direct flops / time")
            tot_ms = sum(v[3] for v in agg.values())
            line["kernel_time_share"] = {k: round(v[3] / tot_ms, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][3])}
            line["kernel_tflops"] = {k: round(v[1] / (v[3] * 1e-3) / 1e12, 2) for k, v in agg.items()
                                     if v[3] > 0 and ("conv" in k or "wino" in k)}
            line["kernel_time_over_step_time"] = round(tot_ms * 1e-3 / elapsed, 4)
        # grid_sample roofline on a batch large enough not to be launch-latency-bound (537 MB per launch: beyond the 256 MB
        # Infinity Cache).  Field = what a stabiliser emits: a random ~5 % affine map + a smooth +-2 px residual; the same
        # launch on a field with 1 px of WHITE NOISE per pixel (no trained generator emits that) is reported beside it.
        GB = a.gs_batch
        big = torch.rand((GB, 3, 256, 256), device=dev) * 255
        theta = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(GB, 1)
        theta = theta + 0.05 * torch.randn_like(theta)
        base = PF.affine_grid(theta, (GB, 3, 256, 256))
        ramp = torch.linspace(0, 6.28, 256, device=dev)
        smooth = base + (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1))
        noisy = base + (2.0 / 256) * torch.randn((GB, 256, 256, 2), device=dev)  # ~1 px jitter
        res_gs = {}
        for tag, grid in (("smooth", smooth), ("noisy", noisy)):
            with torch.no_grad():
                for _ in range(3):
                    PF.grid_sample(big, grid)
                torch.cuda.synchronize()
                A.lib().pws_prof_enable(1)
                for _ in range(20):
                    PF.grid_sample(big, grid)
                A.lib().pws_prof_enable(0)
            r = [x_ for x_ in A.prof_collect() if x_[0] == "grid_sample_fwd_kernel"]
            ms = sorted(x_[4] for x_ in r)[len(r) // 2]  # median launch
            res_gs[tag] = (r[0][3] / (ms * 1e-3) / 1e9, ms, r[0][3])
        gbs, ms, by = res_gs["smooth"]
        line["roofline_grid_sample"] = {"kernel": "grid_sample_fwd_kernel", "bound": "hbm", "achieved": round(gbs, 1),
                                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                        "traffic": pmc_traffic("grid_sample_fwd_kernel@roofline"), "avg_launch_us": round(1e3 * ms, 2),
                                        "bytes_per_launch": by, "frames_per_launch": GB,
                                        "field": "random 5% affine + smooth +-2 px residual",
                                        "achieved_white_noise_field": round(res_gs["noisy"][0], 1),
                                        "frac_white_noise_field": round(res_gs["noisy"][0] / PEAK_HBM_GBS, 4),
                                        "frac_of_measured_copy_rate": None}
        # the practical ceiling: a plain device copy of the same number of bytes (read half, write half)
        cp_src = torch.empty(int(by // 8), device=dev)
        cp_dst = torch.empty_like(cp_src)
        for _ in range(3):
            cp_dst.copy_(cp_src)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in evs:
            e0.record(); cp_dst.copy_(cp_src); e1.record()  # noqa: E702
        torch.cuda.synchronize()
        cp_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[5]
        line["roofline_grid_sample"]["device_copy_same_bytes_gb_per_s"] = round(by / (cp_ms * 1e-3) / 1e9, 1)
        line["roofline_grid_sample"]["frac_of_measured_copy_rate"] = round(gbs / (by / (cp_ms * 1e-3) / 1e9), 4)
        # grid_sample backward, the field gradient of a warp (frame = data, main_new.py:106-118): 40 B/pixel, same batch / field
        try:
            gsm = smooth.clone().requires_grad_(True)
            gup = torch.randn((GB, 3, 256, 256), device=dev)
            for _ in range(2):
                PF.grid_sample(big, gsm).backward(gup)
            gsm.grad = None
            torch.cuda.synchronize()
            A.lib().pws_prof_enable(1)
            for _ in range(10):
                PF.grid_sample(big, gsm).backward(gup)
                gsm.grad = None
            A.lib().pws_prof_enable(0)
            r = [x_ for x_ in A.prof_collect() if x_[0] == "grid_sample_bwd_kernel"]
            ms = sorted(x_[4] for x_ in r)[len(r) // 2]
            byb = 40.0 * GB * 256 * 256
            line["roofline_grid_sample_bwd"] = {"kernel": "grid_sample_bwd_field_kernel", "bound": "hbm",
                                                "achieved": round(byb / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                "frac": round(byb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                                "avg_launch_us": round(1e3 * ms, 2), "bytes_per_launch": byb, "frames_per_launch": GB,
                                                "what": "gradient wrt the field only (8 field + 12 upstream + 12 frame + 8 result B/px)"}
            del gsm, gup
        except Exception as e:
            line["roofline_grid_sample_bwd"] = {"error": str(e)[:200]}
        del big, base, smooth, noisy, cp_src, cp_dst
        # the objective kernels round the path (csrc/objective.hip: warp + L1, temporal, feature, smoothness, shape terms, forward and
        # backward) on 2 x 128 samples: algorithmic bytes of every launch over the sum of their durations
        try:
            from pwstablenet_amd.objective import StabObjective
            no, mo = 128, 256
            th = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(mo, 1)
            ramp = torch.linspace(0, 6.28, 256, device=dev)
            res_o = (2.0 / 256) * torch.stack([torch.sin(3 * ramp).view(1, 256, 1) * torch.cos(2 * ramp).view(1, 1, 256),
                                               torch.cos(2 * ramp).view(1, 256, 1) * torch.sin(ramp).view(1, 1, 256)], -1).repeat(mo, 1, 1, 1)
            grids_o = [(PF.affine_grid(th + 0.02 * torch.randn_like(th), (mo, 3, 256, 256)) + res_o).requires_grad_(True) for _ in range(3)]
            resid_o = [res_o.clone().requires_grad_(True) for _ in range(3)]
            rgb_o = torch.rand((mo, 3, 256, 256), device=dev) * 2 - 1
            stab_o = torch.rand((mo, 3, 256, 256), device=dev) * 2 - 1
            st_ = torch.rand((mo, 400, 2), device=dev) * 1.9 - 0.95
            one_ = torch.ones((mo, 400, 1), device=dev)
            feats_o = torch.cat([st_, one_, st_ + 0.02 * torch.randn_like(st_), one_], 2)
            adj_o = (torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(no, 1) + 0.01 * torch.randn((no, 6), device=dev))
            objective_o = StabObjective(batchSize=no)

            def obj_step():
                o_ = objective_o(grids_o, resid_o, rgb_o, stab_o, feats_o, adj_o)
                o_.loss_g.backward()
                for t_ in grids_o + resid_o:
                    t_.grad = None
            obj_step()
            torch.cuda.synchronize()
            A.lib().pws_prof_enable(1)
            for _ in range(3):
                obj_step()
            A.lib().pws_prof_enable(0)
            r = [x_ for x_ in A.prof_collect() if x_[0] == "objective_kernels"]
            tot_b, tot_ms = sum(x_[3] for x_ in r), sum(x_[4] for x_ in r)
            slow = sorted(r, key=lambda x_: -x_[4])[0]
            line["roofline_objective"] = {"kernel": "objective kernels (csrc/objective.hip), forward + backward, 256 samples", "bound": "hbm",
                                          "achieved": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": round(tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                          "launches_per_step": len(r) // 3, "ms_per_step": round(tot_ms / 3, 3),
                                          "bytes_per_step": tot_b / 3,
                                          "slowest_launch": {"us": round(1e3 * slow[4], 1), "gb_per_s": round(slow[3] / (slow[4] * 1e-3) / 1e9, 1)}}
            del grids_o, resid_o, rgb_o, stab_o, feats_o, res_o
        except Exception as e:
            line["roofline_objective"] = {"error": str(e)[:200]}
        # roofline of the fused 720p warp (the timing of the leg itself ran on every rank above)
        with torch.no_grad():
            A.lib().pws_prof_enable(1)
            for _ in range(5):
                step720()
            A.lib().pws_prof_enable(0)
        r = [x_ for x_ in A.prof_collect() if x_[0] == "upsample_grid_sample_fwd_kernel"]
        ms = sorted(x_[4] for x_ in r)[len(r) // 2]
        gbs = r[0][3] / (ms * 1e-3) / 1e9
        line["value_720p"] = {"value": round(world * B * a.steps / dt720, 2), "unit": "frames/s", "n_gpus": world,
                              "workload": "batch=%d per GPU: netG(31x256x256 window, fp32) + fused upsample(256^2 field)+grid_sample "
                                          "of 3x720x1280 fp32 frames; frame-sharded, no collective; %d batches in flight" % (B, D),
                              "roofline_warp": {"kernel": "upsample_grid_sample_fwd_kernel", "bound": "hbm",
                                                "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_us": round(1e3 * ms, 2),
                                                "bytes_per_launch": r[0][3], "traffic": pmc_traffic("upsample_grid_sample_fwd_kernel")}}
        # the same kernel on the field a stabiliser emits (2 % affine + smooth residual) instead of the random-weight
        # generator's (whose residual jumps by up to ~100 px between neighbouring 256x256 cells), inputs rotated so that
        # they come from HBM, not from the Infinity Cache
        try:
            rot = [f720] + [torch.rand((B, 3, 720, 1280), device=dev) * 255 for _ in range(3)]
            th = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(B, 1)
            ramp = torch.linspace(0, 6.28, 256, device=dev)
            fsm = PF.affine_grid(th + 0.02 * torch.randn_like(th), (B, 3, 256, 256)) + \
                (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1))
            with torch.no_grad():
                for b_ in rot:
                    PF.upsample_grid_sample(b_, fsm)
                torch.cuda.synchronize()
                A.lib().pws_prof_enable(1)
                for i_ in range(12):
                    PF.upsample_grid_sample(rot[i_ % 4], fsm)
                A.lib().pws_prof_enable(0)
            r = [x_ for x_ in A.prof_collect() if x_[0] == "upsample_grid_sample_fwd_kernel"]
            ms = sorted(x_[4] for x_ in r)[len(r) // 2]
            rw = line["value_720p"]["roofline_warp"]
            # headline figures = the stabiliser's field; the random-weight generator's field is kept beside them
            rw["achieved_random_weight_field"], rw["frac_random_weight_field"] = rw["achieved"], rw["frac"]
            rw["avg_launch_us_random_weight_field"] = rw["avg_launch_us"]
            rw["field"] = "2 % affine + smooth +-2 px residual, inputs rotated through 4 buffers (HBM, not Infinity Cache)"
            rw["achieved"] = round(r[0][3] / (ms * 1e-3) / 1e9, 1)
            rw["frac"] = round(r[0][3] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            rw["avg_launch_us"] = round(1e3 * ms, 2)
            del rot, fsm
        except Exception as e:
            line["value_720p"]["roofline_warp"]["smooth_field_error"] = str(e)[:200]
        del f720
        if not a.no_extra and world == 1:
            # the same leg on uint8 HWC frames (what cv2 hands over / the writer takes, main_new.py:679-721): 6 B/px of frame traffic
            u720 = torch.randint(0, 256, (B, 720, 1280, 3), device=dev, dtype=torch.uint8)
            with torch.no_grad():
                def step720u8():
                    return PF.upsample_grid_sample_u8(u720, net(x, False), swap_rb=True)
                set_in_flight(True)
                try:
                    for i in range(2 * D):
                        on_lane(i, step720u8)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(a.steps):
                        on_lane(i, step720u8)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t1
                finally:
                    set_in_flight(False)
                A.lib().pws_prof_enable(1)
                for _ in range(5):
                    step720u8()
                A.lib().pws_prof_enable(0)
            r = [x_ for x_ in A.prof_collect() if x_[0] == "upsample_grid_sample_u8_kernel"]
            ms = sorted(x_[4] for x_ in r)[len(r) // 2]
            gbs = r[0][3] / (ms * 1e-3) / 1e9
            line["value_720p_u8"] = {"value": round(B * a.steps / dt, 2), "unit": "frames/s", "n_gpus": 1,
                                     "workload": "as value_720p with uint8 HWC BGR frames in, uint8 HWC RGB out",
                                     "roofline_warp": {"kernel": "upsample_grid_sample_u8_kernel", "bound": "hbm",
                                                       "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                       "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_us": round(1e3 * ms, 2),
                                                       "bytes_per_launch": r[0][3]}}
            # as for the fp32 warp above: the same kernel on the field a stabiliser emits, inputs rotated through 4 buffers (HBM)
            try:
                rot8 = [u720] + [torch.randint(0, 256, (B, 720, 1280, 3), device=dev, dtype=torch.uint8) for _ in range(3)]
                th = torch.tensor([1, 0, 0, 0, 1, 0], device=dev, dtype=torch.float32).repeat(B, 1)
                ramp = torch.linspace(0, 6.28, 256, device=dev)
                fsm = PF.affine_grid(th + 0.02 * torch.randn_like(th), (B, 3, 256, 256)) + \
                    (4.0 / 256) * (torch.sin(3 * ramp).view(1, 256, 1, 1) * torch.cos(2 * ramp).view(1, 1, 256, 1))
                with torch.no_grad():
                    for b_ in rot8:
                        PF.upsample_grid_sample_u8(b_, fsm, swap_rb=True)
                    torch.cuda.synchronize()
                    A.lib().pws_prof_enable(1)
                    for i_ in range(12):
                        PF.upsample_grid_sample_u8(rot8[i_ % 4], fsm, swap_rb=True)
                    A.lib().pws_prof_enable(0)
                r = [x_ for x_ in A.prof_collect() if x_[0] == "upsample_grid_sample_u8_kernel"]
                ms = sorted(x_[4] for x_ in r)[len(r) // 2]
                rw = line["value_720p_u8"]["roofline_warp"]
                rw["achieved_random_weight_field"], rw["frac_random_weight_field"] = rw["achieved"], rw["frac"]
                rw["avg_launch_us_random_weight_field"] = rw["avg_launch_us"]
                rw["field"] = "2 % affine + smooth +-2 px residual, inputs rotated through 4 buffers (HBM, not Infinity Cache)"
                rw["achieved"] = round(r[0][3] / (ms * 1e-3) / 1e9, 1)
                rw["frac"] = round(r[0][3] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                rw["avg_launch_us"] = round(1e3 * ms, 2)
                rw["bound_note"] = ("not HBM-bound at 6 B/pixel: bound by the NUMBER and alignment of its vector memory instructions (23 -> 11 per lane "
                                    "of 4 pixels in round 6: 35 -> 22 us), then by the vector ALU (conversions and VOP3 forms issue in ~4.5 cycles); "
                                    "profiles/r06_warp_u8_rewrite.txt, DESIGN.md section 8")
                del rot8, fsm
            except Exception as e:
                line["value_720p_u8"]["roofline_warp"]["smooth_field_error"] = str(e)[:200]
            del u720
            # configs[4] shape on one GPU, PCIe inclusive, the whole device side of process(): 192 decoded uint8 720p frames in
            # PINNED HOST memory -> VideoStabilizer.run_video (H2D per 64-frame chunk on a side stream, gray + INTER_AREA window
            # planes computed on the device from the uploaded frames, batched windows, netG fp32, fused u8 warp, the 2x
            # INTER_AREA down-scale of the output as main_new.py:723, D2H on a third stream) -> 640x360 frames in host memory
            try:
                from pwstablenet_amd.stream import VideoStabilizer
                T = 192
                SB = 32   # windows per generator call = frames per uploaded chunk: tools/stream_sweep.py, round 4 -- fp32 batch 8 / 16 / 32: 1611 / 1751 / 1803 f/s, bf16 4501 / 5818 / 6595 (chunk 64); chunk 16 / 32 / 64 at batch 32: fp32 1834 / 1848 / 1795, bf16 6455 / 7160 / 6365
                u8_h = torch.randint(0, 256, (T, 720, 1280, 3), dtype=torch.uint8).pin_memory()
                vs = VideoStabilizer(net, batch=SB, swap_rb=True)
                vs.run_video(u8_h[:2 * SB], chunk=SB, half_size_output=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out_1 = vs.run_video(u8_h, chunk=32, half_size_output=True)
                torch.cuda.synchronize()
                dt_one = time.perf_counter() - t1   # one chunk at a time (rounds 1-5; D = 1: the run below repeats it)
                vs.run_video(u8_h[:4 * SB], chunk=SB, half_size_output=True, in_flight=D)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out_h = vs.run_video(u8_h, chunk=32, half_size_output=True, in_flight=D)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                assert torch.equal(out_1, out_h), "chunks in flight changed the frames"
                del out_1
                # the three activities ALONE on the same clip (upload, the device side with resident frames, download): what the
                # pipelined run hides.  overlap = slowest alone / pipelined wall (1.0 = the other two fully hidden)
                u8_d = u8_h.to(dev)
                small_d = torch.empty((T, 360, 640, 3), dtype=torch.uint8, device=dev)
                down_h = torch.empty((T, 360, 640, 3), dtype=torch.uint8).pin_memory()   # (not out_h: the bf16 comparison below reads it)

                def alone(fn):
                    fn()
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    return time.perf_counter() - t_
                t_up = alone(lambda: u8_d.copy_(u8_h, non_blocking=True))
                t_down = alone(lambda: down_h.copy_(small_d, non_blocking=True))
                t_comp = alone(lambda: vs.run_video(u8_d, chunk=32, half_size_output=True, in_flight=D))
                line["value_720p_stream_u8"] = {"value": round(T / dt, 1), "unit": "frames/s", "n_gpus": 1, "chunks_in_flight": D,
                                                "value_one_in_flight": round(T / dt_one, 1),
                                                "workload": "%d uint8 1280x720 frames, pinned host -> device (2.76 MB per frame) -> gray+"
                                                            "INTER_AREA window planes on the device -> batch %d windows per netG call -> "
                                                            "fused u8 warp -> 2x INTER_AREA -> pinned host (0.69 MB per frame)" % (T, SB),
                                                "pcie_h2d_gb_per_s": round(T * 2.7648e-3 / dt, 2), "pcie_d2h_gb_per_s": round(T * 0.6912e-3 / dt, 2),
                                                "alone_ms": {"upload": round(1e3 * t_up, 1), "device_side_frames_resident": round(1e3 * t_comp, 1),
                                                             "download": round(1e3 * t_down, 1)},
                                                "pcie_alone_gb_per_s": {"h2d": round(T * 2.7648e-3 / t_up, 1), "d2h": round(T * 0.6912e-3 / t_down, 1)},
                                                "pipelined_ms": round(1e3 * dt, 1), "overlap": round(max(t_up, t_comp, t_down) / dt, 3),
                                                "bound": "the fp32 generator (device side alone = %.0f %% of the pipelined wall); PCIe alone would carry %.0f frames/s"
                                                         % (100 * t_comp / dt, T / max(t_up, t_down))}
                del u8_d, small_d, down_h
                assert not out_h.is_cuda and tuple(out_h.shape) == (T, 360, 640, 3)
                if a.math == "fp32":   # the same clip with the generator on the bf16 matrix cores (an extra, never `value`)
                    net.module.set_math("bf16")
                    vs.run_video(u8_h[:4 * SB], chunk=SB, half_size_output=True, in_flight=D)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    out_b = vs.run_video(u8_h, chunk=32, half_size_output=True, in_flight=D)
                    torch.cuda.synchronize()
                    dtb = time.perf_counter() - t1
                    net.module.set_math("fp32")
                    diff = (out_b.to(torch.int16) - out_h.to(torch.int16)).abs()
                    line["value_720p_stream_u8"]["bf16_generator"] = {
                        "value": round(T / dtb, 1), "unit": "frames/s", "pcie_h2d_gb_per_s": round(T * 2.7648e-3 / dtb, 2),
                        "output_gray_levels_vs_fp32": {"max": int(diff.max()), "mean": round(float(diff.float().mean()), 4)}}
                    del out_b
                del u8_h, out_h
            except Exception as e:  # an extra leg must never cost the headline line
                line["value_720p_stream_u8"] = {"error": str(e)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline((x.cpu(), frames.cpu()) + timed_pair if a.math == "fp32" else None)
            line["gpu_over_cpu"] = round(fps / line["cpu_baseline"]["value"], 1)
        if not a.no_extra and a.math == "fp32" and world == 1:
            line["bf16"] = bf16_legs(net, x, frames, out, a, PF, A, synth)
    else:
        line = None
    if (world > 1 and not a.no_extra) or force:
        # configs[3] on this node: 32 item pairs per GPU per step, replicas + ONE flat-bucket RCCL all-reduce of the 48.5 M
        # gradients per step.  Every rank takes part; a watchdog keeps a stuck collective from costing the headline.
        import threading
        done = threading.Event()
        limit = float(os.environ.get("PWS_BENCH_WATCHDOG_S", "300"))   # (the variable is a test hook)
        # rank 0 arrives here after its single-rank legs (rooflines, CPU baseline): until every rank has arrived the deadline
        # covers that section too; from the rendezvous on, `limit` seconds on every rank (the others a little later, so that
        # rank 0's line is out before a non-zero exit of another rank makes the launcher end the job)
        deadline = [time.time() + limit + 900.0]

        def watchdog():
            # a stuck collective must not cost the headline LINE (rank 0 still prints it, with the error), but it is a failure:
            # the process exits non-zero so that no driver mistakes a hung RCCL run for a clean one
            while not done.wait(min(1.0, max(0.001, limit / 4))):
                if time.time() > deadline[0]:
                    if rank == 0:
                        line["training_ddp"] = {"error": "did not finish within %g s (watchdog; exit code 3)" % limit}
                        print(json.dumps(line), file=json_out, flush=True)
                    os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        dist.barrier()
        deadline[0] = time.time() + limit + (0.0 if rank == 0 else 20.0)
        # configs[4] on this node, PCIe inclusive: every rank stabilises its own shard of 128 decoded uint8 720p frames from
        # pinned host memory to pinned host memory (VideoStabilizer.run_video: frames up, window planes + generator + fused warp
        # + 2x down-scale on the device, 640x360 frames down); no collective, the host's PCIe / memory system is shared
        try:
            if world == 1:
                raise StopIteration   # --force-collectives: the training leg only
            from pwstablenet_amd.stream import VideoStabilizer
            Ts = a.stream_frames
            net.module.enable_graph(False)
            net.module.set_math("fp32")
            u8_h = torch.randint(0, 256, (Ts, 720, 1280, 3), dtype=torch.uint8).pin_memory()
            SB = min(32, Ts)   # windows per generator call (tools/stream_sweep.py: 32 is the fastest of 8 / 16 / 32)
            vs = VideoStabilizer(net, batch=SB, swap_rb=True)
            vs.run_video(u8_h[:min(Ts, 4 * SB)], chunk=SB, half_size_output=True, in_flight=D)
            barrier()
            t1 = time.perf_counter()
            out_h = vs.run_video(u8_h, chunk=32, half_size_output=True, in_flight=D)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            barrier()
            t = torch.tensor([dts], device=ctl_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if rank == 0:
                line["value_720p_stream_u8"] = {"value": round(world * Ts / float(t.item()), 1), "unit": "frames/s", "n_gpus": world,
                                                "workload": "%d uint8 1280x720 frames per GPU, pinned host -> device -> window planes, netG fp32 "
                                                            "(batch %d, %d chunks in flight), fused u8 warp, 2x INTER_AREA -> pinned host; "
                                                            "frame-sharded, no collective" % (Ts, SB, D)}
            del u8_h, out_h, vs
        except StopIteration:
            pass
        except Exception as e:
            if rank == 0:
                line["value_720p_stream_u8"] = {"error": str(e)[:300]}
        try:
            from pwstablenet_amd import distributed as D
            dist.barrier()
            NI = a.ddp_items
            grad_bytes = 4 * A.lib().pws_netg_grad_floats(31, 64)
            DDP_REPS, DDP_WARMUP = 10, 3   # per mode: the driver's one 8-GPU run is this leg's only measurement

            def timed(nparts):
                # the exchange lives in the generator's backward: the gradient SLAB (one flat buffer in the kernels' layout) is
                # averaged in place -- whole (nparts = 1: after backward) or range by range as the backward's runs finish layers
                # (nparts = 4: overlapped, on a second stream) -- and unpacked once; no flatten, no copy back
                gs = D.enable_overlapped_grad_sync(net, nparts=nparts, force=force)
                dist.barrier()
                dt_, loss_ = configs2_step_leg(net, dev, NI, "bf16", DDP_REPS, sync=None, seed=500 + rank, warmup=DDP_WARMUP)
                ncoll_, moved_ = gs.collectives, gs.bytes_reduced
                net.module.grad_sync = None
                t_ = torch.tensor([dt_], device=ctl_device, dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                return float(t_.item()), loss_, ncoll_, moved_
            dt, loss, ncoll_after, moved = timed(1)
            dt_ov, _, ncoll, moved_ov = timed(4)
            assert moved == grad_bytes and moved_ov == grad_bytes, (moved, moved_ov, grad_bytes)
            # the step WITHOUT any exchange on the same box (what the exchange adds), and the collective alone, same message
            # sizes, for the xGMI bus-bandwidth figure
            dt_none, _ = configs2_step_leg(net, dev, NI, "bf16", DDP_REPS, sync=None, seed=500 + rank, warmup=DDP_WARMUP)
            flat = torch.zeros(grad_bytes // 4, device=dev)
            D.allreduce_slab(flat, [(0, flat.numel())], force=force)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                D.allreduce_slab(flat, [(0, flat.numel())], force=force)
            torch.cuda.synchronize()
            ar = (time.perf_counter() - t1) / 5
            del flat
            if rank == 0:
                line["training_ddp"] = {
                    "workload": "configs[3]: %d item pairs per GPU per step (%d netG forwards + objective + backward), bf16 math, "
                                "the %.1f MB gradient slab averaged IN PLACE over %s (ReduceOp.AVG on views of the slab, 64 MB messages), "
                                "one unpack, fused Adam; weak scaling"
                                % (NI, 2 * NI, grad_bytes / 1e6, "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (test hook)"),
                    "items_per_gpu_per_step": NI, "reps": DDP_REPS, "warmup": DDP_WARMUP, "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                    "collectives_forced_in_one_rank_group": force,
                    "items_per_s": round(world * NI / min(dt, dt_ov), 1), "ms_per_step": round(1e3 * min(dt, dt_ov), 2),
                    "ms_per_step_allreduce_after_backward": round(1e3 * dt, 2), "collectives_per_step_after_backward": ncoll_after,
                    "ms_per_step_allreduce_overlapped": round(1e3 * dt_ov, 2), "overlapped_collectives_per_step": ncoll,
                    "ms_per_step_no_exchange": round(1e3 * dt_none, 2),
                    "tflops_per_gpu": round(2 * NI / min(dt, dt_ov) * GFLOP_PER_SAMPLE_TRAIN / 1e3, 1), "loss_g_rank0": round(loss, 4),
                    "allreduce_bytes": grad_bytes, "extra_passes_over_the_gradients": 0, "allreduce_alone_ms": round(1e3 * ar, 3),
                    "allreduce_bus_gb_per_s": round(2.0 * (world - 1) / world * grad_bytes / ar / 1e9, 1)}
        except Exception as e:
            if rank == 0:
                line["training_ddp"] = {"error": str(e)[:300]}
        done.set()
    if rank == 0:
        print(json.dumps(line), file=json_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Parity of what bench.py actually TIMES, at BASELINE.json's own sizes:

  * configs[1]: batch 8, ngf 64, fp32, launched as a hipGraph replay with the two-queue schedule -- against the CPU
    restatement of the reference graph (oracle/torch_ref.py) on the same inputs, bound 5e-4 on the warp field (normalised
    coordinates) and 1e-3 on the warped frames scaled to [-1, 1] (north_star); graph replays on changed input contents,
    on another input address, after the eager arena was replaced by other calls; returned tensors are fresh;
  * configs[2]: 32 item pairs = 64 generator forwards per step, bf16 math + bf16 storage, ``objective.train_step`` -- loss
    and all 48.5 M gradient entries against the fp32 HIP path of the same step, and the same step at one item pair against
    torch-CPU autograd through the restated generator + objective (oracle/torch_ref.py, oracle/objective_ref.py).

Tile shape and split-K of every conv launch depend on the batch, so these sizes run kernel selections the N=2 goldens do not.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from pwstablenet_amd import synth  # noqa: E402

FIELD_TOL = 5e-4
WARP_TOL = 1e-3          # frames scaled to [-1, 1]
# Tripwire UNDER the bound (round-5 review): the saturating weights W2 at the timed batch stood at 0.77e-3 in round 3, 0.89e-3 after the first
# layer went Winograd -- each multiply-count reduction spends margin, and the first cut of F(2x2,5x5) had silently broken the bound (1.11e-3).
# A change that moves W2 past this line fails HERE, at the commit that makes it: run tools/parity_budget.py (error per kernel family,
# profiles/r06_parity_budget.txt), recover the margin or justify the new row before touching this number.
WARP_TRIPWIRE_W2 = 0.92e-3


def make_net(kind="W1", ngf=64):
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, ngf, "normal", 0.02)
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights(kind, seed=123, ngf=ngf)})
    return net.cuda()


@pytest.fixture(scope="module", params=["W1", "W2"])
def cpu_ref8(request):
    """The CPU path on bench.py's own rank-0 inputs (batch 8): ~1.5 s on the box's host cores.  W1: the bench's weights; W2:
    saturating residuals and a field that leaves [-1, 1] (out-of-range taps) at the timed batch size."""
    from oracle import torch_ref
    params = [torch.from_numpy(v) for _, v in synth.make_weights(request.param, seed=123, ngf=64)]
    x = torch.from_numpy(synth.noise_window(8, 31, 256, seed=123))
    fr = torch.from_numpy(synth.make_frames(8, 3, 256, 256, seed=321))
    with torch.no_grad():
        field = torch_ref.netg_forward(params, x, is_training=False)
        warped = torch.nn.functional.grid_sample(fr, field, mode="bilinear", padding_mode="zeros", align_corners=False)
    return x, fr, field, warped, request.param


def test_configs1_graph_two_queue_path_vs_cpu_oracle(hip, cpu_ref8):
    from pwstablenet_amd import functional as PF
    x_h, fr_h, ref_field, ref_warp, kind = cpu_ref8
    net = make_net(kind)
    x, fr = x_h.cuda(), fr_h.cuda()
    assert hip.lib().pws_get_option(hip.OPT_TWO_QUEUES) == 1   # the product default, what the bench times
    with torch.no_grad():
        eager = net(x, False).clone()
        net.module.enable_graph(True)
        f1 = net(x, False)            # capture + first replay
        f2 = net(x, False)            # replay
        w2 = PF.grid_sample(fr, f2)
    assert net.module._graph is not None and not net.module._graph["static"]
    ferr = float((f2.cpu() - ref_field).abs().max())
    werr = float((w2.cpu() - ref_warp).abs().max()) / 127.5
    print("configs[1] graph + two queues, batch 8, %s: field max-abs err %.3g, warped-frame err %.3g vs the CPU path" % (kind, ferr, werr))
    if kind == "W2":
        assert float(ref_field.abs().max()) > 1.2   # the case is what it claims: taps outside the frame
    assert ferr < FIELD_TOL and werr < WARP_TOL
    if kind == "W2":
        assert werr < WARP_TRIPWIRE_W2, ("warped-frame error %.3g on W2 is inside the bound but past the tripwire %.3g: the parity margin is being "
                                         "spent -- tools/parity_budget.py attributes it per kernel family" % (werr, WARP_TRIPWIRE_W2))
    assert torch.equal(f1, eager) and torch.equal(f2, eager)   # no atomics, ordered split-K sums: the schedule changes nothing
    assert f1.data_ptr() != f2.data_ptr()                      # fresh tensors: a later call does not overwrite f1


def test_graph_replay_on_changed_contents_other_address_and_alias_mode(hip):
    net = make_net()
    xa = torch.from_numpy(synth.noise_window(8, 31, 256, seed=5)).cuda()
    xb = torch.from_numpy(synth.make_window(8, 31, 256, seed=6)).cuda()
    with torch.no_grad():
        ea, eb = net(xa, False).clone(), net(xb, False).clone()
        assert float((ea - eb).abs().max()) > 1e-3
        net.module.enable_graph(True)
        x = xa.clone()
        ga = net(x, False)
        x.copy_(xb)                       # same address, new contents: the replay must read them
        gb = net(x, False)
        assert torch.equal(ga, ea) and torch.equal(gb, eb)
        g = net.module._graph["g"]
        gc = net(xa, False)               # another address: the graph switches to a private static input, once
        assert net.module._graph["static"] and net.module._graph["g"] is not g
        g = net.module._graph["g"]
        gd = net(xb, False)
        ge = net(xa.clone(), False)
        assert net.module._graph["g"] is g, "no re-capture per call"
        assert torch.equal(gc, ea) and torch.equal(gd, eb) and torch.equal(ge, ea)
        assert torch.equal(xa, torch.from_numpy(synth.noise_window(8, 31, 256, seed=5)).cuda())   # inputs never mutated
        # opt-in aliasing: the graph's own output buffer, overwritten by the next call
        net.module.enable_graph(True, alias_output=True)
        ha = net(xa, False)
        keep = ha.clone()
        hb = net(xb, False)
        assert ha.data_ptr() == hb.data_ptr() and torch.equal(hb, eb) and not torch.equal(ha, keep)


def test_graph_follows_a_weight_update_without_recapture(hip):
    """the packed buffer is re-packed in place at a fixed address, so an optimizer step / load_state_dict needs no
    re-capture -- and the replay must compute with the NEW weights."""
    net = make_net("W1", 32)
    x = torch.from_numpy(synth.make_window(4, 31, 256, seed=4)).cuda()
    with torch.no_grad():
        net.module.enable_graph(True)
        a = net(x, False)
        g = net.module._graph["g"]
        net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in synth.make_weights("W2", seed=7, ngf=32)})
        b = net(x, False)
        assert net.module._graph["g"] is g, "a weight update must not re-capture"
        net.module.enable_graph(False)
        want = net(x, False)
    assert torch.equal(b, want) and float((a - b).abs().max()) > 1e-3


def test_graph_survives_replacement_of_the_eager_arena(hip):
    """the graph used to bake in the address of the shared arena, which the next eager call with another
    (batch, is_training) dropped.  The graph owns its arena now."""
    net = make_net("W2", 32)
    x8 = torch.from_numpy(synth.make_window(8, 31, 256, seed=1)).cuda()
    x3 = torch.from_numpy(synth.make_window(3, 31, 256, seed=2)).cuda()
    with torch.no_grad():
        want = net(x8, False).clone()
        net.module.enable_graph(True)
        a = net(x8, False)
        graph = net.module._graph["g"]
        # calls that replace the eager arena: the training outputs, another batch size, and (grad mode on) the no-grad inference
        net(x3)
        filler = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(4)]   # lands in whatever was freed
    y = net(x3, False)                   # grad mode on, graph mode does not apply (autograd Function)
    assert y.shape[0] == 3
    with torch.no_grad():
        b = net(x8, False)
        assert net.module._graph["g"] is graph
    del filler
    assert torch.isfinite(b).all() and torch.equal(a, want) and torch.equal(b, want)


def test_backward_after_parameter_update_raises(hip):
    """the packed weights are one shared buffer; a backward after optimizer.step() must not silently use new weights."""
    from pwstablenet_amd.optim import Adam
    net = make_net("W1", 16)
    x = torch.from_numpy(synth.make_window(2, 31, 256, seed=3)).cuda()
    opt = Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.999))
    g1, _ = net(x)
    g2, _ = net(x)                              # the reference's order: two forwards, one backward, then the step
    (g1[2].mean() + g2[2].mean()).backward()
    opt.step()
    g3, _ = net(x)
    opt.step()                                  # parameters move between a forward and its backward
    with pytest.raises(RuntimeError, match="modified"):
        g3[2].mean().backward()


def test_align_corners_switch_matches_torch_legacy_convention(hip):
    """checkpoints trained on the reference's pinned torch 0.4 expect align_corners=True in affine_grid /
    grid_sample; the generator's fused affine grid follows ``UnetGenerator.align_corners``."""
    from oracle import torch_ref
    net = make_net("W2", 16)
    params = [torch.from_numpy(v) for _, v in synth.make_weights("W2", seed=123, ngf=16)]
    x_h = torch.from_numpy(synth.make_window(2, 31, 256, seed=9))
    with torch.no_grad():
        net.module.align_corners = True
        got = net(x_h.cuda(), False).cpu()
        net.module.enable_graph(True)
        got_g = net(x_h.cuda(), False).cpu()
        net.module.enable_graph(False)
        net.module.align_corners = False
        got0 = net(x_h.cuda(), False).cpu()
        want = torch_ref.netg_forward(params, x_h, is_training=False, align_corners=True)
        want0 = torch_ref.netg_forward(params, x_h, is_training=False, align_corners=False)
    assert float((got - want).abs().max()) < FIELD_TOL and torch.equal(got, got_g)
    assert float((got0 - want0).abs().max()) < FIELD_TOL
    assert float((got - got0).abs().max()) > 1e-3   # half a pixel at 256 px = 1/256 in normalised coordinates


# ---------------------------------------------------------------------------------------------- configs[2]
class _NoStep:
    def __init__(self, net):
        self.net = net

    def zero_grad(self):
        self.net.zero_grad(set_to_none=True)

    def step(self):
        pass


def _grad_stats(ga, gb):
    dot = sum((a.double() * b.double()).sum().item() for a, b in zip(ga, gb))
    na = sum((a.double() ** 2).sum().item() for a in ga) ** 0.5
    nb = sum((b.double() ** 2).sum().item() for b in gb) ** 0.5
    return dot / (na * nb), na / nb


def test_configs2_bf16_batch32_train_step_vs_fp32_path(hip):
    """The configs[2] step exactly as bench.py runs it (32 item pairs -> one batch of 64 windows, bf16 math + storage, fused
    objective, backward) against the same step on the fp32 HIP path: stated bf16 tolerance of the STEP -- loss within 1 %,
    cosine of the whole gradient > 0.98, norm ratio within 5 %."""
    from pwstablenet_amd.objective import LOSS_NAMES, StabObjective, train_step
    n = 32
    net = make_net("W1", 64)
    small = synth.make_train_batch(4, seed=500)
    batch = [torch.from_numpy(t).repeat((n // 4,) + (1,) * (t.ndim - 1)).cuda() for t in small]
    obj = StabObjective(batchSize=n)
    res = {}
    for math in ("fp32", "bf16"):
        net.module.set_math(math)
        assert net.module.store == ("bf16" if math == "bf16" else "fp32")
        out = train_step(net, _NoStep(net), batch, obj)
        res[math] = ({k: float(out[k]) for k in LOSS_NAMES}, [p.grad.clone() for p in net.parameters()])
        assert all(torch.isfinite(g).all() for g in res[math][1])
    l32, l16 = res["fp32"][0], res["bf16"][0]
    cos, ratio = _grad_stats(res["bf16"][1], res["fp32"][1])
    print("configs[2] step, 64 forwards: loss_g bf16 %.6g vs fp32 %.6g; gradient cosine %.5f, norm ratio %.4f" % (
        l16["loss_g"], l32["loss_g"], cos, ratio))
    for k in ("loss_g", "loss_mse", "loss_g2", "loss_feature"):
        assert abs(l16[k] - l32[k]) <= 1e-2 * abs(l32[k]) + 1e-6, (k, l16[k], l32[k])
    assert cos > 0.98 and abs(ratio - 1) < 0.05, (cos, ratio)


def test_configs2_train_step_vs_torch_cpu_autograd(hip):
    """One item pair (two forwards batched as one) through train_step, fp32 and bf16, against torch-CPU autograd through the
    restated generator and objective: fp32 losses 5e-5 / gradient cosine > 0.9999; bf16 loss 1 % / cosine > 0.98."""
    from oracle import objective_ref as R
    from oracle import torch_ref
    from pwstablenet_amd.objective import LOSS_NAMES, StabObjective, train_step
    n = 1
    small = synth.make_train_batch(n, seed=77)
    params = [torch.from_numpy(v).requires_grad_(True) for _, v in synth.make_weights("W1", seed=123, ngf=64)]
    im1, f1, _a1, im2, f2, _a2, adj = [torch.from_numpy(t) for t in small]
    ns1, nu1, fs1, fu1 = R.pre_processing(im1, f1.float())
    ns2, nu2, fs2, fu2 = R.pre_processing(im2, f2.float())
    g1, r1 = torch_ref.netg_forward(params, nu1[:, :31])
    g2, r2 = torch_ref.netg_forward(params, nu2[:, :31])
    ref = R.objective(g1, r1, g2, r2, nu1, ns1, fs1, fu1, nu2, ns2, fs2, fu2, adj.float(), n)
    ref["loss_g"].backward()
    want = [p.grad for p in params]
    net = make_net("W1", 64)
    batch = [torch.from_numpy(t).cuda() for t in small]
    obj = StabObjective(batchSize=n)
    for math, ltol, ctol in (("fp32", 5e-5, 0.9999), ("bf16", 1e-2, 0.98)):
        net.module.set_math(math)
        out = train_step(net, _NoStep(net), batch, obj)
        got = [p.grad.cpu() for p in net.parameters()]
        cos, ratio = _grad_stats(got, want)
        print("train_step (%s) vs torch-CPU autograd: loss_g %.6g vs %.6g, gradient cosine %.6f, norm ratio %.4f" % (
            math, float(out.loss_g), float(ref["loss_g"]), cos, ratio))
        for k in LOSS_NAMES:
            assert abs(float(out[k]) - float(ref[k])) <= ltol * abs(float(ref[k])) + 1e-6, (math, k, float(out[k]), float(ref[k]))
        assert cos > ctol, (math, cos)


def test_bench_self_launches_its_ranks(hip):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself; on this one-GPU box both sit on
    cuda:0 and the control plane goes through gloo (test hooks), the data path is the real one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PWS_BENCH_ONE_DEVICE="1", PWS_BENCH_BACKEND="gloo")
    # the N>1 legs included, at a small size: the configs[3] step with both gradient-exchange modes (training_ddp) and the
    # frame-sharded streaming leg
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-prof", "--gs-batch", "8", "--ddp-items", "4", "--stream-frames", "32"], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 0 and "gloo" in line["control_plane"], line
    assert line["value"] > 0 and line["config"]["frames_per_gpu_per_step"] == 8
    ddp = line["training_ddp"]
    assert "error" not in ddp, ddp
    assert ddp["items_per_gpu_per_step"] == 4 and ddp["items_per_s"] > 0 and ddp["overlapped_collectives_per_step"] > 0
    # the gradient slab: the 48 535 944 gradients in the kernels' layout (input channels padded to 16, 64-float alignment), exchanged
    # in place -- exactly once, no flatten / copy-back passes
    assert 4 * 48535944 <= ddp["allreduce_bytes"] == 4 * hip.lib().pws_netg_grad_floats(31, 64) <= 4 * 48535944 * 1.001 and "gloo" in ddp["workload"]
    assert ddp["extra_passes_over_the_gradients"] == 0 and ddp["collectives_per_step_after_backward"] > 0
    assert line["value_720p_stream_u8"]["value"] > 0 and line["value_720p_stream_u8"]["n_gpus"] == 2


def test_bench_watchdog_exits_non_zero(hip):
    """A hung collective in the N>1 legs must not look like a clean run: the watchdog still prints the
    headline line (with the error) but the job exits non-zero.  Forced here by a watchdog limit of 10 ms."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PWS_BENCH_ONE_DEVICE="1", PWS_BENCH_BACKEND="gloo", PWS_BENCH_WATCHDOG_S="0.01")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-prof", "--gs-batch", "8", "--ddp-items", "4", "--stream-frames", "32"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode != 0
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] > 0 and "watchdog" in line["training_ddp"]["error"]


@pytest.mark.parametrize("math", ["fp32", "bf16"])
def test_two_batches_in_flight_on_two_streams(hip, math):
    """bench.py --in-flight 2 (the headline's launch path since round 6): consecutive steps alternate over two streams without waiting for each
    other.  The generator keeps one graph + arena per STREAM, so the two forwards in flight do not share activations: every step returns the bits
    of the one-at-a-time forward of its own input, no stream re-captures after its first call, and eager launches behave the same."""
    from pwstablenet_amd import functional as PF
    net = make_net("W2")
    net.module.set_math(math)
    xs = [torch.from_numpy(synth.noise_window(8, 31, 256, seed=40 + i)).cuda() for i in range(2)]
    fr = torch.from_numpy(synth.make_frames(8, 3, 256, 256, seed=7)).cuda()
    with torch.no_grad():
        want = [net(x, False).clone() for x in xs]
        want_w = [PF.grid_sample(fr, w) for w in want]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
    net.module.two_queues = False   # one queue per forward: two forwards forking into the device's one side queue would serialise
    try:
        for graph in (True, False):
            net.module.enable_graph(graph, per_stream=True)
            outs, graphs = [], {}
            with torch.no_grad():
                for i in range(8):
                    with torch.cuda.stream(lanes[i % 2]):
                        f = net(xs[i % 2], False)
                        outs.append((i % 2, f, PF.grid_sample(fr, f)))
                    if graph:
                        g = net.module._graph["g"]
                        assert graphs.setdefault(i % 2, g) is g, "a stream re-captured its graph"
            torch.cuda.synchronize()
            if graph:
                assert len(net.module._graphs) == 2 and graphs[0] is not graphs[1]
            for k, f, w in outs:
                assert torch.equal(f, want[k]) and torch.equal(w, want_w[k]), (graph, k)
    finally:
        net.module.two_queues = None
        net.module.enable_graph(False)
        net.module.set_math("fp32")


def test_kernel_events_enabled_across_a_graph_capture(hip):
    """pws_prof_enable(1) while a forward is CAPTURED (bench.py --no-prof used to reach this on its first 720p step after leaving the
    in-flight mode): the captured launches record nothing (an event recorded into a capture is a graph node, not a timestamp), the launches
    outside the graph are timed as before, and the collection does not fail -- nor would a failed one poison the next."""
    from pwstablenet_amd import functional as PF, hipabi as A
    net = make_net("W1")
    x = torch.from_numpy(synth.noise_window(2, 31, 256, seed=3)).cuda()
    fr = torch.from_numpy(synth.make_frames(2, 3, 256, 256, seed=4)).cuda()
    with torch.no_grad():
        want = net(x, False).clone()
        net.module.enable_graph(True)
        try:
            A.lib().pws_prof_enable(1)
            got = [PF.grid_sample(fr, net(x, False)) for _ in range(3)]   # capture, replay, replay
            torch.cuda.synchronize()
            A.lib().pws_prof_enable(0)
            recs = A.prof_collect()
            # (the capture's eager warm-up forward is timed like any eager launch; the captured and the replayed launches are not)
            assert sum(r[0] == "grid_sample_fwd_kernel" for r in recs) == 3 and all(r[4] > 0 for r in recs)
            eager_forward = len(recs) - 3
            assert 0 < eager_forward < 120, eager_forward
            assert torch.equal(net(x, False), want)
            assert all(torch.equal(g, got[0]) for g in got)
            assert A.prof_collect() == []
        finally:
            A.lib().pws_prof_enable(0)
            net.module.enable_graph(False)

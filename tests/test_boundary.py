"""CPU tests of the drop-in boundary: config surface, state-dict keys, C-ABI symbols, loud failure without a GPU."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_meta.json")))


def test_cfg_matches_reference_defaults():
    """Every flag of the reference's lib/cfg.py:7-39 with the same type and default; constants too."""
    from pwstablenet_amd.lib import cfg
    for name, (typ, default) in META["cfg_defaults"].items():
        assert hasattr(cfg.opt, name), name
        v = getattr(cfg.opt, name)
        assert type(v).__name__ == typ, (name, type(v).__name__, typ)
        assert v == default, (name, v, default)
    assert cfg.REFERENCE_FLAGS and set(cfg.REFERENCE_FLAGS) == set(META["cfg_defaults"])
    c = META["cfg_constants"]
    assert cfg.period == c["period"]
    assert cfg.index_sample.tolist() == c["index_sample"]
    assert cfg.index_sample_discriminator.tolist() == c["index_sample_discriminator"]
    assert (cfg.train_files, cfg.val_files, cfg.test_files) == (c["train_files"], c["val_files"], c["test_files"])
    for name in ("opt", "period", "index_sample", "index_sample_discriminator", "train_files", "val_files", "test_files",
                 "np", "cudnn", "parser", "argparse"):
        assert hasattr(cfg, name)
    # legacy main.py flags accepted
    for name in ("gpu_ids", "dataset", "dir_logs", "train", "testBatchSize", "start_loss_affine", "affine_weight"):
        assert hasattr(cfg.opt, name)


def test_cfg_survives_foreign_argv_and_parses_flags():
    code = ("import sys; sys.argv=['prog','--ngf','32','--mode','train','--unknown-flag','7'];"
            "from pwstablenet_amd.lib.cfg import opt; print(opt.ngf, opt.mode, opt.batchSize)")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode().split()
    assert out == ["32", "train", "16"]


def test_state_dict_keys_shapes_order_match_reference():
    import torch
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, 64, "normal", 0.02)
    got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    assert got == META["state_dict"]
    assert sum(p.numel() for p in net.parameters()) == META["num_params"] == 48535944
    # legacy 6th positional argument of main.py:28
    net2 = define_G(31, 2, 16, "xavier", 0.02, [0])
    assert len(net2.state_dict()) == 92
    with pytest.raises(NotImplementedError):
        define_G(31, 2, 16, "nope", 0.02)
    # init rule 'normal': weights ~ N(0, 0.02), biases 0 (reference :29-40)
    w = net.module.down4.mpconv[0].weight
    assert abs(float(w.std()) - 0.02) < 1e-3 and float(net.module.down4.mpconv[0].bias.abs().max()) == 0.0
    assert isinstance(net.module.down4.mpconv[1], torch.nn.LeakyReLU)


def test_product_path_fails_loudly_without_gpu():
    import torch
    from pwstablenet_amd import functional as PF
    from pwstablenet_amd.lib.networks_cascading import define_G
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = define_G(31, 2, 16, "normal", 0.02)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 31, 256, 256), False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PF.grid_sample(torch.zeros(1, 3, 8, 8), torch.zeros(1, 8, 8, 2))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pwstablenet_amd/ may import, load or execute it."""
    for dp, _, files in os.walk(os.path.join(ROOT, "pwstablenet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(dp, f)
                assert "pws_oracle" not in txt and "libpws_oracle" not in txt, os.path.join(dp, f)


def test_c_abi_exports_every_declared_symbol():
    """libpwstable_hip.so loads (no GPU needed) and exports every function include/pwstable.h declares."""
    from pwstablenet_amd import build, hipabi
    build.build(force=False, verbose=False)
    hdr = open(os.path.join(ROOT, "include", "pwstable.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pws_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = hipabi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing export " + name
    assert declared == set(hipabi.SIGNATURES), declared ^ set(hipabi.SIGNATURES)
    assert L.pws_version() == 5
    assert L.pws_netg_packed_floats(31, 64) >= 48535944
    assert L.pws_netg_workspace_bytes(8, 31, 64, 0) > 10 ** 9
    assert L.pws_packed_weight_floats(hipabi.CONVT_K4S2, 1024, 256) == 16 * 1024 * 256
    assert L.pws_conv2d_fwd(None, None) == -22 and b"NULL" in L.pws_last_error()


def test_dropin_shims_resolve():
    code = ("import sys; sys.argv=['x']; sys.path[:0]=[%r, %r];"
            "from lib.networks_cascading import define_G, define_D, GANLoss; from lib.cfg import opt, period, index_sample;"
            "import lib.networks_cascading as m; print(m.define_G.__module__, opt.input_nc, period)") % (
        ROOT, os.path.join(ROOT, "dropin"))
    out = subprocess.check_output([sys.executable, "-c", code], cwd="/tmp").decode().split()
    assert out[-3:] == ["pwstablenet_amd.lib.networks_cascading", "31", "30"]


def test_torch_ref_matches_reference_golden(netg_golden):
    """The PyTorch-CPU restatement used as the host-CPU baseline reproduces the reference's outputs."""
    import torch
    from oracle import torch_ref
    from pwstablenet_amd import synth
    torch.set_num_threads(8)
    params = [torch.from_numpy(v) for _, v in synth.make_weights("W2", seed=123, ngf=16)]
    x = torch.from_numpy(synth.make_window(1, 31, 256, seed=123))
    with torch.no_grad():
        grids, resid = torch_ref.netg_forward(params, x, True)
        g_inf = torch_ref.netg_forward(params, x, False)
    assert torch.equal(g_inf, grids[2])
    # (1e-5: the goldens were made in the build container; another host CPU's oneDNN sums in another order -- 1.7e-6 on an EPYC 9575F)
    np.testing.assert_allclose(g_inf.numpy(), netg_golden["W2_g16_grid2_full"], rtol=0, atol=1e-5)


def test_no_tracked_binaries_in_the_package():
    """History stays source-only: no tracked file under pwstablenet_amd/ (or anywhere else but the .npz fixtures) is an ELF
    object, an offload bundle or an extracted code object (44 of those once slipped past the *.so rule)."""
    try:
        tracked = subprocess.check_output(["git", "ls-files"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().split("\n")
    except (subprocess.CalledProcessError, FileNotFoundError):
        pytest.skip("not a git checkout")
    bad = []
    for rel in filter(None, tracked):
        p = os.path.join(ROOT, rel)
        if not os.path.isfile(p):
            continue
        if "hipv4" in rel or ".host-x86_64" in rel or re.search(r"\.(so|o|a|hsaco|co)(\.\d+)*$", rel):
            bad.append(rel)
            continue
        with open(p, "rb") as fh:
            head = fh.read(24)
        if head[:4] == b"\x7fELF" or head.startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
            bad.append(rel)
    assert not bad, bad


def test_no_kernel_spills_to_scratch():
    """hipcc's per-kernel resource report (written by pwstablenet_amd.build): accumulators or prefetch registers that end
    up in scratch make a kernel several times slower without failing any numerical test (seen once: a by-reference
    accumulator array put 732 B per lane in scratch).  Known small spills are listed with their budget."""
    from pwstablenet_amd import build
    build.build(verbose=False)
    res = build.kernel_resources()
    assert {"conv_mfma.hip", "conv_bf16.hip", "wgrad_bf16.hip", "conv_wino.hip", "grid_sample.hip"} <= set(res)
    allowed = {  # substring of the mangled name -> bytes per lane
        "conv_mfma_kernelINS_7ConvCfgILi5ELi1ELi2ELi0ELi16ELi16ELi1ELi8ELi4ELi1ELi2ELi2ELb1": 64,   # first layer, NCHW staging
        "wgrad_mfma_kernelINS_5WgCfgILi3ELi2ELi1ELi0ELi2ELi2ELi16ELi9": 64,
        "wino_kernelILi1E": 64,                                                                      # F(3x3,2x2), opt-in only
        # ring Winograd, 16-wide geometry (6 raw DMA pieces for wave 0): hipcc hoists the lane decode of the once-per-unit offset
        # computation out of the slot loop and parks 8-11 dwords of it in scratch; they are reloaded once per unit, outside the matrix phase
        "wino_ring_kernelILi0ELi1E": 64,
        "wino_ring_kernelILi1ELi1E": 64,
    }
    bad = []
    for src, kernels in res.items():
        for name, r in kernels.items():
            budget = max([b for k, b in allowed.items() if k in name] + [0])
            if r.get("scratch", 0) > budget:
                bad.append((src, name, r))
    assert not bad, bad


def test_use_bn_variant_state_dict_matches_reference_and_training_raises():
    """use_BN=True (lib/cfg.py:37): same module tree / state-dict keys as the reference built with --use_BN 1
    (tests/golden/netg_bn.npz holds the reference's key list); training-mode BatchNorm is refused loudly."""
    import numpy as np
    import torch
    from pwstablenet_amd.lib.networks_cascading import SingleDeviceParallel, UnetGenerator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netg_bn.npz"))
    net = SingleDeviceParallel(UnetGenerator(31, 2, 16, use_BN=True))
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    with pytest.raises((NotImplementedError, RuntimeError)):
        net.train()(torch.zeros(1, 31, 256, 256))


def test_checkpoint_files_in_the_reference_format(tmp_path):
    """main_new.py:433-443 / :469-471: ``{'net': state_dict (module.-prefixed), 'epoch': n}`` under
    checkpoint/<dir>/netG_model_epoch_<n>.pth.  A file built from the REFERENCE's key / shape list (tests/golden/reference_meta.json)
    loads strictly; a file written here holds exactly those keys, on the host; un-prefixed state dicts are re-keyed."""
    import torch
    from pwstablenet_amd.checkpoint import checkpoint_path, load_checkpoint, save_checkpoint
    from pwstablenet_amd.lib.networks_cascading import define_G
    net = define_G(31, 2, 64, "normal", 0.02)
    g = torch.Generator().manual_seed(0)
    ref_state = {k: torch.randn(shape, generator=g) * 0.01 for k, shape in META["state_dict"]}   # what the authors' file holds
    ref_file = tmp_path / "netG_model_epoch_35.pth"
    torch.save({"net": ref_state, "epoch": 35}, ref_file)
    assert load_checkpoint(net, str(ref_file)) == 35
    for k, v in net.state_dict().items():
        assert torch.equal(v, ref_state[k]), k
    path = save_checkpoint(net, 36, "unet_512", root=str(tmp_path / "checkpoint"))
    assert path == checkpoint_path("unet_512", 36, root=str(tmp_path / "checkpoint")) and path.endswith("checkpoint/unet_512/netG_model_epoch_36.pth")
    back = torch.load(path, weights_only=True)
    assert set(back) == {"net", "epoch"} and back["epoch"] == 36
    assert [[k, list(v.shape)] for k, v in back["net"].items()] == META["state_dict"]
    assert all(not v.is_cuda for v in back["net"].values())
    bare = tmp_path / "bare.pth"
    torch.save({k[len("module."):]: v + 1 for k, v in ref_state.items()}, bare)       # saved from netG.module
    assert load_checkpoint(net, str(bare)) is None
    assert torch.equal(net.state_dict()["module.out.mpconv.0.bias"], ref_state["module.out.mpconv.0.bias"] + 1)
    broken = dict(ref_state)
    broken.pop("module.linear.mpconv.0.bias")
    torch.save({"net": broken, "epoch": 1}, bare)
    with pytest.raises(RuntimeError):
        load_checkpoint(net, str(bare))


def test_gradient_slab_layout_is_consistent():
    """pws_netg_grad_floats / pws_netg_grad_layout (ABI 4, host-side functions: no GPU needed): the 46 layers' ranges abut, cover the slab,
    and each holds its weight gradient in the forward packed layout + its bias gradient; the slab is the 48.5 M gradients plus alignment
    only (what a data-parallel host all-reduces in place instead of nn.DataParallel's reduce, lib/networks_cascading.py:51-52)."""
    import ctypes
    from pwstablenet_amd import hipabi as A
    from pwstablenet_amd.spec import layer_specs
    L = A.lib()
    for ngf in (16, 64):
        specs = layer_specs(31, 2, ngf)
        nl = len(specs)
        assert nl == 46
        first, count = (ctypes.c_size_t * nl)(), (ctypes.c_size_t * nl)()
        assert L.pws_netg_grad_layout(31, ngf, first, count) == 0
        total = L.pws_netg_grad_floats(31, ngf)
        assert first[0] == 0 and first[nl - 1] + count[nl - 1] == total
        nparam = 0
        for i, ls in enumerate(specs):
            if i + 1 < nl:
                assert first[i] + count[i] == first[i + 1]        # consecutive layers abut: finished layers merge into few large messages
            w = ls.cin * ls.cout * ls.k * ls.k
            assert count[i] >= w + ls.cout and count[i] % 64 == 0  # (input channels padded to 16, 64-float alignment)
            nparam += w + ls.cout
        assert nparam <= total <= nparam * 1.02 + 64 * 2 * nl
        assert total < L.pws_netg_packed_floats(31, ngf)            # (the weight buffer also holds the Winograd / bf16 copies)
    assert L.pws_netg_grad_floats(31, 60) == 0                      # ngf must be a multiple of 16
    assert L.pws_netg_grad_layout(31, 64, None, None) == -22


def test_train_step_never_guesses_the_gradient_world_in_a_multi_rank_group():
    """objective.resolve_grad_world (round-5 review): the sum-type shape term's gradient is scaled by the number of ranks the gradients are
    AVERAGED over.  A step that knows its exchange takes the group's size; no group means 1; a multi-rank group with NO exchange this step
    can see (a wrapper averaging behind its back) must come out as None, so that StabObjective raises instead of scaling for a world of one."""
    import torch

    from pwstablenet_amd.objective import resolve_grad_world

    class Net(torch.nn.Module):
        grad_sync = None

    class Wrapped(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m

    plain, hooked = Net(), Net()
    hooked.grad_sync = object()
    sync = lambda params: None  # noqa: E731
    assert resolve_grad_world(plain, None, None, None) == 1            # no process group
    assert resolve_grad_world(plain, None, None, 1) == 1               # a group of one
    assert resolve_grad_world(plain, None, None, 8) is None            # eight ranks, nothing known to exchange: refuse to guess
    assert resolve_grad_world(Wrapped(plain), None, None, 8) is None   # ... also behind a DataParallel-style wrapper
    assert resolve_grad_world(plain, None, sync, 8) == 8               # train_step's own sync_gradients averages over the group
    assert resolve_grad_world(Wrapped(hooked), None, None, 8) == 8     # the overlapped exchange attached to the generator
    assert resolve_grad_world(plain, None, sync, None) == 1
    assert resolve_grad_world(plain, 4, None, 8) == 4                  # the objective's own setting wins
    assert resolve_grad_world(plain, 1, None, 8) == 1


def test_documents_cite_files_that_exist():
    """DESIGN.md, README.md, INTEGRATION.md and profiles/README.md name tools, tests, sources and profile files in backticks; a file that a clean-up
    removed must not stay cited (round 6 removed 27 probe scripts).  docs/ROUNDS.md is history ("as written at the time") and is not checked."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        with open(os.path.join(root, doc)) as fh:
            txt = fh.read()
        for ref in set(re.findall(r"`((?:tools|tests|pwstablenet_amd|oracle|profiles|include|dropin|docs)/[A-Za-z0-9_./\-]+)`", txt)):
            ref = ref.rstrip(".,")
            if not os.path.exists(os.path.join(root, ref)):
                missing.append((doc, ref))
    assert not missing, missing

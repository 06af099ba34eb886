"""CPU-only tests: the C-ABI library loads and exports what include/bflow_hip.h declares, host-side logic (config
composition, parameter inventory / checkpoint compatibility, sharding, the N>1 exchange step over gloo), and that the
product path refuses to run without a GPU.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import bflow_amd
from bflow_amd import configs, dist as bdist, hip
from bflow_amd.bezier import BezierCurves
from bflow_amd.corr import CorrComputation
from bflow_amd.weights import deterministic_state_dict
from oracle import raft_spline_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "bflow_hip.h")).read()
    declared = set(re.findall(r"\b(bflow_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = ctypes.CDLL(hip.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib().bflow_version() == hip.ABI_VERSION == 2
    # only the C ABI is exported (built with -fvisibility=hidden)
    syms = subprocess.run(["nm", "-D", "--defined-only", hip.library_path()], stdout=subprocess.PIPE, text=True).stdout
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    assert exported == declared, exported ^ declared


def test_host_side_abi_functions_and_error_reporting():
    np.testing.assert_array_equal(hip.bezier_coeffs([0.25, 0.5, 0.75, 1.0], 2), O.bezier_coeffs([0.25, 0.5, 0.75, 1.0], 2).astype(np.float32))
    np.testing.assert_array_equal(hip.bezier_coeffs([0.2, 0.4, 0.6, 0.8, 1.0, 1], 10), O.bezier_coeffs([0.2, 0.4, 0.6, 0.8, 1.0, 1], 10).astype(np.float32))
    with pytest.raises(hip.BflowHipError, match="outside"):
        hip.bezier_coeffs([1.5], 2)
    with pytest.raises(hip.BflowHipError, match="degree"):
        hip.bezier_coeffs([0.5], 99)
    # argument validation of the training entry points happens before any launch: callable without a GPU
    L = hip.lib()
    assert L.bflow_corr_pool2x2_bwd(None, None, 1, 4, 4, None) != 0 and b"corr_pool2x2_bwd" in L.bflow_last_error_string()
    assert L.bflow_cvx_upsample_bwd(None, None, None, None, None, None, 1, 4, 2, 2, None) != 0
    assert L.bflow_l1_masked_accumulate(None, None, None, 1, 2, 16, None, None) != 0 and b"l1_masked_accumulate" in L.bflow_last_error_string()
    assert L.bflow_corr_lookup_bwd(None, None, 0, None, 1, None, None, 1, 4, 4, None) != 0


def test_training_mode_has_no_cpu_fallback_either():
    m = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).train()
    with pytest.raises(hip.BflowHipError):
        m(voxel_grid=torch.zeros(1, 9, 64, 64), iters=1, test_mode=False)


def test_no_cpu_fallback():
    cfg = configs.model_config("E_LU4_BD2")
    m = bflow_amd.RAFTSpline(cfg).eval()
    with pytest.raises(hip.BflowHipError, match="MI355X"):
        m(voxel_grid=torch.zeros(1, 9, 64, 64), iters=1, test_mode=True)
    with pytest.raises(hip.BflowHipError):
        hip.corr_build_f32(torch.zeros(1, 16, 4), torch.zeros(1, 1, 16, 4), torch.zeros(1, 1, 4, 4))
    with pytest.raises(hip.BflowHipError):
        BezierCurves(torch.zeros(1, 4, 2, 2)).create_upsampled(torch.zeros(1, 576, 2, 2))
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "bflow_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_unsupported_configurations_raise_instead_of_falling_back():
    """Inference has no library / CPU path: a configuration the HIP engine cannot run is refused with BflowHipError."""
    import copy
    import pytest as _pt
    from bflow_amd import configs, hip
    base = configs.model_config("E_LU4_BD2")
    cases = []
    c = copy.deepcopy(base); c["feature"]["norm"] = "layer"; cases.append((c, None))     # not a norm_fn of the reference either (extractor.py:13-37)
    c = copy.deepcopy(base); c["feature"]["dim"] = 320; cases.append((c, "output dim 320"))   # (<= 256 is zero-padded to 64 / 128 / 256: round 5)
    c = copy.deepcopy(base); c["bezier_degree"] = 20; cases.append((c, "bezier_degree"))
    c = copy.deepcopy(base); c["motion"]["dim"] = 100; cases.append((c, "multiples of 32"))
    for cfg, needle in cases:
        if needle is None:
            with _pt.raises(NotImplementedError):
                bflow_amd.RAFTSpline(cfg)
            continue
        m = bflow_amd.RAFTSpline(cfg)
        with _pt.raises(hip.BflowHipError) as ei:
            m.check_engine_support()
        assert needle in str(ei.value), (needle, str(ei.value))
    for fn, cn in (("group", "none"), ("none", "group")):       # round 5: the whole norm_fn surface of the reference passes
        c = copy.deepcopy(base); c["feature"]["norm"], c["context"]["norm"] = fn, cn
        bflow_amd.RAFTSpline(c).check_engine_support()
    for fd in (32, 96, 192):                                      # and feature dims below 256
        c = copy.deepcopy(base); c["feature"]["dim"] = fd
        bflow_amd.RAFTSpline(c).check_engine_support()
    bflow_amd.RAFTSpline(base).check_engine_support()     # every shipped configuration passes
    for name in configs.EXPERIMENTS:
        bflow_amd.RAFTSpline(configs.model_config(name)).check_engine_support()


@pytest.mark.parametrize("name", list(configs.EXPERIMENTS))
def test_config_tree_and_state_dict_compat(name):
    cfg = configs.model_config(name)
    ref = O.model_config(name)
    for k in ("num_bins", "bezier_degree", "detach_bezier", "use_boundary_images", "use_events", "hidden", "context", "feature",
              "motion", "num_iter"):
        assert cfg[k] == ref[k], k
    assert cfg["correlation"]["ev"] == ref["correlation"]["ev"] and cfg["correlation"]["use_cosine_sim"] is False
    model = bflow_amd.RAFTSpline(cfg)
    shapes = O.param_shapes(cfg)   # checked against the reference's own state dict in test_oracle_vs_reference.py
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    a, b = deterministic_state_dict(model, 3), O.make_state_dict(cfg, 3)
    assert all(torch.equal(a[k], b[k]) for k in b)
    model.load_state_dict(a)
    assert model.lookup_timestamps == O.lookup_times(cfg)


def test_constructor_asserts_match_reference():
    cfg = configs.model_config("E_LU4_BD2")
    bad = {**cfg, "correlation": {**cfg["correlation"], "ev": {**cfg["correlation"]["ev"], "target_indices": [0, 1, 2, 3]}}}
    with pytest.raises(AssertionError):
        bflow_amd.RAFTSpline(bad)            # raft.py:64
    bad = {**cfg, "correlation": {**cfg["correlation"], "ev": {**cfg["correlation"]["ev"], "target_indices": [1, 2, 3, 5]}}}
    with pytest.raises(AssertionError):
        bflow_amd.RAFTSpline(bad)            # raft.py:66
    bad = {**cfg, "bezier_degree": 0}
    with pytest.raises(AssertionError):
        bflow_amd.RAFTSpline(bad)            # raft.py:24


def test_corr_computation_bookkeeping():
    f1 = torch.zeros(2, 8, 4, 6)
    f2 = torch.zeros(3, 2, 8, 4, 6)
    cc = CorrComputation(f1, f2, [1, 2, 3])
    assert (cc.batch, cc.dim, cc.height, cc.width) == (2, 8, 4, 6)
    assert cc.num_targets_overall == 3 and cc.num_levels_per_target_merged.tolist() == [1, 2, 3]
    both = cc + CorrComputation(f1, f2[0], 4)
    assert both.num_references == 2 and both.num_targets_per_reference == [3, 1] and both.levels_flat() == [1, 2, 3, 4]
    with pytest.raises(AssertionError):
        CorrComputation(f1, f2, [1, 2])      # corr.py:159
    with pytest.raises(AssertionError):
        CorrComputation(f1, torch.zeros(3, 2, 8, 4, 7), [1, 2, 3])   # corr.py:156


def test_bezier_container_on_host():
    p = torch.from_numpy(np.random.RandomState(0).standard_normal((2, 20, 3, 4)).astype(np.float32))
    c = BezierCurves(p)
    assert (c.degree, c.batch_size, c.dim, c.height, c.width) == (10, 2, 20, 3, 4)
    assert torch.equal(c.get_flow_from_reference(1.0), O.bezier_flow(p, 1.0))
    assert torch.equal(c.get_flow_from_reference(0), O.bezier_flow(p, 0))
    np.testing.assert_allclose(c.get_flow_from_reference([0.2, 0.7]).numpy(), O.bezier_flow(p, [0.2, 0.7]).numpy(), rtol=1e-6, atol=1e-6)
    c.delta_update_params(torch.ones_like(p))
    assert torch.equal(c.get_params(), p + 1)
    z = BezierCurves.create_from_voxel_grid(torch.zeros(2, 9, 64, 96), bezier_degree=2)
    assert z.get_params().shape == (2, 4, 8, 12) and float(z.get_params().abs().sum()) == 0
    with pytest.raises(AssertionError):
        BezierCurves.create_from_voxel_grid(torch.zeros(1, 9, 60, 96))   # bezier.py:67-68


def test_bench_self_launches_its_ranks(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment must become the launcher of N ranks (the driver's scaling tier may call
    it either way), on 127.0.0.1 with a free port, passing its own arguments through."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.relaunch(bench.parse_args()) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(seen["env"]["OMP_NUM_THREADS"]) >= 1            # the per-rank host-thread policy is stated by the launcher, not left to torchrun
    # the global batch of configs[3] splits into micro-batches of 8 for every N of the scaling curve
    for n in (1, 2, 4, 8):
        a, b = bdist.shard_range(bench.GLOBAL_BATCH, 0, n)
        assert (b - a) % min(bench.MICRO_BATCH, b - a) == 0 and (b - a) * n == bench.GLOBAL_BATCH


def test_micro_batch_plan_covers_the_global_batch_exactly_once():
    """bench.py's C4 leg: for every N of the scaling curve the ranks' micro-batches tile the 64 frames exactly once; at N = 8 a rank runs
    its 8 frames as two micro-batches of 4 (both in flight in one graph)."""
    import bench
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            micro, plan = bench.micro_batch_plan(bench.GLOBAL_BATCH, r, world)
            assert len(plan) >= 2 and all(n == micro <= bench.MICRO_BATCH for _, n in plan)
            seen += [f for first, n in plan for f in range(first, first + n)]
        assert sorted(seen) == list(range(bench.GLOBAL_BATCH))
    assert bench.micro_batch_plan(bench.GLOBAL_BATCH, 3, 8) == (4, [(24, 4), (28, 4)])
    assert bench.micro_batch_plan(bench.GLOBAL_BATCH, 0, 1)[0] == 8 and len(bench.micro_batch_plan(bench.GLOBAL_BATCH, 0, 1)[1]) == 8
    assert bench.micro_batch_plan(bench.GLOBAL_BATCH, 1, 8, graph=False) == (8, [(8, 8)])


def test_shard_ranges_cover_the_global_batch():
    for G in (1, 7, 8, 64):
        for Wd in (1, 2, 4, 8):
            spans = [bdist.shard_range(G, r, Wd) for r in range(Wd)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import torch
torch.set_num_threads(2)
from bflow_amd import dist as bdist, synthetic
from oracle import raft_spline_oracle as O
rank, world, local = bdist.init_from_env("gloo")
cfg = O.model_config("E_LU4_BD2")
sd = O.make_state_dict(cfg, 0)
G, H, W = 4, 128, 160
def fwd(first, n):
    vox = torch.from_numpy(synthetic.voxel_grid(n, 9, H, W, seed=1234, first_sample=first))
    with torch.inference_mode():
        low, up = O.forward(sd, cfg, vox, None, iters=2, test_mode=True)
    return O.bezier_flow(up, 1.0)
def gt(first, n):
    return torch.from_numpy(synthetic.gt_flow(n, H, W, seed=99, first_sample=first))
mean, s, c = bdist.evaluate_sharded(fwd, gt, O.epe_masked, G, 1, rank, world)
# the validation harness's (M, 2) metric-state table (SURVEY 8(f-3)): every rank holds the rows of ITS samples, one all-gather sums them
start, stop = bdist.shard_range(G, rank, world)
table = torch.zeros(9, 2, dtype=torch.float64)
for i in range(start, stop):
    p, g = fwd(i, 1), gt(i, 1)
    table[0] += torch.stack([O.epe_masked(p, g).double(), torch.tensor(1.0, dtype=torch.float64)])
    table[1] += torch.stack([O.ae_masked(p, g).double(), torch.tensor(1.0, dtype=torch.float64)])
    table[3] += torch.stack([O.n_pixel_error_masked(p, g, None, 2).double(), torch.tensor(1.0, dtype=torch.float64)])
table = bdist.reduce_metric_states(table)
if rank == 0:
    print("RESULT " + json.dumps(dict(mean=float(mean), sum=float(s), count=float(c), world=world, table=table.tolist())))
import torch.distributed as d
if d.is_initialized():
    d.barrier(); d.destroy_process_group()
"""


def _run_world(world: int, port: int):
    code = _WORKER.format(root=ROOT)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")][0]
    import json
    return json.loads(line[len("RESULT "):])


def test_two_rank_gloo_shard_equivalence():
    """N-rank EPE == 1-rank EPE on the same global batch: contiguous shards, per-rank metric state,
    one all-gather of the (epe_sum, count) record.  gloo on CPU with the oracle as the forward function."""
    one = _run_world(1, 29631)
    two = _run_world(2, 29632)
    assert one["count"] == two["count"] == 4
    assert abs(one["sum"] - two["sum"]) < 1e-9 and abs(one["mean"] - two["mean"]) < 1e-9
    t1, t2 = np.array(one["table"]), np.array(two["table"])
    assert t1.shape == (9, 2) and t1[0, 1] == 4 and np.allclose(t1, t2, rtol=0, atol=1e-9)
    assert abs(t1[0, 0] - one["sum"]) < 1e-9                      # row 0 is the EPE state


_DDP_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch
torch.set_num_threads(2)
from bflow_amd import dist as bdist
rank, world, local = bdist.init_from_env("gloo")
torch.manual_seed(100 + rank)                       # different initial weights per rank: the broadcast must make them equal
net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                          torch.nn.Conv2d(8, 2, 1)).double()
unused = torch.nn.Parameter(torch.ones(5, dtype=torch.float64))   # a parameter that never gets a gradient
net.register_parameter("unused", unused)
bdist.broadcast_module_state(net, src=0)
buckets = bdist.GradientBuckets(net, bucket_bytes=2048)           # small buckets: several all-reduces in flight during backward
G = 4
per = bdist.per_gpu_batch_size(G, world)
start, stop = bdist.shard_range(G, rank, world)
gen = torch.Generator().manual_seed(7)
x = torch.randn(G, 3, 12, 10, generator=gen, dtype=torch.float64); y = torch.randn(G, 2, 12, 10, generator=gen, dtype=torch.float64)
for step in range(2):                                # second step: the bucket bookkeeping resets
    net.zero_grad(set_to_none=True)
    loss = (net(x[start:stop]) - y[start:stop]).abs().sum(1).mean()
    loss.backward()
    buckets.finish()
flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
if rank == 0:
    print("RESULT " + json.dumps(dict(grad=flat.tolist(), weights=w.tolist(), buckets=buckets.num_buckets, per=per)))
import torch.distributed as d
if d.is_initialized():
    d.barrier(); d.destroy_process_group()
"""


def test_two_rank_gloo_gradient_buckets():
    """SURVEY 8(f-4), DDP: the rank-averaged gradient of 2 ranks x 2 samples == the gradient of 1 rank x 4 samples (the loss is a
    mean over samples), with the bucketed asynchronous all-reduce launched from the backward hooks; weights follow rank 0."""
    def run(world, port):
        code = _DDP_WORKER.format(root=ROOT)
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=300)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        import json
        return json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    one, two = run(1, 29641), run(2, 29642)
    assert one["per"] == 4 and two["per"] == 2 and two["buckets"] >= 3
    assert np.allclose(one["weights"], two["weights"], rtol=0, atol=0)
    assert np.allclose(one["grad"], two["grad"], rtol=0, atol=1e-12)
    with pytest.raises(AssertionError, match="divisible"):
        bdist.per_gpu_batch_size(6, 4)


def test_dropin_modules_resolve_to_this_package():
    """INTEGRATION.md seam 1: with <repo>/dropin first on PYTHONPATH the reference's import lines (modules/raft_spline.py:9,13,
    callbacks/logger.py:20, models/raft_spline/raft.py's own imports) resolve to bflow_amd.  Separate process: other tests import
    the real reference under the same module names."""
    code = r"""
import sys
sys.path[:0] = [{dropin!r}, {root!r}]
from models.raft_spline.raft import RAFTSpline, BezierCurves
from models.raft_spline.bezier import BezierCurves as B2
from models.raft_utils.corr import CorrComputation, CorrBlockParallelMultiTarget
from models.raft_utils.utils import cvx_upsample, coords_grid
from utils.losses import l1_seq_loss_channel_masked, l1_multi_seq_loss_channel_masked, l1_loss_channel_masked
import bflow_amd, bflow_amd.corr, bflow_amd.training
assert RAFTSpline is bflow_amd.RAFTSpline and BezierCurves is bflow_amd.bezier.BezierCurves is B2
assert CorrComputation is bflow_amd.corr.CorrComputation and CorrBlockParallelMultiTarget is bflow_amd.corr.CorrBlockParallelMultiTarget
assert l1_seq_loss_channel_masked is bflow_amd.training.l1_seq_loss_channel_masked
print("OK")
""".format(dropin=os.path.join(ROOT, "dropin"), root=ROOT)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout


def test_root_configs_compose_like_the_reference_entry_points():
    """train.yaml -> general.yaml + groups (train.py:26), val.yaml (val.py:22): same top-level sections, the experiment file's overrides win,
    and the one interpolation of the tree (`training.lr_scheduler.total_steps: ${..max_steps}`) resolves against the MERGED config."""
    c = configs.compose(root="train", dataset="dsec", experiment="dsec/raft_spline/E_LU4_BD2_lowpyramid")
    assert sorted(c) == ["dataset", "debugging", "hardware", "logging", "model", "training", "wandb"]
    assert c["training"]["max_steps"] == 250000 and c["training"]["lr_scheduler"]["total_steps"] == 250000     # experiment override + interpolation
    assert c["training"]["multi_loss"] is False and c["training"]["learning_rate"] == 1e-4 and c["wandb"]["project_name"] == "contflow"
    g = configs.compose(root="train", dataset="dsec", model="raft-spline")   # no experiment file: general.yaml's own values
    assert g["training"]["max_steps"] == 200000 and g["training"]["lr_scheduler"]["total_steps"] == 200000 and g["training"]["multi_loss"] is True
    v = configs.compose(root="val", dataset="dsec", experiment="dsec/raft_spline/E_LU4_BD2_lowpyramid")
    assert v["batch_size"] == 8 and v["hardware"] == {"num_workers": 4, "gpus": 0} and v["checkpoint"] == "???"


def test_every_shipped_config_file_parses_like_the_reference_file():
    """All 13 YAML files of the Hydra tree (root, dataset, model and experiment groups) carry the reference's keys and values: the same
    relative paths exist on both sides and `yaml.safe_load` of each pair is equal.  Needs the mounted reference (build container only)."""
    import yaml
    ref_root = "/root/reference/config"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not mounted")

    def tree(root):
        return sorted(os.path.relpath(os.path.join(d, f), root) for d, _, fs in os.walk(root) for f in fs if f.endswith(".yaml"))
    ours, theirs = tree(configs.CONFIG_ROOT), tree(ref_root)
    assert ours == theirs and len(ours) == 13, (ours, theirs)
    for rel in ours:
        assert yaml.safe_load(open(os.path.join(ref_root, rel))) == yaml.safe_load(open(os.path.join(configs.CONFIG_ROOT, rel))), rel


def test_hipgraph_mode_selection_at_the_seam(monkeypatch):
    """Which forwards replay a captured graph (host logic only, no launch): by default the call val.py makes -- eval(), grad disabled,
    test_mode=True (modules/raft_spline.py:57-58 under val.py:75) -- and nothing else; BFLOW_HIPGRAPH=0 / enable_hipgraph(False) opt out,
    enable_hipgraph() opts every inference forward in, stage timing is always eager; the weight generation moves with load_state_dict /
    .to() / train()."""
    m = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).eval()
    monkeypatch.delenv("BFLOW_HIPGRAPH", raising=False)
    assert m._graph_mode == "auto" and m.graph_replays() == 0
    assert not m._use_graph(True)                       # grad enabled: eager
    with torch.inference_mode():
        assert m._use_graph(True) and not m._use_graph(False)
        m.enable_stage_timing()
        assert not m._use_graph(True)
        m.enable_stage_timing(False)
        monkeypatch.setenv("BFLOW_HIPGRAPH", "0")
        assert not m._use_graph(True)
        monkeypatch.delenv("BFLOW_HIPGRAPH")
        m.enable_hipgraph(False)
        assert not m._use_graph(True) and m._graphs is None
        m.enable_hipgraph(None)
        assert m._use_graph(True)
    with torch.no_grad():
        assert m._use_graph(True)
    m.enable_hipgraph()
    assert m._use_graph(False) and m._use_graph(True)   # explicit mode: every inference forward
    g0 = m._weights_gen
    m.load_state_dict(deterministic_state_dict(m, seed=1))
    g1 = m._weights_gen
    m.float()
    g2 = m._weights_gen
    m.train()
    g3 = m._weights_gen
    m.eval()
    assert g0 < g1 < g2 < g3 == m._weights_gen
    from bflow_amd.graph import WeightsWatch
    w = WeightsWatch(m)
    assert w.changed() and not w.changed()
    with torch.no_grad():
        next(m.parameters()).mul_(1.0)                  # an in-place edit no hook sees: the version-counter sum does
    assert w.changed() and not w.changed()
    m.load_state_dict(m.state_dict())
    assert w.changed() and not w.changed()


def test_baseline_configs_and_the_precision_resolver(monkeypatch):
    """BASELINE.json configs in order; configs[4] ("fp16 MFMA correlation") selects `correlation.precision = "f16/w"` and the model takes it
    from its config; everything else resolves through ONE function (corr.default_precision): split8 on tiled planes with D in {128, 256},
    split otherwise, BFLOW_CORR_PRECISION read at call time and validated up front."""
    from bflow_amd import corr
    monkeypatch.delenv("BFLOW_CORR_PRECISION", raising=False)
    names = [(c["name"], c["experiment"], c["height"], c["width"], c["batch"], c["iters"]) for c in configs.BASELINE_CONFIGS]
    assert names == [("C1", "E_LU5_BD10", 384, 384, 1, 4), ("C2", "E_LU4_BD2", 480, 640, 1, 12), ("C3", "E_I_LU4_BD2", 480, 640, 8, 12),
                     ("C4", "E_LU4_BD2", 480, 640, 8, 12), ("C5", "E_I_LU5_BD10", 1024, 1024, 1, 20)]
    c5 = configs.baseline_config(4)
    assert c5["model"]["correlation"]["precision"] == "f16/w"
    assert bflow_amd.RAFTSpline(c5["model"]).resolved_corr_precision() == "f16/w"
    m2 = bflow_amd.RAFTSpline(configs.baseline_config(1)["model"])
    assert "precision" not in configs.baseline_config(1)["model"]["correlation"] and m2.corr_precision is None
    assert m2.resolved_corr_precision() == "split8" == corr.default_precision(256, tiled=True)
    assert corr.default_precision(256, tiled=False) == "split" == corr.default_precision(64, tiled=True)
    monkeypatch.setenv("BFLOW_CORR_PRECISION", "split")
    assert m2.resolved_corr_precision() == "split" == corr.default_precision(256, True)
    monkeypatch.setenv("BFLOW_CORR_PRECISION", "fp13")
    with pytest.raises(ValueError):
        m2.resolved_corr_precision()
    with pytest.raises(ValueError):
        corr.default_precision(256)
    monkeypatch.delenv("BFLOW_CORR_PRECISION")
    m2.corr_precision = "nonsense"
    with pytest.raises(ValueError):
        m2.resolved_corr_precision()


def test_lightning_checkpoint_loads_strictly(tmp_path):
    """weights.load_lightning_checkpoint: a file shaped like the reference's published checkpoints (`RAFTSplineModule.state_dict()`:
    every model tensor under the `net.` prefix, modules/raft_spline.py:24, next to Lightning's bookkeeping keys; loaded by
    `load_from_checkpoint`, val.py:58) loads STRICTLY into RAFTSpline -- and a bare state dict (no wrapper, no prefix) too.
    (The published files themselves cannot be fetched here: README.md:63-95, no network.)"""
    from bflow_amd.weights import load_lightning_checkpoint
    cfg = configs.model_config("E_I_LU4_BD2")
    src = bflow_amd.RAFTSpline(cfg)
    sd = deterministic_state_dict(src, seed=3)
    ckpt = {"epoch": 7, "global_step": 1234, "pytorch-lightning_version": "1.8.6", "state_dict": {"net." + k: v.clone() for k, v in sd.items()},
            "optimizer_states": [], "lr_schedulers": [], "hyper_parameters": {"config": {"model": cfg}}}
    path = str(tmp_path / "E_I_LU4_BD2.ckpt")
    torch.save(ckpt, path)
    dst = bflow_amd.RAFTSpline(cfg)
    res = load_lightning_checkpoint(dst, path, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    got = dst.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    bare = str(tmp_path / "bare.pt")
    torch.save(sd, bare)
    dst2 = bflow_amd.RAFTSpline(cfg)
    load_lightning_checkpoint(dst2, bare, strict=True)
    assert all(torch.equal(dst2.state_dict()[k], sd[k]) for k in sd)
    # a checkpoint of ANOTHER configuration is refused under strict=True (events-only model: no fnet_img, 567 correlation channels)
    other = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2"))
    with pytest.raises(RuntimeError):
        load_lightning_checkpoint(other, path, strict=True)


def test_split_tensor_pad_rows_and_the_lookup_callable():
    """Host logic of round 4's launch trims: `SplitTensor.empty(zero_tail=True)` zeroes exactly the pad rows [H*W, rows) (what K5's 128-row operand
    padding reads; the convolution writes every pixel row), and `SplitLookup` hands the im2col request through to the pyramid's look-up."""
    from bflow_amd import split as S
    from bflow_amd.update import SplitLookup
    t = S.SplitTensor.empty(2, 3, 5, 40, "cpu", rows=24, zero_tail=True)
    assert t.planes.shape == (2, 2, 2, 24, 32) and float(t.planes[:, :, :, 15:].abs().max()) == 0.0
    t.planes[:, :, :, :15] = 1.0                       # (the pixel rows are left to the producer)
    assert S.SplitTensor.empty(1, 2, 2, 32, "cpu", zero=True).planes.abs().max() == 0
    assert S.SplitTensor.empty(1, 2, 2, 32, "cpu", zero_tail=True).planes.shape == (2, 1, 1, 4, 32)   # no pad rows: nothing to do

    class Block:
        im2col_rider = True
        def lookup_bezier_split(self, params, coef, out, im2col=None):
            self.seen = (params, coef, out, im2col)
            return out
    blk = Block()
    call = SplitLookup(blk, "bezier", "coef", "feat")
    assert call.im2col_rider and call() == "feat" and blk.seen == ("bezier", "coef", "feat", None)
    assert call(im2col=("col", 7, 7, 3)) == "feat" and blk.seen[3] == ("col", 7, 7, 3)
    class Rows:                                        # a row-major pyramid has no rider
        def lookup_bezier_split(self, *a, **k): return None
    assert not SplitLookup(Rows(), 0, 0, 0).im2col_rider


def test_window_descriptors_of_the_twostep_assembly_on_the_host():
    """TwoStepAssembler.window_descriptor (what bflow_voxel_grid_rectified_window reads from device memory): the extended window of
    base.py:165-200 as {first event, count, centres}, equal to the slice EventStream.window hands the plain K1 call; consecutive frames of a
    100-ms-step sequence share a window (twostep.py:63-64) -- the condition EventFrameGraph / assemble() reuse a grid on.  Host logic only."""
    import numpy as np
    from bflow_amd.dsec import EventStream, TwoStepAssembler, event_window_indices, twostep_windows
    rs = np.random.RandomState(3)
    n = 20000
    t = np.sort(rs.randint(1_000_000, 1_400_000, n)).astype(np.int64)
    ev = EventStream.__new__(EventStream)              # host fields only (no device in the CPU tier)
    ev.t_host = t
    asm = TwoStepAssembler.__new__(TwoStepAssembler)
    from bflow_amd.representations import VoxelGrid
    asm.voxel_grid, asm.version = VoxelGrid(5, 48, 64), 1
    ts = np.array([[1_030_000 + 100_000 * k, 1_130_000 + 100_000 * k] for k in range(3)], dtype=np.int64)
    for k in range(3):
        i0, cnt, t0c, t1c = asm.window_descriptor(ev, int(ts[k][0]), int(ts[k][1]))
        ta, tb = asm.voxel_grid.get_extended_time_window(int(ts[k][0]), int(ts[k][1]))
        ta, tb = max(ta, int(t[0])), min(tb, int(t[-1]))
        assert (i0, i0 + cnt) == event_window_indices(t, ta, tb) and (t0c, t1c) == (int(ts[k][0]), int(ts[k][1]))
        assert cnt > 0 and np.all(t[i0:i0 + cnt] >= ta) and np.all(t[i0:i0 + cnt] < tb)
        assert (i0 == 0 or t[i0 - 1] < ta) and (i0 + cnt == n or t[i0 + cnt] >= tb)
    assert twostep_windows(ts, 2)[1] == twostep_windows(ts, 1)[0]          # frame 2's previous window IS frame 1's current one
    asm.version = 0
    with pytest.raises(AssertionError, match="extended_voxel_grid"):
        asm.window_descriptor(ev, int(ts[0][0]), int(ts[0][1]))

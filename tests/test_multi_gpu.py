"""N-rank == 1-rank on the HIP path over RCCL (reference metric sync utils/metrics.py:34-35,42-49).
Needs >= 2 GPUs in one node: skipped on the single-GPU box the driver's `-m gpu` tier runs on; bench.py --gpus N exercises the same
launch + collective path when a multi-GPU node is available."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world: int):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_hip.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one node")
def test_two_rank_rccl_epe_equals_one_rank():
    one, two = _run(1), _run(2)
    assert one["world"] == 1 and two["world"] == 2 and one["count"] == two["count"] == 4
    # per-sample EPE values are the same kernels on the same inputs on either GPU; the fp64 sum is order-independent to round-off
    assert abs(one["sum"] - two["sum"]) <= 1e-9 * abs(one["sum"]) and abs(one["mean"] - two["mean"]) <= 1e-9 * abs(one["mean"])


def test_one_rank_worker_runs_on_this_box():
    """The worker itself (model build, hipGraph forward, sharded evaluation, reduce with world = 1) on whatever GPU is here."""
    r = _run(1)
    assert r["world"] == 1 and r["count"] == 4 and r["mean"] > 0


def test_bench_n2_path_on_one_gpu_with_gloo():
    """The N > 1 branch of bench.py executed for real BEFORE an 8-GPU node does it: `bench.py --gpus 2 --steps 2 --warmup 1 --no-extras`
    self-launches two ranks through torch.distributed.run; BFLOW_DIST_BACKEND=gloo + BFLOW_DEVICE=0 put both ranks on the one GPU of this
    box (RCCL itself stays with test_two_rank_rccl_epe_equals_one_rank).  Exercises: rendezvous, contiguous sharding of the global batch
    (32 frames per rank in micro-batches of 8), ConcurrentRunner (two micro-batches in flight) under two processes, the barrier +
    all-reduce MAX of `time_steps`, and the single exchange step (all-gather of the per-rank EPE record).  `value` must be the SAME
    workload as at N = 1 (C2 weak: one frame per rank and step); no scaling number is asserted -- two ranks share one GPU."""
    env = dict(os.environ, BFLOW_DIST_BACKEND="gloo", BFLOW_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                       # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["epe_ranks_gathered"] == 2
    assert d["scaling"] == "weak" and d["config"]["frames_per_rank_per_step"] == 1 and d["config"]["global_batch"] == 2
    assert d["value"] == d["c2_weak"]["value"] and d["value"] > 0 and d["c2_weak"]["steps"] == 2
    c4 = d["c4_strong"]
    assert c4["steps"] == 2 and c4["frames_per_rank_per_step"] == 32 and c4["micro_batch"] == 8 and c4["micro_batches_per_rank"] == 4
    assert c4["value"] > 0 and "two micro-batches in flight" in c4["workload"]
    # whole-job frames/s = frames of all ranks / max-over-ranks time
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
    assert abs(c4["value"] - 64 * 2 / (c4["ms_per_step"] * 2e-3)) < 0.02 * c4["value"]

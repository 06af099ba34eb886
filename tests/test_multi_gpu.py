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

#!/usr/bin/env python
"""Launches ONE of bench.py's roofline kernels a few times on its operands, for rocprofv3 --pmc passes (tools only).
Prints `PROBE {json}` (key, name, regex, algorithmic bytes / flops per launch) for tools/pmc_to_json.py."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict
from tools.roofline_kernels import build, build_big, build_c4_convs
ap = argparse.ArgumentParser(); ap.add_argument("--key", default="roofline"); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev)
if a.key in ("roofline_lookup_c4_shard", "roofline_corr_build_c5"):
    ks = build_big(m, cfg, dev)
elif a.key.endswith("_c4"):
    ks = build_c4_convs(m, dev)
else:
    vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 480, 640, seed=1234)).to(dev)
    low, _ = m(voxel_grid=vox, iters=12, test_mode=True)
    ks = build(m, vox, cfg, low.get_params())
k = [k for k in ks if k["key"] == a.key][0]
for _ in range(a.reps): k["launch"]()
torch.cuda.synchronize()
print("PROBE " + json.dumps({"key": k["key"], "name": k["name"], "regex": k["regex"], "bytes": k["bytes"], "flops": k.get("flops"), "reps": a.reps}))

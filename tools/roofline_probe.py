#!/usr/bin/env python
"""Launches ONE of bench.py's roofline kernels a few times on the C2 operands, for rocprofv3 --pmc passes (tools only)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict
from tools.roofline_kernels import build
ap = argparse.ArgumentParser(); ap.add_argument("--key", default="roofline"); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev)
vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 480, 640, seed=1234)).to(dev)
low, _ = m(voxel_grid=vox, iters=12, test_mode=True)
k = [k for k in build(m, vox, cfg, low.get_params()) if k["key"] == a.key][0]
for _ in range(a.reps): k["launch"]()
torch.cuda.synchronize()
print(k["name"], "launched", a.reps, "times")

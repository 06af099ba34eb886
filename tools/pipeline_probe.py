#!/usr/bin/env python
"""bench.py's `pipeline_from_events` and `voxel_kernels` legs alone (tools only)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bflow_amd
from bflow_amd import configs
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev)
print(json.dumps({"voxel_kernels": bench.voxel_kernels(dev), "pipeline_from_events": bench.pipeline_from_events(m, cfg, dev)}, indent=1))

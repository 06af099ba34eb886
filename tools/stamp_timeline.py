#!/usr/bin/env python
"""Timeline of one hipGraph replay of the forward WITHOUT the profiler: one-thread kernels write the 100 MHz wall clock at
stage boundaries (bflow_clock_stamp, bflow_amd/timers.py StampTimer).  Usage: python tools/stamp_timeline.py [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict
from bflow_amd.timers import StampTimer

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).eval()
model.load_state_dict(deterministic_state_dict(model, seed=0)); model.to(dev); model.enable_hipgraph()
st = StampTimer(dev)
model._probe = st
vox = torch.from_numpy(synthetic.voxel_grid(B, 9, 480, 640, seed=1234)).to(dev)
for _ in range(6): model(voxel_grid=vox, iters=12, test_mode=True)
for n, us in sorted(st.read_us().items(), key=lambda kv: kv[1]):
    if not n.startswith("iter") or n in ("iters.end", "iter0.begin", "iter11.begin", "iter0.lookup_end"):
        print(f"{us:9.1f} us  {n}")
print({k: round(v, 4) for k, v in st.stage_ms().items()})

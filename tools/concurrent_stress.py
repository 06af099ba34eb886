#!/usr/bin/env python
"""Stress of ConcurrentRunner captures / replays next to ordinary graph forwards (tools only): usage concurrent_stress.py <streams> <rounds>"""
import faulthandler, os, sys
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.pipeline import ConcurrentRunner
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
streams, rounds = min(int(sys.argv[1]), 2), int(sys.argv[2])
for r in range(rounds):
    m = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).eval()
    m.load_state_dict(deterministic_state_dict(m, r)); m.to(dev); m.enable_hipgraph()
    for hw in ((96, 128), (480, 640)):
        fr = [torch.from_numpy(synthetic.voxel_grid(1, 9, *hw, seed=10 * r + i)).to(dev) for i in range(streams)]
        ref = [m(voxel_grid=f, iters=4, test_mode=True)[0].get_params().clone() for f in fr]
        run = ConcurrentRunner(m, 4, streams)
        for _ in range(5):
            outs = run(fr)
        assert all(torch.equal(o[0].get_params(), q) for o, q in zip(outs, ref))
    print("round", r, "ok", flush=True)

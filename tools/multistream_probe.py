#!/usr/bin/env python
"""Serving-throughput probe (tools only, NOT the headline metric): independent batch-1 frames in flight on several streams, one
captured hipGraph per stream.  The batch-1 GRU loop is a latency chain that leaves most CUs idle; a second frame fills them.
Usage: python tools/multistream_probe.py [n_streams ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
cfg = configs.model_config("E_LU4_BD2")
vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 480, 640, seed=1234)).to(dev)
for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    models, streams = [], []
    for i in range(ns):
        m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, seed=0)); m.to(dev); m.enable_hipgraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): m(voxel_grid=vox, iters=12, test_mode=True)
        models.append(m); streams.append(s)
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k % ns]):
            models[k % ns](voxel_grid=vox, iters=12, test_mode=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{ns} frame(s) in flight: {K/dt:7.1f} frames/s ({dt/K*1e3:.2f} ms per frame of throughput)", flush=True)
    del models, streams
    torch.cuda.empty_cache()

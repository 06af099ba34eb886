#!/usr/bin/env python
"""Measures the per-node cost of a tiny kernel inside a replayed hipGraph (tools only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip
dev = torch.device("cuda:0")
x = torch.zeros(1, 4, 8, 8, device=dev); bias = torch.zeros(4, device=dev)
def tiny(): hip.bias_act_inplace(x, bias, 0)
for n in (1, 50, 200):
    tiny(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): tiny()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f"graph of {n} tiny kernels: {a.elapsed_time(b)*1e3:.1f} us total, {a.elapsed_time(b)*1e3/n:.2f} us/node")

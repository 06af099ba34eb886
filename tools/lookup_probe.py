#!/usr/bin/env python
"""Graph-timed look-up variants on the C2 shape (tools only)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip, split as S
from bflow_amd.corr import CorrComputation, CorrBlockParallelMultiTarget
dev = torch.device("cuda:0")
B, D, h, w = 1, 256, 60, 80
f1, f2 = torch.randn(B, D, h, w, device=dev), torch.randn(1, B, D, h, w, device=dev)
blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, [4]))
deg = 2
params = torch.randn(B, 2 * deg, h, w, device=dev) * 3
coef = hip.bezier_coeffs([1.0], deg)
o1, o2 = blk.new_output(), blk.new_output_split()
def timed(fn, name):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f"{name:40s} {a.elapsed_time(b)/20*1e3:7.1f} us")
timed(lambda: blk.lookup_bezier(params, coef, out=o1), "lookup fp32 NCHW")
timed(lambda: S.from_nchw(o1), "from_nchw")
timed(lambda: blk.lookup_bezier_split(params, coef, o2), "lookup split")

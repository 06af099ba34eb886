#!/usr/bin/env python
"""Look-up kernel (tiled planes, split output) at the BASELINE shapes: time per launch inside a graph and a hash of the output (tools only; the
hash lets two builds of the kernel be compared bit for bit)."""
import hashlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip
from bflow_amd.corr import CorrComputation, CorrBlockParallelMultiTarget
dev = torch.device("cuda:0")
def ev(fn, n=30):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, B, D, h, w, levels, deg, f16 in (("C2", 1, 256, 60, 80, [1, 1, 1, 4], 2, False), ("C4 shard", 8, 256, 60, 80, [1, 1, 1, 4], 2, False),
                                          ("C5 f16", 1, 256, 128, 128, [1, 1, 1, 1, 4], 10, True), ("odd", 2, 64, 15, 21, [1, 2, 3], 3, False)):
    torch.manual_seed(1)
    T = len(levels)
    f1, f2 = torch.randn(B, D, h, w, device=dev), torch.randn(T, B, D, h, w, device=dev)
    blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, levels), layout="tiled", precision="f16" if f16 else None)
    params = torch.randn(B, 2 * deg, h, w, device=dev) * 4
    params[:, :, 0, :3] *= 50
    coef = hip.bezier_coeffs([(i + 1) / T for i in range(T)], deg)
    out = blk.new_output_split()
    out.planes.fill_(7.0)
    blk.lookup_bezier_split(params, coef, out); torch.cuda.synchronize()
    hsh = hashlib.sha1(out.planes.cpu().numpy().tobytes()).hexdigest()[:12]
    t = ev(lambda: blk.lookup_bezier_split(params, coef, out))
    print(f"{name:9s} B={B} {h}x{w} planes={blk.num_planes} deg={deg}: {t:7.1f} us   sha1 {hsh}")
    del blk, f1, f2

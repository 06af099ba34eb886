#!/usr/bin/env python
"""K1 (tile-binned voxel grid) timing: graph-timed whole calls on DSEC-shaped synthetic events, float and integer x/y, 5 / 15 / 65 bins
(tools only; bench.py's `voxel_kernels` carries the judged numbers)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip, synthetic
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


CASES = ((15, 480, 640, 2_000_000), (5, 480, 640, 2_000_000), (5, 480, 640, 500_000), (65, 384, 384, 2_000_000), (65, 1024, 1024, 4_000_000))
if len(sys.argv) > 1:
    CASES = CASES[int(sys.argv[1]):int(sys.argv[1]) + 1]
for C, H, W, n_ev in CASES:
    for int_xy in ((False,) if len(sys.argv) > 2 else (False, True)):
        ev = synthetic.events(n_ev, H, W, 0, 100_000, seed=7, int_xy=int_xy)
        x, y, p, t = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in ev)
        grid = torch.empty((C, H, W), device=dev)
        ws = hip._voxel_workspace(n_ev, C, H, W, not int_xy, dev)
        us = timed(lambda: hip.voxel_grid(x, y, p, t, 0, 100_000, grid, ws))
        atom = 2 if int_xy else 8
        by = n_ev * (16 + atom * 8) + C * H * W * 4
        print(f"K1 {'int' if int_xy else 'float'}-xy C={C:2d} {H}x{W} {n_ev/1e6:.1f} M events: {us:8.1f} us  {n_ev/us/1e3:6.2f} G events/s  "
              f"{by/us/1e3:7.1f} GB/s algorithmic ({by/us/1e3/8000:.3f} of 8 TB/s)  workspace {ws.numel()/1e6:.0f} MB")

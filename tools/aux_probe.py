#!/usr/bin/env python
"""Graph-timed launches of the streaming kernels either side of the network: K1 voxel scatter, K2 voxel normalisation, K6 pooling,
K13 convex up-sampling, K15 EPE, on DSEC-sized operands (tools only)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
def timed(fn, name, nbytes, reps=20, note=""):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    print(f"{name:44s} {us:8.1f} us  {nbytes/us/1e3:8.1f} GB/s algorithmic  {note}")
H, W, C = 480, 640, 9
for n_ev in (500_000, 2_000_000):
    x = torch.from_numpy(rs.uniform(0, W - 1, n_ev).astype(np.float32)).to(dev)
    y = torch.from_numpy(rs.uniform(0, H - 1, n_ev).astype(np.float32)).to(dev)
    p = torch.from_numpy(rs.randint(0, 2, n_ev).astype(np.int8)).to(dev)
    t = torch.from_numpy(np.sort(rs.randint(0, 100_000, n_ev)).astype(np.int64)).to(dev)
    grid = torch.zeros((C, H, W), device=dev)
    timed(lambda: hip.voxel_grid(x, y, p, t, 0, 100_000, grid), f"K1 voxel_scatter f32xy, {n_ev/1e6:.1f} M events", n_ev * (17 + 8 * 8),
          note=f"{n_ev/1e6:.1f} M events -> {n_ev*8/1e9:.3f} G atomics")
    xi, yi = x.round().to(torch.int16), y.round().to(torch.int16)
    timed(lambda: hip.voxel_grid(xi, yi, p, t, 0, 100_000, grid), f"K1 voxel_scatter i16xy, {n_ev/1e6:.1f} M events", n_ev * (13 + 2 * 8))
grid = torch.from_numpy((rs.standard_normal((C, H, W)) * (rs.uniform(size=(C, H, W)) < 0.3)).astype(np.float32)).to(dev)
ws = hip.voxel_norm_workspace(dev)
timed(lambda: hip.voxel_norm(grid, ws), "K2 voxel_norm 9x480x640", 4.0 * grid.numel() * 4, note="(3 reads + 1 write)")
N = 4800
src = torch.randn((N, 60, 80), device=dev); dst = torch.empty((N, 30, 40), device=dev)
timed(lambda: hip.corr_pool2x2(src, dst), "K6 corr_pool2x2 level 0 (one target)", 4.0 * N * (4800 + 1200))
data = torch.randn((1, 4, 60, 80), device=dev); mask = torch.randn((1, 576, 60, 80), device=dev)
timed(lambda: hip.cvx_upsample(data, mask, None, 0.25), "K13 cvx_upsample deg 2", 4.0 * (576 * 4800 + 4 * 4800 + 4 * 480 * 640))
pred = torch.randn((1, 2, H, W), device=dev); gt = torch.randn((1, 2, H, W), device=dev); valid = torch.rand((1, H, W), device=dev) > 0.2
acc = torch.zeros(2, dtype=torch.float64, device=dev)
timed(lambda: hip.epe_accumulate(pred, gt, valid, acc), "K15 epe_accumulate 480x640 masked", 4.0 * 4 * H * W + H * W)

#!/usr/bin/env python
"""The per-rank shape of BASELINE configs[3] at N = 8 (8 frames per rank and step) as 2 x 4 / 1 x 8 / 2 x 8 frames in flight (tools only;
BFLOW_SMALL_GRID_MAX_PIXELS moves the batch-4 forwards between the small-grid and the large-grid launch plans of the update block)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.pipeline import ConcurrentRunner
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev); m.enable_hipgraph()

def timed(fn, frames, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return frames * k / (time.perf_counter() - t0)

vox = lambda n, k: torch.from_numpy(synthetic.voxel_grid(n, 9, 480, 640, seed=1234, first_sample=n * k)).to(dev)
with torch.inference_mode():
    v4 = [vox(4, 0), vox(4, 1)]
    pair = ConcurrentRunner(m, 12, streams=2)
    print(f"small-grid threshold {os.environ.get('BFLOW_SMALL_GRID_MAX_PIXELS', '20000')}: 2 x 4 in flight {timed(lambda: pair(v4), 8):.1f} frames/s", flush=True)
    pair.close()
    v8 = vox(8, 0)
    print(f"   1 x 8  {timed(lambda: m(voxel_grid=v8, iters=12, test_mode=True), 8):.1f} frames/s", flush=True)
    v4s = vox(4, 0)
    print(f"   1 x 4  {timed(lambda: m(voxel_grid=v4s, iters=12, test_mode=True), 4):.1f} frames/s", flush=True)
    for nb in (8, 16):
        vv = [vox(nb, 0), vox(nb, 1)]
        pr = ConcurrentRunner(m, 12, streams=2)
        print(f"   2 x {nb} in flight  {timed(lambda: pr(vv), 2 * nb, k=6):.1f} frames/s", flush=True)
        pr.close()

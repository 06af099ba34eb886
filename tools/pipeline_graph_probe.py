#!/usr/bin/env python
"""Where the per-frame time of bflow_amd.pipeline.EventFrameGraph goes (tools only): the forward alone, the serial graph with / without the
per-frame descriptor copy, eager assembly alone."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.dsec import EventStream, TwoStepAssembler
from bflow_amd.pipeline import EventFrameGraph
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev)
H, W, bins = 480, 640, cfg["num_bins"]["correlation"]
rs = np.random.RandomState(21)
n_frames = 16
span = 60_000 + 100_000 * n_frames
n = 20 * span
ev = dict(x=rs.randint(0, W, n, dtype=np.int32).astype(np.uint16), y=rs.randint(0, H, n, dtype=np.int32).astype(np.uint16),
          p=rs.randint(0, 2, n, dtype=np.int32).astype(np.uint8), t=np.sort(rs.randint(1_000_000, 1_000_000 + span, n)).astype(np.int64))
yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
rect = np.stack([xx * 1.01 - 3 + np.sin(yy / 40.0), yy * 0.99 + 2 + np.cos(xx / 50.0)], -1).astype(np.float32)
ts = np.array([[1_030_000 + 100_000 * k, 1_130_000 + 100_000 * k] for k in range(n_frames)], dtype=np.int64)
stream = EventStream(**ev, device=dev)
asm = TwoStepAssembler(bins, H, W, rect, device=dev)
vox = asm.assemble(stream, ts, 1, check=False)

def timed(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

def fwd():
    with torch.inference_mode(): return m(voxel_grid=vox[None], iters=12, test_mode=True)
def asm_only():
    with torch.inference_mode(): return asm.assemble(stream, ts, 1, check=False)
g = EventFrameGraph(m, asm, stream, 12, reuse_windows=False)
def graph_full():
    with torch.inference_mode(): return g(ts, 1)
gg = torch.cuda.CUDAGraph()
g._write_windows(ts, 1)
with torch.inference_mode(False), torch.no_grad():
    g._assemble(g.grid[0]); torch.cuda.synchronize()
    with torch.cuda.graph(gg):
        g._assemble(g.grid[0])
gi = torch.cuda.CUDAGraph()
with torch.inference_mode(False), torch.no_grad():
    with torch.cuda.graph(gi):
        g._assemble(g.grid[0], True, True)
for rep in range(2):
    print(f"assembly as a graph (2 x K1)  {timed(gg.replay):.4f} ms;  one K1 + merge + keep {timed(gi.replay):.4f} ms")
    print(f"forward replay alone          {timed(fwd):.4f} ms")
    print(f"eager assembly alone          {timed(asm_only):.4f} ms")
    print(f"serial graph, both windows    {timed(graph_full):.4f} ms")
    w = g._write_windows
    g._write_windows = lambda *a: None          # the descriptor stays what the last call wrote: what the per-frame host work + copy cost
    print(f"  ... without the descriptor   {timed(graph_full):.4f} ms")
    g._write_windows = w
    def both():
        with torch.inference_mode():
            v = asm.assemble(stream, ts, 1, check=False)
            return m(voxel_grid=v[None], iters=12, test_mode=True)
    print(f"eager assembly + replay       {timed(both):.4f} ms")

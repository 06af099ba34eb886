#!/usr/bin/env python
"""One C2 forward (480x640, B, 12 iters) in eager mode, for `rocprofv3 --kernel-trace --stats`: 3 warm-ups + `--reps` timed."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--no-branch", action="store_true", help="every side-stream branch on the main stream (hip.BRANCHING = False)")
args = ap.parse_args()
if args.no_branch:
    from bflow_amd import hip as _hip
    _hip.BRANCHING = False
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = configs.model_config("E_LU4_BD2")
m = bflow_amd.RAFTSpline(cfg).eval()
m.load_state_dict(deterministic_state_dict(m, 0))
m.to(dev)
if args.graph:
    m.enable_hipgraph()
vox = torch.from_numpy(synthetic.voxel_grid(args.batch, 9, 480, 640, seed=1234)).to(dev)
for _ in range(3):
    m(voxel_grid=vox, iters=args.iters, test_mode=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    m(voxel_grid=vox, iters=args.iters, test_mode=True)
torch.cuda.synchronize()
print(f"forward B={args.batch} iters={args.iters} graph={args.graph}: {(time.perf_counter() - t0) / args.reps * 1e3:.3f} ms")

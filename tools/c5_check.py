#!/usr/bin/env python
"""BASELINE config C5 at full size (E_I_LU5_BD10: events + images, 1024x1024, degree 10, 6 targets, 20 iterations; 6.4 GB volume)
on the GPU against the CPU oracle (minutes of CPU time: tools only, not part of the pytest suite).  Usage: tools/c5_check.py [--no-oracle]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bflow_amd
from bflow_amd import synthetic
from oracle import raft_spline_oracle as O
dev = "cuda"
from bflow_amd import configs
c5 = configs.baseline_config(4)   # BASELINE configs[4]: E_I_LU5_BD10, 1024 x 1024, 20 iterations, "fp16 MFMA correlation" = correlation.precision "f16/w"
cname, B, H, W, iters = c5["experiment"], c5["batch"], c5["height"], c5["width"], c5["iters"]
cfg = O.model_config(cname); sd = O.make_state_dict(cfg, 0)
m = bflow_amd.RAFTSpline(c5["model"]).eval(); m.load_state_dict(sd); m.to(dev); m.enable_hipgraph()
if "--f16" in sys.argv:
    m.corr_precision = "f16"      # opt-in: fp16 operands AND fp16 volume (half the volume bytes, 2.3e-3 px)
elif "--split8" in sys.argv:
    m.corr_precision = "split8"
C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
vox = torch.from_numpy(synthetic.voxel_grid(B, C, H, W, seed=7))
a, b = synthetic.image_pair(B, H, W, seed=8); imgs = [torch.from_numpy(a), torch.from_numpy(b)]
gv, gi = vox.to(dev), [i.to(dev) for i in imgs]
for _ in range(2):
    lo, up = m(voxel_grid=gv, images=gi, iters=iters, test_mode=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    lo, up = m(voxel_grid=gv, images=gi, iters=iters, test_mode=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
f = up.get_flow_from_reference(1.0).cpu()
print(f"C5 GPU (corr_precision = {m.resolved_corr_precision()}): {dt*1e3:.1f} ms/frame ({1/dt:.1f} frames/s), |flow| mean {float(f.abs().mean()):.3f}, finite {bool(torch.isfinite(f).all())}, "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
if "--no-oracle" not in sys.argv:
    t0 = time.perf_counter()
    with torch.inference_mode():
        olo, oup = O.forward(sd, cfg, vox, imgs, iters=iters, test_mode=True)
    of = O.bezier_flow(oup, 1.0)
    epe = float(torch.sqrt(((f - of) ** 2).sum(1)).mean())
    print(f"C5 oracle: {time.perf_counter()-t0:.0f} s on CPU; EPE GPU vs oracle {epe:.3e} px (|flow| {float(torch.sqrt((of**2).sum(1)).mean()):.2f})")
    assert epe < (1e-2 if "--f16" in sys.argv else 1e-3)

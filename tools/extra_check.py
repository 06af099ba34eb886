#!/usr/bin/env python
"""Full-size parity runs of the other shipped configurations (C3-shaped batch 2 with images, odd batch, the degree-10 family)
against the CPU oracle -- slower than the pytest suite allows (tools only)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bflow_amd
from bflow_amd import synthetic
from oracle import raft_spline_oracle as O
dev = "cuda"
def run(cname, B, H, W, iters, graph):
    cfg = O.model_config(cname); sd = O.make_state_dict(cfg, 0)
    m = bflow_amd.RAFTSpline(cfg).eval(); m.load_state_dict(sd); m.to(dev)
    if graph: m.enable_hipgraph()
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    vox = synthetic.voxel_grid(B, C, H, W, seed=7)
    imgs = None
    if cfg["use_boundary_images"]:
        a, b = synthetic.image_pair(B, H, W, seed=8); imgs = [torch.from_numpy(a), torch.from_numpy(b)]
    with torch.no_grad():
        for _ in range(2):
            lo, up = m(voxel_grid=torch.from_numpy(vox).to(dev), images=None if imgs is None else [i.to(dev) for i in imgs], iters=iters, test_mode=True)
        torch.cuda.synchronize()
        olo, oup = O.forward(sd, cfg, torch.from_numpy(vox), imgs, iters=iters, test_mode=True)
    f, of = up.get_flow_from_reference(1.0).cpu(), O.bezier_flow(oup, 1.0)
    epe = float(torch.sqrt(((f - of) ** 2).sum(1)).mean()); mag = float(torch.sqrt((of ** 2).sum(1)).mean())
    print(f"{cname:14s} B={B} {H}x{W} iters={iters} graph={graph}: EPE vs oracle {epe:.3e} px (|flow| {mag:.2f})", flush=True)
    assert epe < 1e-3
run("E_I_LU4_BD2", 2, 480, 640, 12, True)      # C3-shaped
run("E_LU4_BD2", 3, 480, 640, 12, True)        # odd batch
run("E_I_LU5_BD10", 1, 384, 512, 6, False)     # C5-family at a medium size, eager
run("E_LU5_BD10", 2, 384, 384, 4, True)        # C1 at its own size

#!/usr/bin/env python
"""End-to-end effect of the correlation arithmetic / storage variants (tools only): EPE of the full forward against the fp32 CPU oracle
and the frame time, per `RAFTSpline.corr_precision`.  Decomposes the fp16 variant's error into operand and storage rounding.
    python tools/corr_precision_probe.py [--c5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bflow_amd  # noqa: E402
from bflow_amd import configs, synthetic  # noqa: E402
from bflow_amd.weights import deterministic_state_dict  # noqa: E402
from oracle import raft_spline_oracle as O  # noqa: E402  (checker only)


def run(cname, H, W, iters, precisions, seeds):
    dev = torch.device("cuda:0")
    cfg = configs.model_config(cname)
    model = bflow_amd.RAFTSpline(cfg).eval()
    sd = deterministic_state_dict(model, seed=0)
    model.load_state_dict(sd)
    model.to(dev)
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    for seed in seeds:
        vox = torch.from_numpy(synthetic.voxel_grid(1, C, H, W, seed=seed))
        imgs = None
        if cfg["use_boundary_images"]:
            a, b = synthetic.image_pair(1, H, W, seed=seed + 1)
            imgs = [torch.from_numpy(a), torch.from_numpy(b)]
        with torch.inference_mode():
            _, rup = O.forward(sd, cfg, vox, imgs, iters=iters, test_mode=True)
        rflow = O.bezier_flow(rup, 1.0)
        for prec in precisions:
            model.corr_precision = prec
            model.enable_hipgraph()
            kw = dict(voxel_grid=vox.to(dev), images=None if imgs is None else [i.to(dev) for i in imgs], iters=iters, test_mode=True)
            low, up = model(**kw)
            flow = up.get_flow_from_reference(1.0).cpu()
            e = float(O.epe_masked(flow, rflow))
            for _ in range(3):
                model(**kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model(**kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"{cname} {H}x{W} {iters} it seed {seed} corr_precision={prec:9s}: EPE vs fp32 oracle {e:.3e} px (|flow| mean {float(rflow.abs().mean()):.1f} px), "
                  f"{ms:.3f} ms/frame", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5", action="store_true")
    ap.add_argument("--precisions", default="split,split8,f16/w,split/h,split8/h,f16")
    args = ap.parse_args()
    precs = args.precisions.split(",")
    run("E_LU4_BD2", 480, 640, 12, precs, seeds=(7, 21))
    if args.c5:
        run("E_I_LU5_BD10", 1024, 1024, 20, precs, seeds=(7,))

#!/usr/bin/env python
"""K7 probe (tools only): duration and algorithmic GB/s of the fused Bezier look-up (split output, the product path) at the BASELINE
shapes -- C2 (B=1), C4 shard (B=8), C3 (B=8, 11 planes), C1, C5 -- on a volume built from random features.
    python tools/k7_probe.py [--reps 20] [--shapes c2,c4,c3,c1,c5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip  # noqa: E402
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation  # noqa: E402

SHAPES = {   # name: (B, h, w, levels of the event targets, image target levels or None, degree)
    "c2": (1, 60, 80, [1, 1, 1, 4], None, 2),
    "c4": (8, 60, 80, [1, 1, 1, 4], None, 2),
    "c3": (8, 60, 80, [1, 1, 1, 4], 4, 2),
    "c1": (1, 48, 48, [1, 1, 1, 1, 4], None, 10),
    "c5": (1, 128, 128, [1, 1, 1, 1, 4], 4, 10),
}


def timeit(fn, reps):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", default="c2,c4,c3,c1,c5")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    D = 256
    for name in args.shapes.split(","):
        B, h, w, lv, img_lv, deg = SHAPES[name]
        N = h * w
        T = len(lv)
        f1 = torch.randn((B, D, h, w), generator=g).to(dev)
        f2 = torch.randn((T, B, D, h, w), generator=g).to(dev)
        cc = CorrComputation(f1, f2, lv)
        cimg = None
        if img_lv is not None:
            cimg = CorrComputation(torch.randn((B, D, h, w), generator=g).to(dev), torch.randn((1, B, D, h, w), generator=g).to(dev), [img_lv])
        Tall = T + (1 if cimg is not None else 0)
        params = (torch.randn((B, 2 * deg, h, w), generator=g) * 3).to(dev)
        params[:, :, 0, :4] *= 40.0      # a few far-out-of-plane look-ups (all-zero windows) and border cases
        coef = hip.bezier_coeffs([(i + 1) / Tall for i in range(Tall)], deg)
        outs = {}
        for layout in ("rows", "tiled", "tiled-f16"):
            blk = CorrBlockParallelMultiTarget(corr_computation_events=cc, corr_computation_frames=cimg, layout=layout.split("-")[0],
                                               precision="f16" if layout.endswith("f16") else None)
            out = blk.new_output_split()
            P = blk.num_planes
            by = B * N * P * ((2.0 if layout.endswith("f16") else 4.0) * 100 + 4.0 * 81)
            ms, mn = timeit(lambda: blk.lookup_bezier_split(params, coef, out), args.reps)
            gms = graph_time(lambda: blk.lookup_bezier_split(params, coef, out))
            print(f"{name} {layout:9s}: B={B} N={N} P={P} deg={deg}: events median {ms*1e3:.1f} us (min {mn*1e3:.1f}), in-graph {gms*1e3:.1f} us; "
                  f"{by/1e6:.1f} MB algorithmic -> {by/gms/1e6:.0f} GB/s = {by/gms/1e6/8000:.3f} of 8 TB/s", flush=True)
            outs[layout] = out.float_nhwc()
            del blk, out
            torch.cuda.empty_cache()
        d = (outs["rows"] - outs["tiled"]).abs().max().item()
        print(f"   max |rows - tiled| = {d:.3e} (values up to {outs['rows'].abs().max().item():.1f})", flush=True)
        del cc, cimg, f1, f2, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

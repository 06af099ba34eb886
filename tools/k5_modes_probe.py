#!/usr/bin/env python
"""K5 arithmetic / storage variants (tools only): accuracy of the tiled volume against an fp64 GEMM and duration at the BASELINE sizes.
    python tools/k5_modes_probe.py [--reps 20] [--big]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip  # noqa: E402

MODES = [("split", hip.ARITH_SPLIT, torch.float32), ("split8", hip.ARITH_SPLIT8, torch.float32), ("f16/w", hip.ARITH_F16, torch.float32),
         ("split/h", hip.ARITH_SPLIT, torch.float16), ("split8/h", hip.ARITH_SPLIT8, torch.float16), ("f16", hip.ARITH_F16, torch.float16)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--modes", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    modes = [m for m in MODES if not args.modes or m[0] in args.modes.split(",")]
    g = torch.Generator(device="cpu").manual_seed(3)
    # ---- accuracy on a small ragged shape (edge tiles, panel tails) and on C2's shape (one target checked in fp64)
    for (B, D, hh, ww, T, shared) in ((2, 256, 13, 21, 3, True), (1, 128, 24, 40, 2, False), (1, 256, 60, 80, 4, True)):
        N = hh * ww
        f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
        f2 = torch.randn((T * B, D, N), generator=g).to(dev)
        f2[0, :, 0] *= 3e-3
        f2[0, :, 1] *= 100.0
        p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
        a = f1.double().view(-1, B, D, N)
        a = a.expand(T, B, D, N) if shared else a
        b = f2.double().view(T, B, D, N)
        ref = (a.transpose(2, 3) @ b / np.sqrt(D))
        mag = (a.abs().transpose(2, 3) @ b.abs() / np.sqrt(D))
        for name, ar, dt in modes:
            vol = torch.full((T, B, N, hip.tiled_plane_size(hh, ww)), float("nan"), dtype=dt, device=dev)
            hip.corr_build_tiled(p1, p2, vol, T, B, N, shared_f1=shared, tiled_hw=(hh, ww), arithmetic=ar)
            out = hip.untile_planes(vol.float(), hh, ww).reshape(T, B, N, N).double()
            err = (out - ref).abs()
            print(f"  B={B} D={D} {hh}x{ww} T={T} shared={shared} {name:9s}: max err / sum|a||b| = {float((err / mag).max()):.2e}  rms err / rms value = "
                  f"{float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}  finite={bool(torch.isfinite(vol).all())}", flush=True)
    # ---- durations
    shapes = [("C2", 1, 4, 60, 80, True)]
    if args.big:
        shapes += [("C4 shard", 8, 4, 60, 80, True), ("C5", 1, 6, 128, 128, False)]
    D = 256
    for name, B, T, hh, ww, shared in shapes:
        N = hh * ww
        f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
        f2 = torch.randn((T * B, D, N), generator=g).to(dev)
        p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
        x8 = (hip.split_to_x8(p1), hip.split_to_x8(p2))
        for mname, ar, dt in modes:
            vol = torch.empty((T, B, N, hip.tiled_plane_size(hh, ww)), dtype=dt, device=dev)
            ms = timeit(lambda: hip.corr_build_tiled(p1, p2, vol, T, B, N, shared_f1=shared, tiled_hw=(hh, ww), arithmetic=ar, x8=x8), args.reps)
            eb = vol.element_size()
            by = 4.0 * (((1 if shared else T) + T) * B * D * N) * (0.5 if ar == hip.ARITH_F16 else 1.0) + eb * T * B * N * N
            print(f"{name} {mname:9s}: {ms*1e3:8.1f} us  {by/1e6:7.0f} MB algorithmic -> {by/ms/1e6:6.0f} GB/s = {by/ms/1e6/8000:.3f} of 8 TB/s", flush=True)
            del vol
        ms = timeit(lambda: hip.split_to_x8(p1), args.reps)
        print(f"{name} split_to_x8 (one operand group of {p1.shape[1]} maps): {ms*1e3:.1f} us", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

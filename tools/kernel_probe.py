#!/usr/bin/env python
"""Runs only the hand-written correlation kernels on DSEC-sized operands (config C2: T=4, B, D=256, N=60x80), for
`rocprofv3 --pmc ...` passes and quick A/B timing.  Not part of the product path.
    python tools/kernel_probe.py [--batch 1] [--reps 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip  # noqa: E402
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, T, D, h, w = args.batch, 4, 256, 60, 80
    N = h * w
    g = torch.Generator(device="cpu").manual_seed(0)
    f1 = torch.randn((B, D, h, w), generator=g).to(dev)
    f2 = torch.randn((T, B, D, h, w), generator=g).to(dev)
    blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, [1, 1, 1, 4]))
    vol = torch.empty((T, B, N, N), device=dev)
    params = (torch.randn((B, 4, h, w), generator=g) * 3).to(dev)
    coef = hip.bezier_coeffs([0.25, 0.5, 0.75, 1.0], 2)
    out = blk.new_output()

    def timeit(fn):
        for _ in range(3):
            fn()
        evs = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    ms = timeit(lambda: hip.corr_build_f32(f1.view(B, D, N), f2.view(T, B, D, N), vol))
    fl = 2.0 * T * B * D * N * N
    by = 4.0 * ((1 + T) * B * D * N + T * B * N * N)
    print(f"corr_build_f32   B={B}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({by/ms/1e6:7.1f} GB/s algorithmic)")
    f1v, f2v = f1.view(B, D, N), f2.view(T * B, D, N)
    ms_p = timeit(lambda: (hip.split_pack(f1v), hip.split_pack(f2v)))
    p1, p2 = hip.split_pack(f1v), hip.split_pack(f2v)
    ms = timeit(lambda: hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=True))
    print(f"corr_build_split B={B}: {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s algorithmic  ({fl/ms/1e9:7.1f} TFLOP/s-equivalent), pack {ms_p*1e3:.1f} us")
    ms = timeit(lambda: blk.lookup_bezier(params, coef, out=out))
    by = 4.0 * B * N * blk.num_planes * 181
    print(f"corr_lookup_bez  B={B}: {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s algorithmic")
    src = blk.pyramid_level(0)[0][3]
    dst = torch.empty((B * N, h // 2, w // 2), device=dev)
    ms = timeit(lambda: hip.corr_pool2x2(src.squeeze(1), dst))
    by = 4.0 * B * N * (N + N // 4)
    print(f"corr_pool2x2 L0  B={B}: {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()

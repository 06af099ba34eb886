#!/usr/bin/env python
"""Times the split-fp16 conv engine on the update-block / encoder shapes of config C2 (tools only)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--tile", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
B = args.batch
shapes = [("convc1 1x1 576->256", 576, 256, (1, 1), 1, (0, 0), 60, 80, B), ("convc2 3x3 256->192", 256, 192, (3, 3), 1, (1, 1), 60, 80, B),
          ("gru zr 1x5 288->256", 288, 256, (1, 5), 1, (0, 2), 60, 80, B), ("gru q 5x1 288->128", 288, 128, (5, 1), 1, (2, 0), 60, 80, B),
          ("gru zr 5x1 288->256", 288, 256, (5, 1), 1, (2, 0), 60, 80, B),
          ("head2 3x3 256->4", 256, 4, (3, 3), 1, (1, 1), 60, 80, B), ("enc l1 3x3 64->64 @240x320", 64, 64, (3, 3), 1, (1, 1), 240, 320, 5 * B),
          ("enc l3 3x3 128->128 @60x80", 128, 128, (3, 3), 1, (1, 1), 60, 80, 5 * B)]
for name, cin, cout, k, st, pad, H, W, n in shapes:
    x = S.from_nchw(torch.randn(n, cin, H, W, device=dev))
    pk = S.PackedConvWeight().get(torch.randn(cout, cin, *k, device=dev) * 0.05)
    kw = dict(stride=st, padding=pad, tile=(args.tile or None))
    for _ in range(3): S.conv(x, pk, **kw)
    torch.cuda.synchronize()
    # the launches are captured into a hipGraph: timing eager launches from Python measures the host, not the kernel
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): S.conv(x, pk, **kw)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    Ho, Wo = (H + 2 * pad[0] - k[0]) // st + 1, (W + 2 * pad[1] - k[1]) // st + 1
    fl = 2.0 * n * Ho * Wo * cout * cin * k[0] * k[1]
    print(f"{name:32s} n={n}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s-equivalent")

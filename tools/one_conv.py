#!/usr/bin/env python
"""Runs ONE conv shape of the split engine a few times (for rocprofv3 --pmc passes; tools only)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,64,3,3,240,320,5")   # cin,cout,kh,kw,H,W,n
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--stats", action="store_true")
a = ap.parse_args()
cin, cout, kh, kw, H, W, n = map(int, a.shape.split(","))
dev = torch.device("cuda:0")
x = S.from_nchw(torch.randn(n, cin, H, W, device=dev))
pk = S.PackedConvWeight().get(torch.randn(cout, cin, kh, kw, device=dev) * 0.05)
st = torch.zeros((n, cout, 2), dtype=torch.float64, device=dev) if a.stats else None
for _ in range(a.reps):
    S.conv(x, pk, padding=(kh // 2, kw // 2), stats=st, want_f32=a.stats, want_split=not a.stats)
torch.cuda.synchronize()

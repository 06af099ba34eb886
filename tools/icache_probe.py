#!/usr/bin/env python
"""Is the cold instruction cache part of the 4-5 us a small-grid convolution costs more inside an update iteration than alone?  The SAME set
of launches (7 kernel instances, ~150 KB of code, L2-hot operands) is timed in two orders inside one graph: grouped (each kernel 6 x in a row)
and interleaved (round robin).  Tools only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
dev = torch.device("cuda:0")
H, W = 60, 80
shapes = [(256, 128, 3, 3), (256, 128, 1, 5), (256, 128, 5, 1), (256, 256, 3, 3), (384, 256, 1, 5), (384, 256, 5, 1), (576, 256, 1, 1)]
ops = []
for cin, cout, kh, kw in shapes:
    x = S.from_nchw(torch.randn(1, cin, H, W, device=dev))
    pk = S.PackedConvWeight().get(torch.randn(cout, cin, kh, kw, device=dev) * 0.05)
    o, _ = S.conv(x, pk, padding=(kh // 2, kw // 2))
    ops.append((x, pk, (kh // 2, kw // 2), o))
def run(order):
    for i in order:
        x, pk, pad, o = ops[i]
        S.conv(x, pk, padding=pad, out_split=o)
R = 6
grouped = [i for i in range(len(ops)) for _ in range(R)]
inter = [i for _ in range(R) for i in range(len(ops))]
res = {}
for name, order in (("grouped", grouped), ("interleaved", inter)):
    run(order); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(order)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    res[name] = best * 1e3 / len(order)
    print(f"{name:12s}: {res[name]:6.2f} us per launch ({len(order)} launches)")
print(f"interleaved - grouped = {res['interleaved'] - res['grouped']:.2f} us per launch")

#!/usr/bin/env python
"""Graph-timed stem convolution (7x7/2) on the C2 operands, hand-written kernel vs the MIOpen library conv (tools only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def timed(fn, name, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f"{name:50s} {a.elapsed_time(b)/reps*1e3:8.1f} us")
for n, cin in ((5, 5), (40, 5), (1, 5), (5, 41)):
    x = torch.randn(n, cin, 480, 640, device=dev)
    w = torch.randn(64, cin, 7, 7, device=dev) * 0.05
    pk = S.PackedStemWeight().get(w)
    st = torch.zeros((n, 64, 2), dtype=torch.float64, device=dev)
    timed(lambda: S.conv_stem(x, pk, stats=st, want_split=False, want_f32=True), f"conv_stem n={n} cin={cin} (f32 + stats)")
    if cin == 5:
        timed(lambda: S.conv_stem(x, pk, stats=st, want_split=False, want_f32=True, layout=3), f"conv_stem n={n} cin={cin} (f32 + stats) per-patch form")
        timed(lambda: S.conv_stem(x, pk, stats=st, want_split=False, want_f32=True, layout=2), f"conv_stem n={n} cin={cin} (f32 + stats) PERSISTENT form")
    timed(lambda: S.conv_stem(x, pk, act=S.ACT_RELU), f"conv_stem n={n} cin={cin} (relu -> split)")
    timed(lambda: torch.nn.functional.conv2d(x, w, None, stride=2, padding=3), f"MIOpen conv2d n={n} cin={cin}")

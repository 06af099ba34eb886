#!/usr/bin/env python
"""Times ONE conv shape of the split engine inside a hipGraph (tools only): cin,cout,kh,kw,H,W,n; --stats = fp32 out + statistics (the
InstanceNorm convolutions of the encoder), default = split out."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,64,3,3,240,320,5")
ap.add_argument("--stats", action="store_true")
ap.add_argument("--replicas", type=int, default=8, help="statistics replicas (the encoder uses 8)")
ap.add_argument("--f32", action="store_true", help="fp32 output without statistics")
a = ap.parse_args()
cin, cout, kh, kw, H, W, n = map(int, a.shape.split(","))
dev = torch.device("cuda:0")
x = S.from_nchw(torch.randn(n, cin, H, W, device=dev))
pk = S.PackedConvWeight().get(torch.randn(cout, cin, kh, kw, device=dev) * 0.05)
st = torch.zeros((a.replicas, n, cout, 2), dtype=torch.float64, device=dev) if a.stats else None
kw_ = dict(padding=(kh // 2, kw // 2), stats=st, want_f32=a.stats or a.f32, want_split=not (a.stats or a.f32))
o = S.conv(x, pk, **kw_)
kw_.update(out_split=o[0], out_f32=o[1])
for _ in range(3): S.conv(x, pk, **kw_)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): S.conv(x, pk, **kw_)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
fl = 2.0 * n * H * W * cout * cin * kh * kw
print(f"{a.shape} stats={a.stats} (R={a.replicas}) f32={a.f32}: {best*1e3:8.1f} us  {fl/best/1e9:7.1f} TFLOP/s-equivalent")

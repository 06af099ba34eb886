#!/usr/bin/env python
"""norm_act_split_kernel (the feature encoder's block ends: relu(x + relu(norm2(conv2)))) alone, graph-timed (tools only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time
dev = torch.device("cuda:0")
for B, C, H, W in ((5, 64, 240, 320), (40, 64, 240, 320), (5, 96, 120, 160), (40, 96, 120, 160), (5, 128, 60, 80)):
    a = torch.randn(B, (C + 31) // 32, H * W, 32, device=dev)
    st = torch.zeros((8, B, C, 2), dtype=torch.float64, device=dev); st[0, :, :, 1] = H * W
    res = S.from_nchw(torch.randn(B, C, H, W, device=dev))
    out = S.SplitTensor.empty(B, H, W, C, dev)
    t = graph_time(lambda: S.norm_act(a, (B, H, W, C), stats_a=st, act_a=S.ACT_RELU, res=res, act_out=S.ACT_RELU, out=out))
    t2 = graph_time(lambda: S.norm_act(a, (B, H, W, C), stats_a=st, act_a=S.ACT_RELU, out=out))
    nb = a.numel() * 4 + res.planes.numel() * 2 + out.planes.numel() * 2
    nb2 = a.numel() * 4 + out.planes.numel() * 2
    print(f"B={B} C={C} {H}x{W}: block end (a + res -> out) {t*1e3:7.1f} us  {nb/t/1e9:6.2f} TB/s ({nb/1e6:.0f} MB);   norm+relu only {t2*1e3:7.1f} us {nb2/t2/1e9:6.2f} TB/s")

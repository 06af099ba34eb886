#!/usr/bin/env python
"""Probe (tools only): the gate convolutions of a batch-1 GRU half at DSEC size -- plain output vs fused gate epilogue, full input
[h | M] (256 channels) vs one half (128 channels): what the epilogue's operand loads and half of the k-loop cost in-graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

dev = torch.device("cuda:0")
B, H, W, hd = 1, 60, 80, 128
h = S.from_nchw(torch.randn(B, hd, H, W, device=dev))
m = S.from_nchw(torch.randn(B, hd, H, W, device=dev))
rh = S.SplitTensor.empty(B, H, W, hd, dev)
z = torch.empty((B, hd // 32, H * W, 32), device=dev)
hn = S.SplitTensor.empty(B, H, W, hd, dev)
for k, pad, nm in (((1, 5), (0, 2), "1x5"), ((5, 1), (2, 0), "5x1")):
    for cin, x2 in ((256, m), (128, None)):
        wzr = S.PackedConvWeight().get(torch.randn(2 * hd, cin, *k, device=dev) * 0.03)
        wq = S.PackedConvWeight().get(torch.randn(hd, cin, *k, device=dev) * 0.03)
        azr = torch.randn((B, 2 * hd // 32, H * W, 32), device=dev)
        aq = torch.randn((B, hd // 32, H * W, 32), device=dev)
        ozr = torch.empty_like(azr); oq = torch.empty_like(aq)
        t = {}
        t["zr plain f32"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, want_split=False, out_f32=ozr))
        t["zr +addend f32"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, addend=azr, want_split=False, out_f32=ozr))
        t["zr gate"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, addend=azr, gate=S.GATE_ZR, gate_h=h, out_split=rh, out_f32=z))
        t["q plain f32"] = graph_time(lambda: S.conv(h, wq, x2=x2, padding=pad, want_split=False, out_f32=oq))
        t["q gate"] = graph_time(lambda: S.conv(h, wq, x2=x2, padding=pad, addend=aq, gate=S.GATE_BLEND, gate_h=h, gate_z=z, out_split=hn))
        print(f"{nm} Cin={cin}: " + ", ".join(f"{a} {b*1e3:.1f} us" for a, b in t.items()), flush=True)

#!/usr/bin/env python
"""Probe (tools only): the batch-1 1x1 convolutions of the update block at DSEC size, alone in a graph -- convc1 (352 -> 256), the im2col'd
convf1 (224 -> 128), the mask head's 256 -> 576 (fp32 out) -- per channel tile and kernel, to see what the generic split-k kernel costs them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

dev = torch.device("cuda:0")
B, H, W = 1, 60, 80
for name, cin, cout, f32 in (("convc1", 352, 256, False), ("convf1", 224, 128, False), ("mask2", 256, 576, True), ("mask2-split", 256, 576, False)):
    x = S.from_nchw(torch.randn(B, cin, H, W, device=dev))
    w = S.PackedConvWeight().get(torch.randn(cout, cin, 1, 1, device=dev) * 0.05)
    bias = torch.randn(cout, device=dev)
    res = []
    for tile in (64, 96, 128):
        t = graph_time(lambda: S.conv(x, w, shift=bias, act=S.ACT_RELU, want_split=not f32, want_f32=f32, tile=tile))
        res.append(f"tile {tile}: {t*1e3:.1f} us")
    print(f"{name} {cin}->{cout}: " + ", ".join(res), flush=True)

#!/usr/bin/env python
"""Probe (tools only): in-graph duration of the thin head convolution (bflow_conv_thin_acc: 3x3, 256 -> 2*deg channels, P += dP, Bezier block of M)
at DSEC size, batch 1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

dev = torch.device("cuda:0")
B, H, W = 1, 60, 80
for cout in (4, 20):
    d1 = S.from_nchw(torch.randn(B, 256, H, W, device=dev))
    w = S.ThinConvWeight().get(torch.randn(cout, 256, 3, 3, device=dev) * 0.02)
    bias = torch.randn(cout, device=dev)
    bez = torch.zeros(B, cout, H, W, device=dev)
    M = S.SplitTensor.empty(B, H, W, 128, dev, zero=True)
    off = 128 - cout if cout % 4 == 0 and cout <= 32 and (128 - cout) // 32 == 3 else 96
    t = graph_time(lambda: S.conv_thin_acc(d1, w, bias, bez, out_split=M, channel_offset=off, mfma=False))
    line = f"thin head 3x3 256 -> {cout}: vector ALU {t*1e3:.1f} us in-graph"
    if w[2] is not None:
        t2 = graph_time(lambda: S.conv_thin_acc(d1, w, bias, bez, out_split=M, channel_offset=off, mfma=True))
        line += f", matrix cores (taps as output channels) {t2*1e3:.1f} us"
    print(line)

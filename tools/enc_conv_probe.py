#!/usr/bin/env python
"""Probe (tools only): the feature encoder's 3x3 convolutions alone in a graph -- plain split input (transposed accumulators, fp32 + statistics out)
and the normalise-on-load form -- at the three resolutions of the 5-image stack.  Used with BFLOW_HIP_LIB variants (e.g. -DHALO_FAKE_FEWER_READS=1:
WRONG results, a third fewer LDS fragment reads) to see what bounds them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

dev = torch.device("cuda:0")
for name, C, H, W in (("layer1", 64, 240, 320), ("layer2", 96, 120, 160), ("layer3", 128, 60, 80)):
    B = 5
    x = S.from_nchw(torch.randn(B, C, H, W, device=dev))
    w = S.PackedConvWeight().get(torch.randn(C, C, 3, 3, device=dev) * 0.05)
    raw = torch.randn(B, (C + 31) // 32, H * W, 32, device=dev)
    st_in = torch.zeros((8, B, C, 2), dtype=torch.float64, device=dev)
    st_in[0, :, :, 0] = 0.0; st_in[0, :, :, 1] = float(H * W)          # mean 0, var 1
    st = torch.zeros((8, B, C, 2), dtype=torch.float64, device=dev)
    o32 = torch.empty((B, (C + 31) // 32, H * W, 32), dtype=torch.float32, device=dev)
    t_plain = graph_time(lambda: S.conv(x, w, padding=1, want_split=False, out_f32=o32, stats=st))
    t_nin = graph_time(lambda: S.conv_norm_in(raw, (B, H, W, C), st_in, w, stats=st, out_f32=o32))
    fl = 2.0 * B * H * W * C * C * 9
    print(f"{name} {C}ch {H}x{W} x{B}: plain {t_plain*1e3:.1f} us ({fl/t_plain/1e9:.0f} TFLOP/s-eq), normalise-on-load {t_nin*1e3:.1f} us", flush=True)

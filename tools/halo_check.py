#!/usr/bin/env python
"""Ad-hoc correctness sweep of the split conv engine against torch's fp32 conv on the GPU (tools only)."""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [(288, 256, (1, 5), 60, 80, 1), (288, 128, (5, 1), 60, 80, 1), (256, 192, (3, 3), 60, 80, 1), (64, 64, (3, 3), 240, 320, 5),
         (96, 96, (3, 3), 120, 160, 5), (128, 128, (3, 3), 60, 80, 5), (128, 128, (3, 3), 60, 80, 8), (128, 32, (3, 3), 60, 80, 8),
         (288, 256, (1, 5), 60, 80, 8), (288, 128, (5, 1), 60, 80, 8), (576, 256, (1, 1), 60, 80, 8), (64, 96, (3, 3), 240, 320, 5)] * 3
for cin, cout, k, H, W, n in cases:
    torch.manual_seed(1)
    x = torch.randn(n, cin, H, W, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) / np.sqrt(cin * k[0] * k[1])
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=(k[0] // 2, k[1] // 2))
    xs = S.from_nchw(x)
    pk = S.PackedConvWeight().get(w)
    o = S.conv(xs, pk, stride=1, padding=(k[0] // 2, k[1] // 2))[0]
    got = o.to_nchw().double()
    err = (got - ref).abs()
    bad = (err > 1e-4).nonzero()
    msg = ""
    if len(bad):
        ys, xs_ = bad[:, 2].unique().tolist(), bad[:, 3].unique().tolist()
        msg = f"  BAD n={len(bad)} images={bad[:,0].unique().tolist()} ch[{bad[:,1].min().item()}..{bad[:,1].max().item()}] rows={ys[:12]} cols={xs_[:24]}"
    print(f"{cin}->{cout} {k} {H}x{W} n={n}: max err {err.max().item():.3e}{msg}")

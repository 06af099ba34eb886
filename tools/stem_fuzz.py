#!/usr/bin/env python
"""Random shapes through both forms of the row-window stem kernel (layout 2 = persistent, 3 = per-patch): outputs must be bit-identical,
statistics equal to summation order (tools only; the committed cases live in tests/test_hip_parity.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
dev = "cuda"
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(40):
    H, W = int(rs.randint(8, 200)) * 2 + int(rs.randint(0, 2)), int(rs.randint(8, 260)) * 2 + int(rs.randint(0, 2))
    windows = bool(rs.randint(0, 2))
    w = torch.from_numpy((rs.standard_normal((64, 5, 7, 7)) / 15).astype(np.float32)).to(dev)
    pk = S.PackedStemWeight().get(w)
    if windows:
        nb, nw = int(rs.randint(1, 3)), int(rs.randint(1, 6))
        src = torch.from_numpy(rs.standard_normal((nb, 5 + nw - 1 + int(rs.randint(0, 3)), H, W)).astype(np.float32)).to(dev)
        x = S.ChannelWindows(src, list(range(nw)), 5)
        n = nb * nw
    else:
        n = int(rs.randint(1, 7))
        x = torch.from_numpy((rs.standard_normal((n, 5, H, W)) * rs.choice([1.0, 1e-4, 300.0])).astype(np.float32)).to(dev)
    cout = 64
    bias = torch.from_numpy(rs.standard_normal(cout).astype(np.float32)).to(dev)
    outs = []
    for layout in (2, 3):
        st = torch.zeros((8, n, cout, 2), dtype=torch.float64, device=dev)
        _, f = S.conv_stem(x, pk, shift=bias, stats=st, want_split=False, want_f32=True, layout=layout)
        outs.append((f, st.sum(0)))
    same = torch.equal(outs[0][0], outs[1][0])
    sd = float((outs[0][1] - outs[1][1]).abs().max() / (outs[1][1].abs().max() + 1e-30))
    if not same or sd > 2e-6:
        bad += 1
    print(f"{it:2d} {n} x 5 x {H} x {W} windows={windows}: bit-identical {same}, statistics rel diff {sd:.1e}")
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)

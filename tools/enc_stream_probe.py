#!/usr/bin/env python
"""A/B of the encoder's 3x3 launches: conv_halo_stream_kernel (persistent, round 5) vs conv_halo_kernel (per-item) on the feature encoder's
shapes, graph-timed on the same box in alternating order (tools only).  BFLOW_CONV_KERNEL=halo selects the per-item kernel per call."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


SHAPES = ((64, 64, 240, 320, 5), (96, 96, 120, 160, 5), (128, 128, 60, 80, 5), (64, 64, 240, 320, 40), (96, 96, 120, 160, 40))
if os.environ.get("ENC_PROBE_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["ENC_PROBE_ONLY"].split(",")]
for cin, cout, H, W, B in SHAPES:
    x = S.from_nchw(torch.relu(torch.randn(B, cin, H, W, device=dev)))
    pk = S.PackedConvWeight().get(torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
    st = torch.zeros((8, B, cout, 2), dtype=torch.float64, device=dev)
    o32 = torch.empty((B, (cout + 31) // 32, H * W, 32), dtype=torch.float32, device=dev)
    fn = lambda: S.conv(x, pk, stride=1, padding=1, want_split=False, out_f32=o32, stats=st)
    if os.environ.get("ENC_PROBE_NIN"):     # the normalise-on-load form (x_raw): conv2 of a residual block
        raw = torch.randn(B, cin // 32, H * W, 32, device=dev) * 3 + 0.5
        st_in = torch.zeros((8, B, cin, 2), dtype=torch.float64, device=dev)
        st_in[0, :, :, 0] = raw.double().sum(dim=2).reshape(B, cin) * 0 + H * W * 0.5
        st_in[0, :, :, 1] = H * W * 9.25
        fn = lambda: S.conv_norm_in(raw, (B, H, W, cin), st_in, pk, stats=st, out_f32=o32)
    res = {}
    for rnd in range(2):
        for tag in ("stream", "halo"):
            if tag == "halo": os.environ["BFLOW_CONV_KERNEL"] = "halo"
            else: os.environ.pop("BFLOW_CONV_KERNEL", None)
            res.setdefault(tag, []).append(timed(fn))
    os.environ.pop("BFLOW_CONV_KERNEL", None)
    fl = 2.0 * B * H * W * cout * cin * 9
    print(f"{cin}->{cout} 3x3 {B}x{H}x{W}: stream {min(res['stream']):7.1f} us ({fl / min(res['stream']) / 1e6 / 833.3:.3f})  per-item {min(res['halo']):7.1f} us ({fl / min(res['halo']) / 1e6 / 833.3:.3f})  "
          f"all: {[round(v, 1) for v in res['stream']]} vs {[round(v, 1) for v in res['halo']]}")

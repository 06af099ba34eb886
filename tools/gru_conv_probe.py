#!/usr/bin/env python
"""Probe (tools only): the gate convolutions of a batch-1 GRU half at DSEC size -- plain output vs fused gate epilogue, full input
[h | M] (256 channels) vs one half (128 channels): what the epilogue's operand loads and half of the k-loop cost in-graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

STAMPS = "--stamps" in sys.argv      # needs the H8_STAMPS build (tools/conv_stamps.sh) as BFLOW_HIP_LIB
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
        bias = torch.randn(2 * hd, device=dev)
        t["zr +bias f32"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, shift=bias, want_split=False, out_f32=ozr))
        t["zr +addend f32"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, addend=azr, want_split=False, out_f32=ozr))
        t["zr gate"] = graph_time(lambda: S.conv(h, wzr, x2=x2, padding=pad, addend=azr, gate=S.GATE_ZR, gate_h=h, out_split=rh, out_f32=z))
        t["q plain f32"] = graph_time(lambda: S.conv(h, wq, x2=x2, padding=pad, want_split=False, out_f32=oq))
        t["q gate"] = graph_time(lambda: S.conv(h, wq, x2=x2, padding=pad, addend=aq, gate=S.GATE_BLEND, gate_h=h, gate_z=z, out_split=hn))
        print(f"{nm} Cin={cin}: " + ", ".join(f"{a} {b*1e3:.1f} us" for a, b in t.items()), flush=True)


if STAMPS:
    import ctypes
    import numpy as np
    from bflow_amd import hip
    wq = S.PackedConvWeight().get(torch.randn(hd, 256, 1, 5, device=dev) * 0.03)
    aq = torch.randn((B, hd // 32, H * W, 32), device=dev)
    st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device=dev)
    for label, fn in (("q 1x5 Cin=256, gate epilogue", lambda: S.conv(h, wq, x2=m, padding=(0, 2), addend=aq, gate=S.GATE_BLEND, gate_h=h, gate_z=z, out_split=hn)),
                      ("q 1x5 Cin=256, plain fp32 output", lambda: S.conv(h, wq, x2=m, padding=(0, 2), want_split=False, out_f32=torch.empty_like(aq)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        hip.lib().bflow_conv_set_stamp_buffer(ctypes.c_void_p(st.data_ptr()))
        for _ in range(2):
            st.zero_(); fn(); torch.cuda.synchronize()
        hip.lib().bflow_conv_set_stamp_buffer(None)
        a = st.cpu().numpy().reshape(-1, 16)
        a = a[a[:, 0] != 0]
        rt0, rt1 = a[:, 14], a[:, 15]
        print(f"{label}: {a.shape[0] // 8} workgroups; wall {(rt1.max() - rt0.min()) * 0.01:.1f} us (first start -> last end), start spread {(rt0.max() - rt0.min()) * 0.01:.1f} us, "
              f"per-wave life {np.median(rt1 - rt0) * 0.01:.1f} us median / {(rt1 - rt0).max() * 0.01:.1f} max, clock {np.median((a[:, 6] - a[:, 0]) / ((rt1 - rt0) * 10e-9)) / 1e9:.2f} GHz")
        for i, nme in enumerate(["issue of the first halo + weights", "first wait + barrier (prologue)", "channel block 0 (2 steps)", "channel blocks 1 .. 7",
                                 "drain + barrier", "epilogue (incl. store drain)"]):
            d_ = a[:, i + 1] - a[:, i]
            print(f"   {nme:36s} median {np.median(d_):7.0f} cycles   max {d_.max():7.0f}")

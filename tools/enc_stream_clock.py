#!/usr/bin/env python
"""Shader clock and cycles of conv_halo_stream_kernel (tools only): needs an H8_STAMPS build as BFLOW_HIP_LIB (tools/build_flag_variant.sh
<name> "-DH8_STAMPS [-DCSTREAM_ABL=n]" conv_split.hip).  Every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at
its start and end: clock = cycles / wall, so a time difference between two builds splits into 'fewer cycles' and 'a different clock'."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S, hip
dev = torch.device("cuda:0")
B = int(os.environ.get("CLK_B", "40"))
cin = cout = 64; H, W = 240, 320
x = S.from_nchw(torch.relu(torch.randn(B, cin, H, W, device=dev)))
pk = S.PackedConvWeight().get(torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
st = torch.zeros((8, B, cout, 2), dtype=torch.float64, device=dev)
o32 = torch.empty((B, 2, H * W, 32), dtype=torch.float32, device=dev)
fn = lambda: S.conv(x, pk, stride=1, padding=1, want_split=False, out_f32=o32, stats=st)
if os.environ.get("ENC_PROBE_NIN"):
    raw = torch.randn(B, cin // 32, H * W, 32, device=dev) * 3 + 0.5
    st_in = torch.zeros((8, B, cin, 2), dtype=torch.float64, device=dev); st_in[0, :, :, 0] = H * W * 0.5; st_in[0, :, :, 1] = H * W * 9.25
    fn = lambda: S.conv_norm_in(raw, (B, H, W, cin), st_in, pk, stats=st, out_f32=o32)
for _ in range(5): fn()
torch.cuda.synchronize()
stamps = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
hip.lib().bflow_conv_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
for _ in range(10): fn()       # the last launch's stamps stay: warm clocks
torch.cuda.synchronize()
hip.lib().bflow_conv_set_stamp_buffer(None)
s = stamps.cpu().numpy().reshape(-1, 4); s = s[s[:, 0] != 0]
cyc = (s[:, 2] - s[:, 0]).astype(np.float64); wall = (s[:, 3] - s[:, 1]).astype(np.float64) * 10e-9
span = (s[:, 3].max() - s[:, 1].min()) * 10e-9
print(f"{os.environ.get('CLK_TAG', '')}: workgroups {len(s)}; cycles per workgroup median {np.median(cyc):.0f}; lifetime median {np.median(wall) * 1e6:.1f} us; "
      f"launch span {span * 1e6:.1f} us; shader clock = cycles / wall: median {np.median(cyc / wall) / 1e9:.3f} GHz")
if os.environ.get("CLK_HIST"):
    t0 = s[:, 1].min()
    st_us = (s[:, 1] - t0) * 0.01; en_us = (s[:, 3] - t0) * 0.01
    order = np.argsort(st_us)
    print("start us percentiles 0/25/50/75/90/100:", [round(float(np.percentile(st_us, q)), 1) for q in (0, 25, 50, 75, 90, 100)])
    print("end   us percentiles 0/25/50/75/90/100:", [round(float(np.percentile(en_us, q)), 1) for q in (0, 25, 50, 75, 90, 100)])
    late = st_us > 0.1 * span * 1e6
    print(f"workgroups starting later than 10 % of the span: {int(late.sum())}; their lifetime median {np.median(wall[late]) * 1e6 if late.any() else 0:.1f} us vs {np.median(wall[~late]) * 1e6:.1f} us")
    idx = np.nonzero(s[:, 0] != 0)[0] if False else np.arange(len(s))
    print("first 24 workgroups (start, end):", [(round(float(a), 1), round(float(b), 1)) for a, b in zip(st_us[:24], en_us[:24])])
    print("late ones (index, start, end):", [(int(i), round(float(st_us[i]), 1), round(float(en_us[i]), 1)) for i in np.nonzero(late)[0][:40]])
if os.environ.get("CLK_HIST"):
    ids = np.nonzero(stamps.cpu().numpy().reshape(-1, 4)[:, 0] != 0)[0]
    for x in range(8):
        m = (ids % 8) == x
        print(f"xcd {x}: end us median {np.median(en_us[m]):.1f} min {en_us[m].min():.1f} max {en_us[m].max():.1f}; cycles median {np.median(cyc[m]):.0f}; clock {np.median((cyc / wall)[m]) / 1e9:.3f}")
    slow = en_us > 0.9 * en_us.max()
    print("slow workgroups:", int(slow.sum()), "ids (first 64):", ids[slow][:64].tolist())
    print("slow: cycles median", np.median(cyc[slow]), "fast: cycles median", np.median(cyc[~slow]))

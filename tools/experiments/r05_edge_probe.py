#!/usr/bin/env python
"""The flag-synchronised edge (tools only): the q convolution of the GRU's first half (1x5, gate = blend, writes the new
hidden state in place) and the z|r convolution of its second half (5x1, reads [h | M]) at 60 x 80, batch 1 (update.py:38-47), as
  (a) the two launches of the product path (12-wave q, 10-wave z|r),
  (b) two launches of the 8-wave kernel both (what the edge kernel is made of: BFLOW_CONV_KERNEL=halo8),
  (c) ONE launch with per-patch flags (bflow_conv_split_edge; a memset of the 40 flags rides in front, as it would once per frame).
Graph-timed (20 pairs per replay), alternating; the results of (c) are checked against (a)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import split as S
from k7_probe import graph_time

dev = torch.device("cuda:0")
B, H, W, hd = int(os.environ.get("EDGE_B", "1")), 60, 80, 128
g = torch.Generator(device="cpu").manual_seed(5)
def rnd(*shape, s=1.0): return (torch.randn(shape, generator=g) * s).to(dev)
h0 = rnd(B, hd, H, W, s=0.5)
rh0 = rnd(B, hd, H, W, s=0.5)
m = S.from_nchw(rnd(B, 160, H, W, s=0.5))
wq = S.PackedConvWeight().get(rnd(hd, hd + 160, 1, 5, s=0.03))
wzr = S.PackedConvWeight().get(rnd(2 * hd, hd + 160, 5, 1, s=0.03))
aq = rnd(B, hd // 32, H * W, 32)
azr = rnd(B, 2 * hd // 32, H * W, 32)
z0 = torch.sigmoid(rnd(B, hd // 32, H * W, 32))
flags = torch.zeros((B * 8 * 5,), dtype=torch.int32, device=dev)


def fresh():
    return S.from_nchw(h0), S.from_nchw(rh0), z0.clone()


def two(hs, rhs, z):
    S.conv(rhs, wq, x2=m, padding=(0, 2), addend=aq, gate=S.GATE_BLEND, gate_h=hs, gate_z=z, out_split=hs)
    S.conv(hs, wzr, x2=m, padding=(2, 0), addend=azr, gate=S.GATE_ZR, gate_h=hs, out_split=rhs, out_f32=z)


def edge(hs, rhs, z):
    flags.zero_()
    _, _, fused = S.conv_edge(dict(x=rhs, packed=wq, x2=m, padding=(0, 2), addend=aq, gate=S.GATE_BLEND, gate_h=hs, gate_z=z, out_split=hs),
                              dict(x=hs, packed=wzr, x2=m, padding=(2, 0), addend=azr, gate=S.GATE_ZR, gate_h=hs, out_split=rhs, out_f32=z), flags)
    return fused


# ---- results: one pass of each form from the same state
ref = fresh(); two(*ref)
tst = fresh(); fused = edge(*tst)
torch.cuda.synchronize()
print("one-launch form ran:", fused, " flags (expect 4 everywhere):", sorted(set(flags.cpu().tolist())))
for name, a, b in (("h (new state)", ref[0].to_nchw(), tst[0].to_nchw()), ("r*h", ref[1].to_nchw(), tst[1].to_nchw()), ("z", ref[2], tst[2])):
    d = (a - b).abs().max().item()
    print(f"   {name:14s} max |two launches - edge| = {d:.3e}  (values up to {a.abs().max().item():.2f})")

# ---- timing (the state keeps evolving; timing only)
st = fresh()
res = {}
for rnd_ in range(3):
    os.environ.pop("BFLOW_CONV_KERNEL", None)
    res.setdefault("(a) product kernels, two launches", []).append(graph_time(lambda: two(*st)) * 1e3)
    os.environ["BFLOW_CONV_KERNEL"] = "halo8x16"
    res.setdefault("(b) 8-wave kernel both, two launches", []).append(graph_time(lambda: two(*st)) * 1e3)
    os.environ.pop("BFLOW_CONV_KERNEL", None)
    res.setdefault("(c) one launch, per-patch flags (+ memset)", []).append(graph_time(lambda: edge(*st)) * 1e3)
    res.setdefault("    the memset alone", []).append(graph_time(lambda: flags.zero_()) * 1e3)
for k, v in res.items():
    print(f"{k:46s} " + "  ".join(f"{x:6.1f}" for x in v) + "  us")

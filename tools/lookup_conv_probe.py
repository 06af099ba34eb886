#!/usr/bin/env python
"""Probe (tools only): the fused look-up + convc1 launch (bflow_corr_lookup_conv1x1) against the two separate launches, in-graph
durations at the BASELINE shapes.  BFLOW_LOOKUP_CONV_TP=<pixels per workgroup> overrides the launcher's choice (read once per process).
    python tools/lookup_conv_probe.py [--shapes c2,c4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip, split as S  # noqa: E402
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation  # noqa: E402
from k7_probe import SHAPES, graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c2,c4")
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--stamps", action="store_true", help="needs the LC_STAMPS build (tools/lookup_conv_stamps.sh) as BFLOW_HIP_LIB")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    D = 256
    for name in args.shapes.split(","):
        B, h, w, lv, img_lv, deg = SHAPES[name]
        T = len(lv)
        f1 = torch.randn((B, D, h, w), generator=g).to(dev)
        f2 = torch.randn((T, B, D, h, w), generator=g).to(dev)
        blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, lv), layout="tiled")
        params = (torch.randn((B, 2 * deg, h, w), generator=g) * 3).to(dev)
        coef = hip.bezier_coeffs([(i + 1) / T for i in range(T)], deg)
        C = blk.num_planes * 81
        weight = (torch.randn((args.cout, C, 1, 1), generator=g) / C ** 0.5).to(dev)
        bias = torch.randn(args.cout, generator=g).to(dev)
        packed = S.PackedConvWeight().get(weight)
        feat = blk.new_output_split()
        c1 = S.SplitTensor.empty(B, h, w, args.cout, dev)
        t_l = graph_time(lambda: blk.lookup_bezier_split(params, coef, feat))
        t_c = graph_time(lambda: S.conv(feat, packed, shift=bias, act=S.ACT_RELU, out_split=c1))
        t_s = graph_time(lambda: (blk.lookup_bezier_split(params, coef, feat), S.conv(feat, packed, shift=bias, act=S.ACT_RELU, out_split=c1)))
        line = f"{name}: P={blk.num_planes} look-up {t_l*1e3:.1f} us, convc1 {t_c*1e3:.1f} us, both {t_s*1e3:.1f} us"
        if blk.conv1x1_fusable(args.cout):
            t_f = graph_time(lambda: blk.lookup_bezier_conv1x1(params, coef, packed, bias, S.ACT_RELU, c1))
            line += f"; fused {t_f*1e3:.1f} us (TP env {os.environ.get('BFLOW_LOOKUP_CONV_TP', '-')})"
        print(line, flush=True)
        if args.stamps:
            import ctypes
            import numpy as np
            st = torch.zeros((4096 * 8 * 24,), dtype=torch.int64, device=dev)
            hip.lib().bflow_lookup_conv_set_stamp_buffer(ctypes.c_void_p(st.data_ptr()))
            for _ in range(3):
                st.zero_()
                blk.lookup_bezier_conv1x1(params, coef, packed, bias, S.ACT_RELU, c1)
                torch.cuda.synchronize()
            hip.lib().bflow_lookup_conv_set_stamp_buffer(None)
            a = st.cpu().numpy().reshape(-1, 24)
            a = a[a[:, 0] != 0]                              # one row per wave
            rt0, rt1 = a[:, 20], a[:, 21]
            wall = (rt1.max() - rt0.min()) * 10e-9
            cyc = (a[:, 17] - a[:, 0])
            print(f"   {a.shape[0] // 8} workgroups; kernel wall {wall*1e6:.1f} us (first start -> last end, s_memrealtime); start spread "
                  f"{(rt0.max() - rt0.min()) * 0.01:.1f} us; per-wave life {np.median(rt1 - rt0) * 0.01:.1f} us median, {(rt1 - rt0).max() * 0.01:.1f} max; "
                  f"clock {np.median(cyc / ((rt1 - rt0) * 10e-9)) / 1e9:.2f} GHz")
            names = ["zero+A", "sync", "gather1 issue", "W1 issue", "taps", "wait gather1", "barrier", "interp1", "barrier", "gather2 issue",
                     "gemm1+W2 issue", "wait gather2", "barrier", "interp2", "barrier", "gemm2", "epilogue"]
            for i, nme in enumerate(names):
                d = a[:, i + 1] - a[:, i]
                if (a[:, i + 1] == 0).any() or (a[:, i] == 0).any():
                    continue
                print(f"   {i:2d} {nme:16s} median {np.median(d):8.0f} cycles   max {d.max():8.0f}   (wave 0: {np.median(d[0::8]):8.0f})")


if __name__ == "__main__":
    main()

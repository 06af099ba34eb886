#!/usr/bin/env python
"""K7 phase ablation (tools only; WRONG results for abl != 0): in-graph duration of the tile look-up at C2 / C4 shard under BFLOW_LOOKUP_ABL
(1 no gather, 2 no interpolation, 4 no output, 8 no tap tables, 16 no swizzle) and BFLOW_LOOKUP_TP -- both read once per process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation
from k7_probe import graph_time
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
res = []
for name, B in (("c2", 1), ("c4", 8)):
    h, w, lv, deg, D = 60, 80, [1, 1, 1, 4], 2, 256
    f1 = torch.randn((B, D, h, w), generator=g).to(dev); f2 = torch.randn((4, B, D, h, w), generator=g).to(dev)
    blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, lv), layout="tiled")
    params = (torch.randn((B, 2 * deg, h, w), generator=g) * 3).to(dev)
    coef = hip.bezier_coeffs([0.25, 0.5, 0.75, 1.0], deg)
    out = blk.new_output_split()
    res.append(f"{name} {graph_time(lambda: blk.lookup_bezier_split(params, coef, out))*1e3:.1f} us")
    del blk, out, f1, f2
    torch.cuda.empty_cache()
print(f"ABL={os.environ.get('BFLOW_LOOKUP_ABL','0'):>2} TP={os.environ.get('BFLOW_LOOKUP_TP','2')}: " + ", ".join(res), flush=True)

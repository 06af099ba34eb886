#!/usr/bin/env python
"""K1 on the DSEC two-step shape (raw uint16 coordinates + rectification map, 5 bins, 3 M events per window), 40 calls, for
`rocprofv3 --kernel-trace --stats` (tools only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd.dsec import EventStream, TwoStepAssembler
H, W, bins = 480, 640, 5
rs = np.random.RandomState(21)
n = 20 * 360_000
ev = dict(x=rs.randint(0, W, n, dtype=np.int32).astype(np.uint16), y=rs.randint(0, H, n, dtype=np.int32).astype(np.uint16),
          p=rs.randint(0, 2, n, dtype=np.int32).astype(np.uint8), t=np.sort(rs.randint(1_000_000, 1_360_000, n)).astype(np.int64))
yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
rect = np.stack([xx * 1.01 - 3 + np.sin(yy / 40.0), yy * 0.99 + 2 + np.cos(xx / 50.0)], -1).astype(np.float32)
stream = EventStream(**ev)
asm = TwoStepAssembler(bins, H, W, rect)
asm.keep_last_window = False
ts = np.array([[1_030_000, 1_130_000], [1_130_000, 1_230_000]], dtype=np.int64)
for _ in range(3): asm.assemble(stream, ts, 1, check=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): asm.assemble(stream, ts, 1, check=False)
torch.cuda.synchronize()
print(f"two windows of {asm.window_descriptor(stream, 1_130_000, 1_230_000)[1]} events: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per assembly")

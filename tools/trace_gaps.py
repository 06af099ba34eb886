#!/usr/bin/env python
"""Idle time between the frames of a traced bench run (tools only): for each of the last frames, the gap between the previous frame's last kernel
(the up-sampling) and this frame's first kernel, and the frame period.   usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ups = [i for i, r in enumerate(rows) if "cvx_upsample" in r["Kernel_Name"]]
for a, b in list(zip(ups[:-1], ups[1:]))[-6:]:
    end_prev = int(rows[a]["End_Timestamp"])
    first = rows[a + 1]
    gap = (int(first["Start_Timestamp"]) - end_prev) / 1e3
    period = (int(rows[b]["End_Timestamp"]) - end_prev) / 1e3
    busy = max(int(r["End_Timestamp"]) for r in rows[a + 1:b + 1]) - min(int(r["Start_Timestamp"]) for r in rows[a + 1:b + 1])
    print(f"gap before the frame's first kernel ({first['Kernel_Name'][:40]}) {gap:7.1f} us; frame period {period:8.1f} us; first-to-last kernel {busy/1e3:8.1f} us")

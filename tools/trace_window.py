#!/usr/bin/env python
"""Prints the kernel timeline (start offset, duration) of the last forward in a rocprofv3 kernel-trace CSV (tools only)."""
import csv, sys
path = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 0; count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "cvx_upsample" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
sel = rows[lo:ends[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
for r in sel[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  gap {(s-prev_end)/1e3:6.1f}  q={r.get('Queue_Id','?'):>3}  {n}")
    prev_end = max(prev_end, e)

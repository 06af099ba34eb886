#!/usr/bin/env python
"""Per-launch table of ONE steady-state update iteration from a rocprofv3 --kernel-trace CSV (tools only): every kernel between two consecutive
look-up launches of the last forward -- name, queue, workgroups (grid / workgroup size), duration -- and the iteration's span.
    usage: trace_iteration.py <kernel_trace.csv> [iteration index counted from the end, default 3]"""
import csv, sys
path = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
look = [i for i, r in enumerate(rows) if "corr_lookup_tile_kernel" in r["Kernel_Name"]]
a, b = look[-back - 1], look[-back]
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if n.startswith("_ZN"):
        for key in ("corr_lookup_tile_kernel", "im2col_small_kernel", "conv_thin_mfma_kernel"):
            if key in n: return key
    return n[:44]
print(f"{'start us':>9} {'us':>6} {'queue':>5} {'workgroups':>10} {'threads':>7}  kernel")
tot = 0.0
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    wgs, wx = 1, 1
    for d_ in "XYZ":
        g_, w_ = int(r.get(f"Grid_Size_{d_}", 1) or 1), int(r.get(f"Workgroup_Size_{d_}", 1) or 1)
        wgs *= max(1, g_ // max(w_, 1))          # rocprofv3 reports the grid in work-items
        wx *= max(w_, 1)
    gx = wgs * wx
    tot += (e - s) / 1e3
    print(f"{s/1e3:9.1f} {(e-s)/1e3:6.1f} {r.get('Queue_Id','?'):>5} {gx // max(wx,1):10d} {wx:7d}  {short(r['Kernel_Name'])}")
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"iteration span (look-up to look-up, under the tracer) {span:.1f} us; sum of kernel durations {tot:.1f} us (two queues overlap: the main chain is queue of the look-up)")

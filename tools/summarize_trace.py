#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 kernel-trace CSV over the LAST `reps` forwards (tools only).
A forward ends with the convex-upsampling kernel; everything before the end of upsample launch #(n - reps) is warm-up."""
import csv, sys, collections
path, reps = sys.argv[1], int(sys.argv[2])
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:100]
ends = [int(r["End_Timestamp"]) for r in rows if "cvx_upsample" in r["Kernel_Name"]]
t_begin = ends[len(ends) - reps - 1] if len(ends) > reps else 0
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) < t_begin: continue
    n = short(r["Kernel_Name"])
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
print(f"total kernel time per forward: {sum(tot.values())/reps/1e6:.3f} ms  ({sum(cnt.values())/reps:.0f} launches)")
for n, t in tot.most_common(45):
    print(f"{t/reps/1e3:9.1f} us  {cnt[n]/reps:6.1f} x {t/cnt[n]/1e3:8.1f} us  {n}")

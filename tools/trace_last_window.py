#!/usr/bin/env python
"""Per-kernel totals inside the LAST `window_ms` of a rocprofv3 --kernel-trace CSV (= the last graph replay of a probe): tools only.
   usage: python tools/trace_last_window.py <kernel_trace.csv> <window_ms> [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) >= end - int(float(sys.argv[2]) * 1e6)]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r["Kernel_Name"]]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
print(f"window {sys.argv[2]} ms: {len(win)} kernels, busy {sum(v[0] for v in agg.values()) / 1e6:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    k = k.replace("(anonymous namespace)::", "").replace("at::native::", "")
    print(f"{v[0] / 1e6:7.2f} ms {v[1]:5d} calls avg {v[0] / v[1] / 1e3:7.1f} us  {k[:120]}")

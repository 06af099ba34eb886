#!/usr/bin/env python
"""Per-launch table of the parts of ONE steady-state frame outside the update loop, from a rocprofv3 --kernel-trace CSV (tools only):
  encoder : every kernel from the frame's first launch to its first look-up
  tail    : every kernel from the frame's last look-up to the frame's last launch (last iteration + mask head + up-sampling)
    usage: trace_frame.py <kernel_trace.csv> encoder|tail [frames counted from the end, default 2] [look-ups per frame, default 12]"""
import csv, sys
path, part = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
per = int(sys.argv[4]) if len(sys.argv) > 4 else 12
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
look = [i for i, r in enumerate(rows) if "corr_lookup_tile_kernel" in r["Kernel_Name"]]
first = look[-back * per]              # first look-up of the chosen frame
last = look[-back * per + per - 1]     # its last look-up
prev_last = look[-back * per - 1]      # last look-up of the frame before
nxt_first = look[-back * per + per] if back > 1 else len(rows)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("_ZN"):
        for key in ("corr_lookup_tile_kernel", "im2col_small_kernel", "conv_thin_mfma_kernel", "corr_stream_kernel", "cvx_upsample", "voxel"):
            if key in n: return key
    return n[:60]
# a frame starts with the first kernel that follows the previous frame's up-sampling
ups = [i for i, r in enumerate(rows) if "cvx_upsample" in r["Kernel_Name"]]
start = max(i for i in ups if i < first) + 1
end = min(i for i in ups if i > last)
sel = rows[start:first] if part == "encoder" else rows[last:end + 1]
t0 = int(sel[0]["Start_Timestamp"])
print(f"{'start us':>9} {'us':>7} {'queue':>5} {'workgroups':>10} {'threads':>7}  kernel")
tot = 0.0
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    wgs, wx = 1, 1
    for d_ in "XYZ":
        g_, w_ = int(r.get(f"Grid_Size_{d_}", 1) or 1), int(r.get(f"Workgroup_Size_{d_}", 1) or 1)
        wgs *= max(1, g_ // max(w_, 1)); wx *= max(w_, 1)
    tot += (e - s) / 1e3
    print(f"{s/1e3:9.1f} {(e-s)/1e3:7.1f} {r.get('Queue_Id','?'):>5} {wgs:10d} {wx:7d}  {short(r['Kernel_Name'])}")
span = (int(sel[-1]["End_Timestamp"]) - t0) / 1e3
print(f"{part}: span {span:.1f} us under the tracer; sum of kernel durations {tot:.1f} us; {len(sel)} launches")

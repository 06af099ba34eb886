import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) >= end - int(39.4e6)]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = r["Kernel_Name"]
    if "conv_split_kernel" in n or "wgrad_pack" in n or "wgrad_reduce" in n or "conv_stem" in n:
        key = (n.split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Z", ""), r.get("LDS_Block_Size", ""))
        a = agg[key]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{v[0]/1e3:8.1f} us {v[1]:3d} calls avg {v[0]/v[1]/1e3:7.1f} us  {k}")
print(list(rows[0].keys()))

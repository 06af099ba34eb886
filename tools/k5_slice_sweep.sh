#!/bin/bash
# K5 (corr_stream_kernel) at C2 for several row-slice limits of the XCD grid (BFLOW_CORR_SLICE_KB): duration and fabric reads (tools only)
cd /tmp; export TMPDIR=/tmp
for kb in ${1:-2560 1280 640}; do
  rm -rf /tmp/kt$kb
  BFLOW_CORR_SLICE_KB=$kb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$kb -o k -- python $GRAFT_REPO_ROOT/tools/roofline_probe.py --key roofline_corr_build --reps 40 > /dev/null 2>&1
  python - $(find /tmp/kt$kb -name "*kernel_stats.csv" | head -1) $kb <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "corr_stream" in r["Name"]:
        print(f"slice limit {sys.argv[2]} KB: {r['Calls']} calls avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f}")
PY
done

#!/bin/bash
# Runs ON the GPU box (via gpurun): produces the evidence files of profiles/ under gpurun_out/profiles/ (tools only).
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$REPO/gpurun_out/profiles"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for key in roofline roofline_corr_build roofline_lookup; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -- python "$REPO/tools/roofline_probe.py" --key $key > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); cp "$f" "$OUT/${key}_${c}.csv"
  done
done
python "$REPO/tools/pmc_to_json.py" "$OUT" "$OUT/r02_pmc.json" > /dev/null
# the bench quotes the PMC traffic of kernels built from the SAME sources (kernel_source_hash): counters first, then the bench reads them
cp "$OUT/r02_pmc.json" "$REPO/profiles/r02_pmc.json"
python "$REPO/bench.py" --steps 30 --warmup 5 > "$OUT/r02_bench.json" 2> "$OUT/bench.stderr"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -46 "$f" > "$OUT/r02_rocprofv3_kernel_stats.csv"
# keep only the rows of the three kernels in the committed CSVs
python - "$OUT" <<'PY'
import csv, sys, os
d = sys.argv[1]
for key, rx in (("roofline", "conv_halo_kernel"), ("roofline_corr_build", "corr_stream_kernel"), ("roofline_lookup", "corr_lookup_tile_kernel")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, f"{key}_{c}.csv")
        rows = list(csv.DictReader(open(p)))
        keep = [r for r in rows if rx in r["Kernel_Name"]][-5:]
        with open(os.path.join(d, f"r02_pmc_{c}_{key}.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        os.remove(p)
PY
# training path: step time at the reference's DSEC training shape (eager = host-bound, and as one hipGraph) + the kernel breakdown of the graphed step
BFLOW_TRAIN_PROBE_GRAPH=1 python "$REPO/tools/train_probe.py" 10 2>/dev/null | grep -E "train step|hipGraph" > "$OUT/r02_train_probe.txt"
rm -rf /tmp/ktt; BFLOW_TRAIN_PROBE_GRAPH=engine rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o train -- python "$REPO/tools/train_probe.py" 5 > /tmp/ktt.log 2>&1
f=$(find /tmp/ktt -name "*kernel_stats.csv" | head -1); head -41 "$f" > "$OUT/r02_train_rocprofv3_kernel_stats.csv"
ls -la "$OUT"

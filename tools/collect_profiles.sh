#!/bin/bash
# Runs ON the GPU box (via gpurun): produces the round-6 evidence files of profiles/ under gpurun_out/profiles/ (tools only; the only
# writer of profiles/r06_pmc*, r06_bench.json, r06_rocprofv3_*; the A/B files quoted in DESIGN.md section 12 were written by the commands
# named in them).     usage: gpurun --timeout 3000 -- tools/collect_profiles.sh
# FAILS if the MFMA calibration pass or any kernel's counters are missing (tools/pmc_to_json.py): a table with holes is not evidence.
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$REPO/gpurun_out/profiles"; rm -rf "$OUT"; mkdir -p "$OUT"
R=r06
cd /tmp && export TMPDIR=/tmp
# the measurement binaries are git-ignored: build them HERE if the snapshot did not carry them (round 5 lost its calibration pass to that)
for t in fp8_cross store_patterns; do
  [ -x "$REPO/tools/micro/$t" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o "$REPO/tools/micro/$t" "$REPO/tools/micro/$t.hip" || { echo "collect_profiles: cannot build tools/micro/$t" >&2; exit 1; }
done
PROBE="python $REPO/tools/roofline_probe.py"
KEYS="${KEYS:-roofline roofline_update_conv roofline_corr_build roofline_corr_build_split roofline_lookup roofline_lookup_c4_shard roofline_corr_build_c5 roofline_conv3x3_c4 roofline_conv_stream_nin_c4 roofline_gru_conv_c4}"     # (KEYS=... QUICK=1: a dry run of the counter passes on a subset, nothing copied to profiles/)
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"

# ---- (1) fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots) and matrix-core busy cycles of
#      every roofline kernel; the probe's PROBE line carries name / regex / algorithmic bytes of the launch
for key in $KEYS; do
  for c in FETCH_SIZE WRITE_SIZE MFMA; do
    ctr="$c"; [ $c = MFMA ] && ctr="$SQ"
    rm -rf /tmp/pmc; timeout 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc -- $PROBE --key $key > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${key}_${c}.csv"
    grep "^PROBE " /tmp/pmc.log | tail -1 | sed 's/^PROBE //' > "$OUT/${key}_meta.json"
  done
done
# calibration launch (tools/micro/fp8_cross rate_kernel: a pure MFMA stream on every SIMD, 100 % busy by construction) under the same counters
rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc -- "$REPO/tools/micro/fp8_cross" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/calib_MFMA.csv"
python "$REPO/tools/pmc_to_json.py" "$OUT" "$OUT/${R}_pmc.json" > "$OUT/${R}_pmc_summary.txt" || { echo "collect_profiles: counters incomplete" >&2; cat "$OUT/${R}_pmc_summary.txt"; exit 1; }
# the bench quotes the PMC traffic of kernels built from the SAME sources (kernel_source_hash): counters first, then the bench reads them
[ -n "${QUICK:-}" ] && { cat "$OUT/${R}_pmc_summary.txt"; ls -la "$OUT"; exit 0; }
cp "$OUT/${R}_pmc.json" "$REPO/profiles/${R}_pmc.json"

# ---- (2) the bench line, and the kernel traces: C2 ONLY (the frame's budget per kernel) and the full default command
timeout 1200 python "$REPO/bench.py" --steps 30 --warmup 5 > "$OUT/${R}_bench.json" 2> "$OUT/bench.stderr"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -44 "$f" > "$OUT/${R}_rocprofv3_kernel_stats_c2only.csv"
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_iteration.py" "$f" > "$OUT/${R}_iteration_launches.txt" 2>&1
python "$REPO/tools/trace_frame.py" "$f" encoder > "$OUT/${R}_frame_encoder_launches.txt" 2>&1
python "$REPO/tools/trace_frame.py" "$f" tail > "$OUT/${R}_frame_tail_launches.txt" 2>&1
rm -rf /tmp/kt; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -50 "$f" > "$OUT/${R}_rocprofv3_kernel_stats.csv"
# stage boundaries INSIDE the captured graph, no tracer attached (the tracer serialises the two queues)
timeout 300 python "$REPO/tools/stamp_timeline.py" 2>/dev/null | grep -E " us |cnet" > "$OUT/${R}_stamp_timeline.txt"

# ---- (3) the sustained clock under matrix-core load, the store ceilings, K5 per arithmetic, correlation precision end to end
"$REPO/tools/micro/fp8_cross" > "$OUT/${R}_mfma_clock_fp8_cross.txt" 2>&1
"$REPO/tools/micro/store_patterns" > "$OUT/${R}_store_patterns.txt" 2>&1
timeout 600 python "$REPO/tools/k5_modes_probe.py" --big 2>/dev/null | grep -E "^C|sum" > "$OUT/${R}_k5_modes.txt"
timeout 600 python "$REPO/tools/corr_precision_probe.py" --c5 2>/dev/null | grep corr_precision > "$OUT/${R}_corr_precision_e2e.txt"
timeout 300 python "$REPO/tools/k1_probe.py" 2>/dev/null | grep "^K1" > "$OUT/${R}_k1_probe.txt"

# keep only the rows of the roofline kernels in the committed counter CSVs
python - "$OUT" "$R" <<'PY'
import csv, json, sys, os
d, R = sys.argv[1], sys.argv[2]
metas = {f[:-10]: json.load(open(os.path.join(d, f))) for f in os.listdir(d) if f.endswith("_meta.json")}
metas["calib"] = {"regex": "rate_kernel", "reps": 3}
for key, meta in metas.items():
    for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"):
        p = os.path.join(d, f"{key}_{c}.csv")
        if not os.path.exists(p): continue
        rows = list(csv.DictReader(open(p)))
        keep = [r for r in rows if meta["regex"] in r["Kernel_Name"]]
        keep = keep[-(int(meta.get("reps", 5)) * 2 * (4 if c == "MFMA" else 1)):] if key != "calib" else keep
        if rows:
            with open(os.path.join(d, f"{R}_pmc_{c}_{key}.csv"), "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        os.remove(p)
PY
ls -la "$OUT"

#!/bin/bash
# Runs ON the GPU box (via gpurun): produces the round-5 evidence files of profiles/ under gpurun_out/profiles/ (tools only; the only
# writer of profiles/r05_* except the A/B files quoted in DESIGN.md section 8, which were written by the commands named in them).
#   usage: gpurun --timeout 2400 -- tools/collect_profiles.sh          (variant libraries: tools/k5_ablate.sh stamps:-DSTREAM_STAMPS first)
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$REPO/gpurun_out/profiles"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROBE="python $REPO/tools/roofline_probe.py"
KEYS="roofline roofline_update_conv roofline_corr_build roofline_corr_build_split roofline_lookup"

# ---- (1) fabric traffic of the roofline kernels: FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
for key in $KEYS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -- $PROBE --key $key > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${key}_${c}.csv"
  done
done
# ---- (2) matrix-core utilisation of the same launches: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES + GRBM_GUI_ACTIVE, and the
#      calibration launch (tools/micro/fp8_cross: a pure MFMA stream, 100 % busy by construction) under the same counters
for key in $KEYS; do
  rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- $PROBE --key $key > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${key}_MFMA.csv"
done
rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- $REPO/tools/micro/fp8_cross > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/calib_MFMA.csv"
python "$REPO/tools/pmc_to_json.py" "$OUT" "$OUT/r05_pmc.json" > /dev/null
# the bench quotes the PMC traffic of kernels built from the SAME sources (kernel_source_hash): counters first, then the bench reads them
cp "$OUT/r05_pmc.json" "$REPO/profiles/r05_pmc.json"

# ---- (3) the bench line, and the kernel traces: C2 ONLY (the frame's budget per kernel) and the full default command
timeout 900 python "$REPO/bench.py" --steps 30 --warmup 5 > "$OUT/r05_bench.json" 2> "$OUT/bench.stderr"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -44 "$f" > "$OUT/r05_rocprofv3_kernel_stats_c2only.csv"
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_iteration.py" "$f" > "$OUT/r05_iteration_launches.txt" 2>&1
python "$REPO/tools/trace_frame.py" "$f" encoder > "$OUT/r05_frame_encoder_launches.txt" 2>&1
python "$REPO/tools/trace_frame.py" "$f" tail > "$OUT/r05_frame_tail_launches.txt" 2>&1
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -50 "$f" > "$OUT/r05_rocprofv3_kernel_stats.csv"
# stage boundaries INSIDE the captured graph, no tracer attached (the tracer serialises the two queues)
timeout 300 python "$REPO/tools/stamp_timeline.py" 2>/dev/null | grep -E " us |cnet" > "$OUT/r05_stamp_timeline.txt"

# ---- (4) K1 (tile-binned voxel grid): whole calls + the four kernels of one call
timeout 300 python "$REPO/tools/k1_probe.py" 2>/dev/null | grep "^K1" > "$OUT/r05_k1_probe.txt"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k1 -- python "$REPO/tools/k1_probe.py" 0 f > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); { echo "# rocprofv3 --kernel-trace --stats -- python tools/k1_probe.py 0 f   (2 M float-xy events into 15 x 480 x 640)"; head -6 "$f"; } >> "$OUT/r05_k1_probe.txt"

# ---- (5) the sustained clock under matrix-core load and K5's per-workgroup stamps (K5 itself is unchanged this round)
"$REPO/tools/micro/fp8_cross" > "$OUT/r05_mfma_clock_fp8_cross.txt" 2>&1
"$REPO/tools/micro/store_patterns" > "$OUT/r05_store_patterns.txt" 2>&1
if [ -f "$REPO/bflow_amd/lib/ab/libbflow_hip_stamps.so" ]; then
  for mode in split8 split; do
    BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_stamps.so" timeout 300 python "$REPO/tools/k5_probe.py" --time-only --stamps --stamp-mode $mode 2>/dev/null | sed -n '/stamped launch/,$p' > "$OUT/r05_k5_stamps_$mode.txt"
  done
fi
timeout 600 python "$REPO/tools/k5_modes_probe.py" --big 2>/dev/null | grep -E "^C|sum" > "$OUT/r05_k5_modes.txt"
timeout 600 python "$REPO/tools/corr_precision_probe.py" --c5 2>/dev/null | grep corr_precision > "$OUT/r05_corr_precision_e2e.txt"

# ---- (6) the encoder's persistent 3x3 kernel against the per-item kernel (plain and normalise-on-load), and K7 per pixels-per-workgroup
{ echo "# tools/enc_stream_probe.py (BFLOW_CONV_STREAM=all: the persistent kernel wherever it can run; 'per-item' = BFLOW_CONV_KERNEL=halo)"
  BFLOW_CONV_STREAM=all timeout 300 python "$REPO/tools/enc_stream_probe.py" 2>/dev/null | grep "3x3"
  echo "# the same with the input normalised on load (x_raw: conv2 of every residual block)"
  BFLOW_CONV_STREAM=all ENC_PROBE_NIN=1 timeout 300 python "$REPO/tools/enc_stream_probe.py" 2>/dev/null | grep "3x3"; } > "$OUT/r05_enc_stream_probe.txt"
{ for tp in 2 4 8; do echo "BFLOW_LOOKUP_TP=$tp"; BFLOW_LOOKUP_TP=$tp timeout 200 python "$REPO/tools/k7_probe.py" --shapes c2,c4 2>/dev/null | grep -E "tiled"; done; } > "$OUT/r05_k7_tp_probe.txt"
timeout 900 "$REPO/tools/ab_bench.sh" "BFLOW_CONV_STREAM=0" "BFLOW_CONV_STREAM=1" 2 > "$OUT/r05_stream_frame_ab.txt" 2>&1

# keep only the rows of the roofline kernels in the committed counter CSVs
python - "$OUT" <<'PY'
import csv, sys, os
d = sys.argv[1]
for key, rx in (("roofline", "conv_halo_stream_kernel"), ("roofline_update_conv", "conv_halo8_pair_kernel"), ("roofline_corr_build", "corr_stream_kernel"),
                ("roofline_corr_build_split", "corr_stream_kernel"), ("roofline_lookup", "corr_lookup_tile_kernel"), ("calib", "rate_kernel")):
    for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"):
        p = os.path.join(d, f"{key}_{c}.csv")
        if not os.path.exists(p): continue
        rows = list(csv.DictReader(open(p)))
        keep = [r for r in rows if rx in r["Kernel_Name"]]
        keep = keep[-(5 * (4 if c == "MFMA" else 1)):] if key != "calib" else keep
        if rows:
            with open(os.path.join(d, f"r05_pmc_{c}_{key}.csv"), "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        os.remove(p)
PY
# training path: unchanged this round, re-measured for regressions only
BFLOW_TRAIN_PROBE_GRAPH=1 timeout 600 python "$REPO/tools/train_probe.py" 10 2>/dev/null | grep -E "train step|hipGraph" > "$OUT/r05_train_probe.txt"
ls -la "$OUT"

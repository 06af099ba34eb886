#!/bin/bash
# Runs ON the GPU box (via gpurun): produces the round-4 evidence files of profiles/ under gpurun_out/profiles/ (tools only; the only
# writer of profiles/r04_*).   usage: gpurun --timeout 2400 -- tools/collect_profiles.sh
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$REPO/gpurun_out/profiles"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROBE="python $REPO/tools/roofline_probe.py"

# ---- (1) fabric traffic of the three roofline kernels: FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
for key in roofline roofline_encoder roofline_corr_build roofline_corr_build_split roofline_lookup; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -- $PROBE --key $key > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); cp "$f" "$OUT/${key}_${c}.csv"
  done
done
# ---- (2) matrix-core utilisation of the same launches: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES + GRBM_GUI_ACTIVE, and the
#      calibration launch (tools/micro/fp8_cross: a pure MFMA stream, 100 % busy by construction) under the same counters
for key in roofline roofline_encoder roofline_corr_build roofline_corr_build_split roofline_lookup; do
  rm -rf /tmp/pmc; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- $PROBE --key $key > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); cp "$f" "$OUT/${key}_MFMA.csv"
done
rm -rf /tmp/pmc; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -- $REPO/tools/micro/fp8_cross > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); cp "$f" "$OUT/calib_MFMA.csv"
python "$REPO/tools/pmc_to_json.py" "$OUT" "$OUT/r04_pmc.json" > /dev/null
# the bench quotes the PMC traffic of kernels built from the SAME sources (kernel_source_hash): counters first, then the bench reads them
cp "$OUT/r04_pmc.json" "$REPO/profiles/r04_pmc.json"

# ---- (3) the bench line, and the kernel traces: C2 ONLY (the 3.9-ms frame budget per kernel) and the full default command
python "$REPO/bench.py" --steps 30 --warmup 5 > "$OUT/r04_bench.json" 2> "$OUT/bench.stderr"
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -40 "$f" > "$OUT/r04_rocprofv3_kernel_stats_c2only.csv"
# one steady-state update iteration, launch by launch (workgroups, threads, us) from the same trace
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python "$REPO/tools/trace_iteration.py" "$f" > "$OUT/r04_iteration_launches_one_queue.txt" 2>&1
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -46 "$f" > "$OUT/r04_rocprofv3_kernel_stats.csv"
# stage boundaries INSIDE the captured graph, no tracer attached (the tracer serialises the two queues)
python "$REPO/tools/stamp_timeline.py" 2>/dev/null | grep " us " > "$OUT/r04_stamp_timeline.txt"

# ---- (4) the sustained clock under matrix-core load: (a) pure MFMA streams on random data (3-pass split vs fp8 cross terms): shader clock =
#      s_memtime / wall clock; (b) per-workgroup cycle stamps of K5 (split8 and the 3-pass split) at C2; (c) amd-smi / rocm-smi clock + power
#      sampled while K5 and the layer-1 convolution run back to back for ~2 s each
"$REPO/tools/micro/fp8_cross" > "$OUT/r04_mfma_clock_fp8_cross.txt" 2>&1
"$REPO/tools/micro/store_patterns" > "$OUT/r04_store_patterns.txt" 2>&1
for mode in split8 split; do
  BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_stamps.so" python "$REPO/tools/k5_probe.py" --time-only --stamps --stamp-mode $mode 2>/dev/null | sed -n '/stamped launch/,$p' > "$OUT/r04_k5_stamps_$mode.txt"
done
SMI=$(command -v amd-smi || command -v rocm-smi || true)
if [[ "$SMI" == *amd-smi ]]; then $SMI static --limit 2>/dev/null | grep -iE "POWER|GPU:" | head -12 > "$OUT/r04_power_limit.txt"; fi
for key in roofline_corr_build roofline_encoder; do
  ( for i in $(seq 1 12); do
      if [[ "$SMI" == *amd-smi ]]; then $SMI metric --clock --power 2>/dev/null | grep -E "GFX_0|CLK:|SOCKET_POWER|MIN_CLK|MAX_CLK" | head -8 | tr '\n' ' '; echo
      elif [ -n "$SMI" ]; then $SMI --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; fi
      sleep 0.25
    done ) > "$OUT/r04_smi_$key.txt" 2>&1 &
  SPID=$!
  $PROBE --key $key --reps 20000 > /dev/null 2>&1
  wait $SPID
done
python "$REPO/tools/k5_modes_probe.py" --big 2>/dev/null | grep -E "^C|sum" > "$OUT/r04_k5_modes.txt"
python "$REPO/tools/corr_precision_probe.py" --c5 2>/dev/null | grep corr_precision > "$OUT/r04_corr_precision_e2e.txt"

# the two restructurings of the update iteration that were built and measured this round
{ python "$REPO/tools/lookup_conv_probe.py" --shapes c2,c4 2>/dev/null | grep -E "^c[0-9]"
  BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_lcstamps.so" python "$REPO/tools/lookup_conv_probe.py" --shapes c2 --stamps 2>/dev/null | grep -vE "^c2:"
  for i in 1 2; do
    BFLOW_LOOKUP_CONV=1 python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench, fused look-up + convc1  :', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms per iteration')"
    python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench, separate launches (default):', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms per iteration')"
  done; } > "$OUT/r04_lookup_conv_probe.txt" 2>&1
{ python "$REPO/tools/gru_conv_probe.py" 2>/dev/null | grep Cin
  BFLOW_CONV_NO_HALO12=1 BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_h8stamps.so" python "$REPO/tools/gru_conv_probe.py" --stamps 2>/dev/null | grep -vE "Cin=|amdgpu"
  echo; echo "# the ten launches of one steady-state update iteration of the frame (rocprofv3 --kernel-trace of bench.py; = r04_iteration_launches_one_queue.txt):"
  cat "$OUT/r04_iteration_launches_one_queue.txt"; } > "$OUT/r04_gru_conv_probe.txt"

# keep only the rows of the three kernels in the committed counter CSVs
python - "$OUT" <<'PY'
import csv, sys, os
d = sys.argv[1]
for key, rx in (("roofline", "conv_halo8_pair_kernel"), ("roofline_encoder", "conv_halo_kernel"), ("roofline_corr_build", "corr_stream_kernel"),
                ("roofline_corr_build_split", "corr_stream_kernel"), ("roofline_lookup", "corr_lookup_tile_kernel"), ("calib", "rate_kernel")):
    for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"):
        p = os.path.join(d, f"{key}_{c}.csv")
        if not os.path.exists(p): continue
        rows = list(csv.DictReader(open(p)))
        keep = [r for r in rows if rx in r["Kernel_Name"]]
        keep = keep[-(5 * (4 if c == "MFMA" else 1)):] if key != "calib" else keep
        with open(os.path.join(d, f"r04_pmc_{c}_{key}.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        os.remove(p)
PY
# training path: unchanged this round, re-measured for regressions only
BFLOW_TRAIN_PROBE_GRAPH=1 python "$REPO/tools/train_probe.py" 10 2>/dev/null | grep -E "train step|hipGraph" > "$OUT/r04_train_probe.txt"
ls -la "$OUT"

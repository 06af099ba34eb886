set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_tail"; mkdir -p "$OUT"; rm -f "$OUT/ab.txt"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q -k "upsample or e2e or smoke or encoder or flow_init or golden" > "$OUT/tests_full.txt" 2>&1; tail -4 "$OUT/tests_full.txt"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s', d['ms_per_step'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run mask_blocked
  BFLOW_MASK_NCHW=1 run mask_nchw_copy
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_frame.py" "$f" tail | tee "$OUT/tail_launches.txt"
python "$REPO/tools/trace_frame.py" "$f" encoder | tail -12 | tee "$OUT/encoder_end_launches.txt"

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_pair"; mkdir -p "$OUT"; rm -f "$OUT/ab.txt"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q -k "pair or rider or update_block or e2e or smoke" > "$OUT/tests_full.txt" 2>&1; tail -5 "$OUT/tests_full.txt" | tee "$OUT/tests.txt"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s', d['ms_per_step'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run one_queue_pairs
  BFLOW_CONV_PAIR10=1 run one_queue_pairs_10x16
  BFLOW_NO_ONE_QUEUE=1 run side_stream
done
rm -rf /tmp/kt; BFLOW_CONV_PAIR10=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_iteration.py" "$f" | tee "$OUT/pair10_iteration_launches.txt"

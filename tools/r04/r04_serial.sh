set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_serial"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'iter', d.get('ms_per_gru_iter'))" | tee -a "$OUT/ab.txt"; }
for i in 1 2; do
  run side_stream
  BFLOW_NO_OVERLAP=1 run one_queue
done
rm -rf /tmp/kt; BFLOW_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_iteration.py" "$f" | tee "$OUT/serial_iteration_launches.txt"

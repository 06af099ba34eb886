set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_base"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_window.py" "$f" 0 400 > "$OUT/timeline.txt" 2>&1
python "$REPO/tools/stamp_timeline.py" 2>/dev/null | grep " us " > "$OUT/stamps.txt"
tail -3 "$OUT/bench.json" | cut -c1-600

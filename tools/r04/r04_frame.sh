set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_frame"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c2 -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/trace_frame.py" "$f" encoder | tee "$OUT/encoder_launches.txt"
python "$REPO/tools/trace_frame.py" "$f" tail | tee "$OUT/tail_launches.txt"

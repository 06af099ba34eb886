#!/bin/bash
# Kernel timeline of one hipGraph forward (tools only): rocprofv3 kernel trace of tools/profile_forward.py --graph -> gpurun_out/<name>.txt
set -e
NAME=${1:-timeline}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python tools/profile_forward.py --graph --reps 3 ${TL_ARGS:-} > /tmp/tl/run.log 2>&1 || { tail -20 /tmp/tl/run.log; exit 1; }
tail -1 /tmp/tl/run.log
CSV=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py "$CSV" 0 400 > gpurun_out/$NAME.txt
wc -l gpurun_out/$NAME.txt

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_k5bal"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "corr" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  for bal in 1 0; do
    echo "--- BFLOW_CORR_BALANCE=$bal" | tee -a "$OUT/modes.txt"
    BFLOW_CORR_BALANCE=$bal python "$REPO/tools/k5_modes_probe.py" --modes split,split8,f16/w 2>/dev/null | grep -E "^C2" | tee -a "$OUT/modes.txt"
  done
done
for bal in 1 0; do
  echo "--- stamps BFLOW_CORR_BALANCE=$bal" | tee -a "$OUT/stamps.txt"
  BFLOW_CORR_BALANCE=$bal BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_stamps.so" python "$REPO/tools/k5_probe.py" --time-only --stamps --stamp-mode split8 2>/dev/null | sed -n '/stamped launch/,$p' | tee -a "$OUT/stamps.txt"
done
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s, fixed', d['ms_fixed_part'], 'K5', d['roofline_corr_build']['avg_launch_ms'], d['roofline_corr_build']['frac'], 'split', d['roofline_corr_build_split']['avg_launch_ms'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2; do
  BFLOW_CORR_BALANCE=1 run balanced
  BFLOW_CORR_BALANCE=0 run lockstep
done

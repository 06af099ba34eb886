set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_mask"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s', d['ms_per_step'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run tile64
  BFLOW_MASK_TILE=96 run tile96
  BFLOW_MASK_TILE=128 run tile128
done

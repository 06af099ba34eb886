set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_thinmax"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s, c4', d['c4_strong']['value'], d['c4_strong']['ms_per_step'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run default_20000
  BFLOW_THIN_HEAD_MAX_PIXELS=50000 run thin_up_to_50000
done

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_ht"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "half_tile or encoder_product or e2e_forward_vs_reference or norm_in" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms/iter, fixed', d['ms_fixed_part'], 'c4', d['c4_strong']['value'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run half_tile
  BFLOW_CONV_NO_HALF_TILE=1 run full_tile
done

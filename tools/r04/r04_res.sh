set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_res"; mkdir -p "$OUT"
cd "$REPO"
BFLOW_FUSE_RESIDUAL=1 timeout 1500 python -m pytest tests -m gpu -x -q -k "e2e or encoder_product or update_block or validation_step" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s', d['ms_per_step'], 'c4', d['c4_strong']['value'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run default
  BFLOW_FUSE_RESIDUAL=1 run residual_in_conv2_epilogue
done

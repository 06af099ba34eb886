set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_occ"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s, fixed', d['ms_fixed_part'], 'layer-1 conv', d['roofline_encoder']['avg_launch_ms'], 'ms')" | tee -a "$OUT/ab.txt"; }
for i in 1 2; do
  run two_wgs_per_cu
  BFLOW_HALO_LDS_PAD=20000 run one_wg_per_cu
done

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_nt2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms/iter, fixed', d['ms_fixed_part'], 'c4', d['c4_strong']['value'], 'lookup', d['roofline_lookup']['avg_launch_ms'], 'enc', d['roofline_encoder']['avg_launch_ms'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2 3; do
  run default
  BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_ntenc.so" run nt_encoder
done

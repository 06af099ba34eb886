set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_h12"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "halo12 or halo10 or update_block_step or e2e_forward_vs_reference or freeze_bn" > "$OUT/pytest.txt" 2>&1
tail -6 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
echo "--- probe default (halo12 auto)"; python "$REPO/tools/gru_conv_probe.py" 2>/dev/null | grep Cin | tee "$OUT/probe_h12.txt"
echo "--- probe BFLOW_CONV_NO_HALO12=1"; BFLOW_CONV_NO_HALO12=1 python "$REPO/tools/gru_conv_probe.py" 2>/dev/null | grep Cin | tee "$OUT/probe_h8.txt"
for i in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export BFLOW_CONV_NO_HALO12=1; else unset BFLOW_CONV_NO_HALO12; fi
    python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_halo12=$v', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms/iter, fixed', d['ms_fixed_part'], 'c4', d['c4_strong']['value'])" | tee -a "$OUT/ab.txt"
  done
done

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_thin"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "conv_thin or halo12 or update_block_step or e2e_forward_vs_reference" > "$OUT/pytest.txt" 2>&1
tail -6 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
python "$REPO/tools/thin_probe.py" 2>/dev/null | tee "$OUT/thin_probe.txt"
echo "--- gru probe default"; python "$REPO/tools/gru_conv_probe.py" 2>/dev/null | grep Cin | tee "$OUT/probe_default.txt"
echo "--- gru probe spread DMA"; BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_spread.so" python "$REPO/tools/gru_conv_probe.py" 2>/dev/null | grep Cin | tee "$OUT/probe_spread.txt"
run() { python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms/iter, fixed', d['ms_fixed_part'], 'c4', d['c4_strong']['value'])" | tee -a "$OUT/ab.txt"; }
for i in 1 2; do
  run default
  BFLOW_NO_THIN_MFMA=1 run no_thin_mfma
  BFLOW_HIP_LIB="$REPO/bflow_amd/lib/ab/libbflow_hip_spread.so" run spread_dma
done

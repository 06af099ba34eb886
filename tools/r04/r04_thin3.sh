set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_thin3"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q -k "conv_thin or update_block_step or e2e_forward_vs_reference or baseline_configs_full_size" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
python "$REPO/tools/thin_probe.py" 2>/dev/null | tee "$OUT/thin_probe.txt"
for i in 1 2; do
python "$REPO/tools/c5_check.py" --no-oracle 2>/dev/null | grep "C5 GPU" | sed 's/^/thin on matrix cores: /' | tee -a "$OUT/c5.txt"
BFLOW_NO_THIN_MFMA=1 python "$REPO/tools/c5_check.py" --no-oracle 2>/dev/null | grep "C5 GPU" | sed 's/^/thin on the vector ALU: /' | tee -a "$OUT/c5.txt"
done
python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], 'frames/s,', d['ms_per_gru_iter'], 'ms/iter, fixed', d['ms_fixed_part'])"

cd /root/repo
for m in 1 2; do
BFLOW_GRU_SPLIT=$m timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "e2e_forward_vs_reference_golden or update_block or e2e_full_size_dsec" 2>&1 | tail -4
done
for i in 1 2; do
for m in 0 1 2; do
BFLOW_GRU_SPLIT=$m timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $m', d['value'], d['ms_per_step'], d.get('ms_per_gru_iter'), d.get('ms_fixed_part'))"
done; done

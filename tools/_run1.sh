cd /root/repo
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv or update_block or e2e or gru" 2>&1 | tail -4
python tools/gru_conv_probe.py 2>&1 | grep Cin
BFLOW_HIP_LIB=$PWD/bflow_amd/lib/ab/libbflow_hip_h8stamps.so python tools/gru_conv_probe.py --stamps 2>&1 | grep -v "amdgpu.ids\|Cin="
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('ms_per_gru_iter'), d.get('ms_fixed_part'), d.get('epe_vs_oracle'), d.get('parity'))"
done

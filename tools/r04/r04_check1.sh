set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_check1"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q -k "split8 or no_silent or freeze_bn or n2_path or correlation_precisions or tiled_volume or alternating or one_rank" > "$OUT/pytest.txt" 2>&1
tail -5 "$OUT/pytest.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 python "$REPO/bench.py" --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 1500 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ['value','ms_per_step','ms_per_gru_iter','ms_fixed_part']: print(k,d.get(k))
for k in d:
    if k.startswith('roofline'): print(k, {kk:vv for kk,vv in d[k].items() if kk in ('avg_launch_ms','frac','achieved','frac_mfma','line_bytes','frac_of_line_granular_cap','store_ceilings','frac_of_store_ceiling')})
print('value_split', d.get('value_split')); print('c4', d['c4_strong']['value'], 'two', d['c2_two_in_flight']['value'])
PY

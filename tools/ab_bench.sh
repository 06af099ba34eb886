#!/bin/bash
# Alternating same-box A/B of the bench's frame numbers under two environments (tools only).
#   usage (on the GPU box): tools/ab_bench.sh "<env A>" "<env B>" [rounds]     e.g.  tools/ab_bench.sh "BFLOW_CONV_STREAM=0" "" 3
cd "$(dirname "$0")/.."
A="$1"; B="$2"; N="${3:-3}"
for i in $(seq 1 $N); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-only 2>/dev/null | tail -1)
    echo "$tag [$E] $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('value', d['value'], 'ms', d['ms_per_step'], 'c4_strong', (d.get('c4_strong') or {}).get('value'), 'ms/iter', d.get('ms_per_gru_iter'), 'fixed', d.get('ms_fixed_part'))" "$out")"
  done
done

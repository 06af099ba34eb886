#!/bin/bash
# Alternating same-box A/B of the look-up kernel (tools/k7_probe.py, in-graph times) between two libraries (tools only).
#   usage (on the GPU box): tools/k7_ab.sh <libA.so> <libB.so> [rounds] [shapes]
cd "$(dirname "$0")/.."
A="$1"; B="$2"; N="${3:-3}"; SH="${4:-c2,c4,c3,c5}"
for i in $(seq 1 $N); do
  for tag in A B; do
    if [ $tag = A ]; then L="$A"; else L="$B"; fi
    echo "== $tag $L"
    BFLOW_HIP_ABI_ANY=1 BFLOW_HIP_LIB="$L" python tools/k7_probe.py --shapes $SH 2>/dev/null | grep -E "tiled  |tiled-f16|max"
  done
done

#!/bin/bash
# Ablated builds of the stem kernel (conv_split.hip with -DSTEM_ABL=<bits>) and their timing (tools only): build "<bits...>" | run "<bits...>"
# STEMP=1 tools/stem_ablate.sh build "...": the bits go to the PERSISTENT kernel (-DSTEMP_ABL) instead
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
MODE="$1"; BITS="$2"
mkdir -p "$ROOT/bflow_amd/lib/ab"
if [ "$MODE" = build ]; then
  for b in $BITS; do
    if [ -n "${STEMP:-}" ]; then DEFS="-DSTEM_ABL=0 -DSTEMP_ABL=$b"; else DEFS="-DSTEM_ABL=$b"; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics $DEFS -c "$ROOT/bflow_amd/csrc/conv_split.hip" -o "/tmp/stem_abl_$b.o" &
  done
  wait
  for b in $BITS; do
    OBJS=$(ls "$ROOT"/bflow_amd/lib/*.o | grep -v conv_split.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_sabl$b.so" $OBJS "/tmp/stem_abl_$b.o"
  done
else
  for b in $BITS; do
    echo "STEM_ABL=$b:"; BFLOW_HIP_LIB="$ROOT/bflow_amd/lib/ab/libbflow_hip_sabl$b.so" python "$ROOT/tools/stem_probe.py" 2>/dev/null | grep -E "n=(5|40) cin=5" | grep -v MIOpen
  done
fi

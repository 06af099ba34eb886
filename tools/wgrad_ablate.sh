#!/bin/bash
# Ablated builds of the weight-gradient kernel (wgrad_halo.hip with -DWG_ABL=<bits>) and their timing (tools only): build "<bits...>" | run "<bits...>"
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
MODE="$1"; BITS="$2"
mkdir -p "$ROOT/bflow_amd/lib/ab"
if [ "$MODE" = build ]; then
  for b in $BITS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -DWG_ABL=$b -c "$ROOT/bflow_amd/csrc/wgrad_halo.hip" -o "/tmp/wg_abl_$b.o" &
  done
  wait
  for b in $BITS; do
    OBJS=$(ls "$ROOT"/bflow_amd/lib/*.o | grep -v wgrad_halo.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_wabl$b.so" $OBJS "/tmp/wg_abl_$b.o"
  done
else
  for b in $BITS; do
    echo "WG_ABL=$b:"; BFLOW_HIP_LIB="$ROOT/bflow_amd/lib/ab/libbflow_hip_wabl$b.so" python "$ROOT/tools/wgrad_probe.py" 2>/dev/null | cut -c1-70
  done
fi

#!/bin/bash
# Builds libbflow_hip.so (gfx950 only) in-tree: bflow_amd/lib/libbflow_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
OBJS=()
for f in "$HERE"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/../../include/bflow_hip.h" -nt "$o" ]; then
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
  fi
  OBJS+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbflow_hip.so" "${OBJS[@]}"
echo "built $OUT/libbflow_hip.so"

#!/bin/bash
# Builds libbflow_hip.so (gfx950 only) in-tree: bflow_amd/lib/libbflow_hip.so
#   build.sh          incremental (objects older than their source, any header of csrc/ or the ABI header are rebuilt)
#   build.sh --force  rebuild every object (what __graft_entry__.build() runs: a real "does it build" check)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
FORCE=0; [ "${1:-}" = "--force" ] && FORCE=1
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
OBJS=(); PIDS=()
for f in "$HERE"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  stale=0
  for h in "$HERE"/*.h "$HERE/../../include/bflow_hip.h"; do [ "$h" -nt "$o" ] && stale=1; done
  if [ $FORCE = 1 ] || [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    rm -f "$o"
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
    PIDS+=($!)
  fi
  OBJS+=("$o")
done
for p in "${PIDS[@]:-}"; do [ -z "$p" ] || wait "$p" || { echo "build.sh: a compilation failed" >&2; exit 1; }; done
rm -f "$OUT/libbflow_hip.so"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbflow_hip.so" "${OBJS[@]}"
echo "built $OUT/libbflow_hip.so (${#PIDS[@]} object(s) compiled)"

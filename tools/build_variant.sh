#!/bin/bash
# Builds bflow_amd/lib/ab/libbflow_hip_<name>.so from the csrc/ of a git revision (default HEAD), for A/B timing with
# BFLOW_HIP_LIB=... (tools only).   usage: tools/build_variant.sh <name> [rev]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; REV="${2:-HEAD}"
TMP="$(mktemp -d)"
mkdir -p "$TMP/bflow_amd/csrc" "$TMP/include" "$ROOT/bflow_amd/lib/ab"
for f in $(git -C "$ROOT" ls-tree --name-only "$REV" bflow_amd/csrc/ include/); do git -C "$ROOT" show "$REV:$f" > "$TMP/$f"; done
OBJS=()
for f in "$TMP"/bflow_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -c "$f" -o "${f%.hip}.o" &
  OBJS+=("${f%.hip}.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_$NAME.so" "${OBJS[@]}"
rm -rf "$TMP"
echo "built bflow_amd/lib/ab/libbflow_hip_$NAME.so from $REV"

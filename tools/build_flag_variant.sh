#!/bin/bash
# Builds bflow_amd/lib/ab/libbflow_hip_<name>.so from the WORKING TREE with extra compiler flags on the listed sources (the other objects are the
# ones of the normal build), for A/B timing with BFLOW_HIP_LIB=... (tools only).
#   usage: tools/build_flag_variant.sh <name> "<flags>" <source.hip> [<source.hip> ...]      e.g.  spread "-DCONV_SPREAD_DMA=1" conv_split.hip
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; FLAGS="$2"; shift 2
bash "$ROOT/bflow_amd/csrc/build.sh" > /dev/null
TMP="$(mktemp -d)"; mkdir -p "$ROOT/bflow_amd/lib/ab"
OBJS=()
for o in "$ROOT"/bflow_amd/lib/*.o; do
  b="$(basename "${o%.o}").hip"; skip=0
  for f in "$@"; do [ "$f" = "$b" ] && skip=1; done
  [ $skip = 1 ] || OBJS+=("$o")
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics $FLAGS -c "$ROOT/bflow_amd/csrc/$f" -o "$TMP/${f%.hip}.o" &
  OBJS+=("$TMP/${f%.hip}.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_$NAME.so" "${OBJS[@]}"
rm -rf "$TMP"
echo "built bflow_amd/lib/ab/libbflow_hip_$NAME.so ($FLAGS on $*)"

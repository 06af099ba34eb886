#!/bin/bash
# Compiles one HIP source of the library with -save-temps into /tmp/isa and prints registers / spills of the kernels matching a pattern (tools only).
#   usage: tools/kernel_isa.sh conv_split.hip conv_halo_stream ["-DFLAG=1"]   -> /tmp/isa/<source>-hip-amdgcn-amd-amdhsa-gfx950.s
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics ${3:-} -c "$ROOT/bflow_amd/csrc/$1" -o "/tmp/isa/${1%.hip}.o" -save-temps=obj 2>&1 | grep -v "argument unused" || true
S="/tmp/isa/${1%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s"
grep -n "\.name:.*$2" "$S" | while IFS=: read -r ln rest; do
  echo "$rest"; sed -n "$((ln+1)),$((ln+14))p" "$S" | grep -E "vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|sgpr_count" | tr '\n' ' '; echo
done

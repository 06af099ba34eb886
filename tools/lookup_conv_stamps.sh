#!/bin/bash
# Stamped build of the fused look-up + convc1 kernel (tools only): bflow_amd/lib/ab/libbflow_hip_lcstamps.so
#   run with BFLOW_HIP_LIB=<that .so> python tools/lookup_conv_probe.py --stamps
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$ROOT/bflow_amd/lib/ab"
OTHERS=$(ls "$ROOT"/bflow_amd/lib/*.o | grep -v lookup_conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -DLC_STAMPS ${LC_EXTRA_FLAGS:-} \
   -c "$ROOT/bflow_amd/csrc/lookup_conv.hip" -o /tmp/lc_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_lcstamps.so" $OTHERS /tmp/lc_stamps.o
echo "built bflow_amd/lib/ab/libbflow_hip_lcstamps.so"

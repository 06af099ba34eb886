#!/bin/bash
# Stamped build of the small-grid halo convolution (tools only): bflow_amd/lib/ab/libbflow_hip_h8stamps.so
#   run with BFLOW_HIP_LIB=<that .so> python tools/gru_conv_probe.py --stamps
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$ROOT/bflow_amd/lib/ab"
OTHERS=$(ls "$ROOT"/bflow_amd/lib/*.o | grep -v conv_split.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -DH8_STAMPS ${H8_EXTRA_FLAGS:-} \
   -c "$ROOT/bflow_amd/csrc/conv_split.hip" -o /tmp/h8_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_h8stamps.so" $OTHERS /tmp/h8_stamps.o
echo "built bflow_amd/lib/ab/libbflow_hip_h8stamps.so"

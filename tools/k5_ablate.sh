#!/bin/bash
# Timing-only variant builds of the streaming K5 kernel (tools only; ablations 1-5 give WRONG results by construction):
#   tools/k5_ablate.sh name1:"-DSTREAM_ABL=1" name2:"-DSTREAM_STORE_AUX=2" ...  ->  bflow_amd/lib/ab/libbflow_hip_<name>.so
#   run with BFLOW_HIP_LIB=<that .so> python tools/k5_probe.py [--time-only]
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$ROOT/bflow_amd/lib/ab"
OTHERS=$(ls "$ROOT"/bflow_amd/lib/*.o | grep -v corr_stream.o)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics $flags \
     -c "$ROOT/bflow_amd/csrc/corr_stream.hip" -o "/tmp/cs_$name.o" &
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/bflow_amd/lib/ab/libbflow_hip_$name.so" $OTHERS "/tmp/cs_$name.o"
done

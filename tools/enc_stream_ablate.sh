#!/bin/bash
# Anatomy of conv_halo_stream_kernel's steady state (tools only; the CSTREAM_ABL builds compute WRONG results): the layer-1 launch
# (64 -> 64, 5 x 240 x 320) and its batch-40 form under each ablation.   usage (on the GPU box): tools/enc_stream_ablate.sh
cd "$(dirname "$0")/.."
for v in "" 1 2 3 4 5; do
  lib="bflow_amd/lib/libbflow_hip.so"; name="full"
  if [ -n "$v" ]; then lib="bflow_amd/lib/ab/libbflow_hip_cs_abl$v.so"; name=$(echo "no-stores no-dma no-mfma no-frag-reads no-barrier" | cut -d' ' -f$v); fi
  [ -f "$lib" ] || continue
  echo "== $name"
  BFLOW_HIP_LIB="$PWD/$lib" ENC_PROBE_ONLY=0,3 python tools/enc_stream_probe.py 2>&1 | grep "3x3"
done

cd /root/repo
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lookup_conv or fused_lookup" -s 2>&1 | tail -8

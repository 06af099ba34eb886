cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -x -q -k "lookup or corr_ or golden" 2>&1 | tail -3
for abl in 0 32 1 2 4 8 3 7 15; do echo "ABL=$abl"; BFLOW_LOOKUP_ABL=$abl python tools/k7_probe.py --shapes c2,c4 2>/dev/null | grep -E "tiled  "; done

#!/bin/bash
# K7: pixels per workgroup x phase-C form (tools only):  tools/k7_tp.sh [shapes]
cd "$(dirname "$0")/.."
for tp in 2 4 8; do for cols in 0 1; do echo "TP=$tp COLS=$cols"; BFLOW_LOOKUP_TP=$tp BFLOW_LOOKUP_COLS=$cols python tools/k7_probe.py --shapes ${1:-c2,c4} 2>/dev/null | grep -E "tiled  "; done; done

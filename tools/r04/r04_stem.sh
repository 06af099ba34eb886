set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_stem"; mkdir -p "$OUT"
cd "$REPO"
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1
tail -8 "$OUT/pytest.txt"

set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$REPO/gpurun_out/r04_fullgpu"; mkdir -p "$OUT"
cd "$REPO"
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > "$OUT/pytest.txt" 2>&1
tail -25 "$OUT/pytest.txt"

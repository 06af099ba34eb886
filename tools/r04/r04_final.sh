set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
cd "$REPO"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/collect_profiles.sh 2>&1 | tail -2

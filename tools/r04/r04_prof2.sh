set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$PWD}"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "stem or image_configs or e2e_forward_vs_reference" 2>&1 | tail -3
bash tools/collect_profiles.sh 2>&1 | tail -3

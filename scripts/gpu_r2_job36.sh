export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_loopback.py -m gpu -q -x 2>&1 | grep -E "^E|passed|failed|FAILED|stuck" | head -20

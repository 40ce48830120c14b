export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | grep -E "^E|^tests.*Error|passed|failed|FAILED" | head -20

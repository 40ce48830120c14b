export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x -k repl 2>&1 | grep -E "^E|Error|passed|failed" | head -20

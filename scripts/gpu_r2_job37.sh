export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_loopback.py -m gpu -v -x 2>&1 | grep -v "^$" | grep -E "PASS|FAIL|Fatal|File|Thread|Current" | head -60

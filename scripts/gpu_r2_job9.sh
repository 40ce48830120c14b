export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -12 ) 2>&1

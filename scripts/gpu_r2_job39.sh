export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "eight_ranks or eigensolve_eight" --durations=5 2>&1 | tail -5 ) 2>&1 | grep -v "^$"

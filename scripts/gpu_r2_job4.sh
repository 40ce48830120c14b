export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x --durations=0 2>&1 | tail -40 ) 2>&1

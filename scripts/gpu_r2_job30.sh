export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_c_example.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "c_caller or complex_vectors" 2>&1 | tail -8

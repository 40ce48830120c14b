# GPU job 35: diagonalize_distributed on one rank (block-distributed output)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_diagonalize.py -q -x > $OUT/pytest_job35.log 2>&1; tail -15 $OUT/pytest_job35.log

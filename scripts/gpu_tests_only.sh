# GPU job: the whole -m gpu suite and the smoke entry on the current tree
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_last.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu_last.log | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2

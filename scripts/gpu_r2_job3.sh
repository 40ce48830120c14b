# GPU job r2/3: native RCCL path (single rank) + the whole -m gpu suite after the per-rank-plan family change
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | tail -15
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --deselect tests/test_gpu_rccl.py 2>&1 | tail -15 ) 2>&1

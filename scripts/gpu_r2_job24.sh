export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "stage_timing" 2>&1 | tail -5
python bench.py --steps 5 --warmup 2 --force-distributed --model heisenberg_chain_28 --no-cpu-baseline --kDisplayTimings 2>&1 | grep -v "^{" | head -30

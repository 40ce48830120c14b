# GPU job 2 of round 3: the whole -m gpu suite on the indexed default; PMC passes of the indexed pull kernel on chain_36_symm.
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu_full.log 2>&1; tail -15 $OUT/pytest_gpu_full.log
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r3/pmc_36symm bash scripts/gpu_pmc_traffic.sh > $OUT/pmc_36symm.log 2>&1; tail -12 $OUT/pmc_36symm.log

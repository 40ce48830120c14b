# GPU job 33: eight ranks over the loop-back transport on the final kernels (replicated-x, indexed), chain_36_symm and chain_40_symm
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 600 python scripts/loopback_bench.py --L 36 --symm --P 8 --mode replicated --steps 5 > $OUT/loopback_36symm_final.txt 2>&1; tail -12 $OUT/loopback_36symm_final.txt
timeout 1200 python scripts/loopback_bench.py --L 40 --symm --P 8 --mode replicated --steps 3 > $OUT/loopback_40symm_final.txt 2>&1; tail -12 $OUT/loopback_40symm_final.txt

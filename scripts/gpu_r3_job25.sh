# GPU job 25: K4 alone (scripts/k4_rate.py) on the packets of chain_36_symm, doubling build and step-by-step build
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 600 python scripts/k4_rate.py heisenberg_chain_36_symm 2>&1 | grep variant | tee $OUT/k4_rate.txt

# GPU job 4 of round 3: occupancy / split sweep of the sibling-tile kernel, its fabric traffic (PMC), parity of the variants
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --maxfail=5 -k "row_kernel_variants" > $OUT/pytest_sib2.log 2>&1; tail -4 $OUT/pytest_sib2.log
C=""
for th in 1024 768 512; do for t in 5 4; do C="$C;LS_AMD_SIB_THREADS=$th,LS_AMD_SIB_T=$t"; done; done
C="$C;LS_AMD_SIB_T=4,LS_AMD_SIB_NL=11;LS_AMD_SIB_T=5,LS_AMD_SIB_NL=11;LS_AMD_SIB_T=6,LS_AMD_SIB_NL=11;LS_AMD_SIB_T=6,LS_AMD_SIB_NL=10;LS_AMD_SIB_T=5,LS_AMD_SIB_NL=10;LS_AMD_SIB_T=3;LS_AMD_SIB=0;"
timeout 600 python scripts/order_sweep.py --steps 8 --configs "$C" > $OUT/sib_sweep2.log 2>&1; cat $OUT/sib_sweep2.log
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r3/pmc_chain32_sib PASSES=min bash scripts/gpu_pmc_traffic.sh > $OUT/pmc_chain32_sib.log 2>&1; tail -25 $OUT/pmc_chain32_sib.log | cut -c1-180

# GPU job 6 of round 3: sibling-tile kernel v3 (launch records, one barrier, one row per lane): sweep, then unit counters of
# the sibling and the staged kernel side by side
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --maxfail=5 -k "row_kernel_variants" > $OUT/pytest_sib4.log 2>&1; tail -3 $OUT/pytest_sib4.log
C=""
for th in 768 512 1024; do for t in 5 4; do C="$C;LS_AMD_SIB_THREADS=$th,LS_AMD_SIB_T=$t"; done; done
C="$C;LS_AMD_SIB_T=3,LS_AMD_SIB_THREADS=256;LS_AMD_SIB_T=3,LS_AMD_SIB_THREADS=512;LS_AMD_SIB_T=2,LS_AMD_SIB_THREADS=256;LS_AMD_SIB_T=1,LS_AMD_SIB_THREADS=256;LS_AMD_SIB_T=4,LS_AMD_SIB_THREADS=384;LS_AMD_SIB=0;"
timeout 600 python scripts/order_sweep.py --steps 8 --configs "$C" > $OUT/sib_sweep4.log 2>&1; cat $OUT/sib_sweep4.log
LS_AMD_SIB_T=4 LS_AMD_SIB_THREADS=512 bash scripts/gpu_pmc_units.sh r3_sib > $OUT/units_sib.log 2>&1
LS_AMD_SIB=0 bash scripts/gpu_pmc_units.sh r3_staged > $OUT/units_staged.log 2>&1
cp gpurun_out/pmc_r3_sib/summary.txt $OUT/units_sib_summary.txt; cp gpurun_out/pmc_r3_staged/summary.txt $OUT/units_staged_summary.txt

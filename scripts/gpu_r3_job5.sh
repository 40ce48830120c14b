# GPU job 5 of round 3: sibling-tile kernel v2 (launch records, one barrier, two rows per lane)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --maxfail=5 -k "row_kernel_variants" > $OUT/pytest_sib3.log 2>&1; tail -4 $OUT/pytest_sib3.log
C=""
for th in 768 512 1024 640; do for t in 5 4; do C="$C;LS_AMD_SIB_THREADS=$th,LS_AMD_SIB_T=$t"; done; done
C="$C;LS_AMD_SIB_T=5,LS_AMD_SIB_NL=11,LS_AMD_SIB_THREADS=512;LS_AMD_SIB_T=3,LS_AMD_SIB_THREADS=512;LS_AMD_SIB_T=3,LS_AMD_SIB_THREADS=256;LS_AMD_SIB_CHUNK=8;LS_AMD_SIB_CHUNK=256;LS_AMD_SIB=0;"
timeout 600 python scripts/order_sweep.py --steps 8 --configs "$C" > $OUT/sib_sweep3.log 2>&1; cat $OUT/sib_sweep3.log

# GPU job 12: sibling-tile kernel with K units per block and register-staged prefetch of the next window
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --maxfail=5 -k "row_kernel_variants" > $OUT/pytest_sib5.log 2>&1; tail -4 $OUT/pytest_sib5.log
S="LS_AMD_SIB=1"
C="$S,LS_AMD_SIB_K=1;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=1,LS_AMD_SIB_THREADS=512;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=4;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=8;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=2;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=16;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=4,LS_AMD_SIB_THREADS=1024;$S,LS_AMD_SIB_T=5,LS_AMD_SIB_K=4,LS_AMD_SIB_THREADS=1024;$S,LS_AMD_SIB_T=3,LS_AMD_SIB_K=4,LS_AMD_SIB_THREADS=512;$S,LS_AMD_SIB_T=3,LS_AMD_SIB_K=8,LS_AMD_SIB_THREADS=384;$S,LS_AMD_SIB_T=4,LS_AMD_SIB_K=4,LS_AMD_SIB_CHUNK=64;LS_AMD_SIB=0"
timeout 600 python scripts/order_sweep.py --steps 8 --configs "$C" > $OUT/sib_sweep6_prefetch.log 2>&1; cut -c1-220 $OUT/sib_sweep6_prefetch.log

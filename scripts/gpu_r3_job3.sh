# GPU job 3 of round 3: the block-aligned sibling-tile row kernel -- parity, then the split / dealing sweep on chain_32.
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 -k "row_kernel_variants or staged_kernel_instantiations or kernel_table or golden_vectors or single_locale_matvec_f64" > $OUT/pytest_sib.log 2>&1; tail -12 $OUT/pytest_sib.log
timeout 600 python scripts/order_sweep.py --steps 8 --configs ";LS_AMD_SIB=0;LS_AMD_SIB_T=4;LS_AMD_SIB_T=6,LS_AMD_SIB_NL=11;LS_AMD_SIB_NL=11;LS_AMD_SIB_NL=13,LS_AMD_SIB_T=4;LS_AMD_SIB_CHUNK=8;LS_AMD_SIB_CHUNK=128;LS_AMD_SIB_CHUNK=1024;LS_AMD_SIB_T=3;LS_AMD_SIB_T=2;" > $OUT/sib_sweep1.log 2>&1; cat $OUT/sib_sweep1.log
timeout 600 python -m pytest tests -m gpu -q -x -k "edge_targeted" > $OUT/pytest_sib_edge.log 2>&1; tail -5 $OUT/pytest_sib_edge.log

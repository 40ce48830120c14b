# GPU job (next experiment): does the set order of the tile map give the staged kernel
# L2 hits on its far-pair gathers?  Parity first, then time and fabric traffic against the default order.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=4 LS_AMD_SET_ROWS=8192 timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x \
  -k "row_kernel_variants and default or single_locale or chain_24" 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
echo "+ default order"; $B
for t in 6 8 10; do
  for r in 65536 131072 262144; do
    echo "+ TRANSPOSED t=$t rows=$r"; LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=$t LS_AMD_SET_ROWS=$r $B
  done
done
# traffic of the best candidate (edit t / rows): FETCH_SIZE and the TCC hit rate, one counter group per pass
export CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=8 LS_AMD_SET_ROWS=131072 bash scripts/gpu_pmc_quick.sh transposed_chain

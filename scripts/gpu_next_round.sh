# GPU job (next measurement): the staged kernel is at the bound of its fabric traffic.
# The hierarchy model (scripts/tools/mallsim.c) predicts -12 % HBM reads and +3 % L2 misses when tiles are dealt to the
# XCDs in round-robin chunks (LS_AMD_TILE_CHUNK) instead of contiguous eighths; the set order (LS_AMD_TRANSPOSED) is
# predicted to give nothing (scripts/tools/l2sim.c).  Parity first, then time, then FETCH_SIZE of the best chunk.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LS_AMD_TILE_CHUNK=3 timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x \
  -k "row_kernel_variants or single_locale or chain_24 or block_rows" 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for rep in 1 2; do
  echo "+ contiguous eighths ($rep)"; $B
  for g in 32 128 512 2048; do echo "+ TILE_CHUNK=$g ($rep)"; LS_AMD_TILE_CHUNK=$g $B; done
done
echo "+ TRANSPOSED t=8 rows=262144"; LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=8 LS_AMD_SET_ROWS=262144 $B
# chip-wide sets (transposed order + chunked dealing): modelled -27 % HBM reads and -6 % L2 misses at t=6, 5.2 M rows
# per set, chunk 32; the model is flat (53-57 B/row of HBM reads against 73) over t = 4..8 and 0.65-10 M rows per set
LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=3 LS_AMD_SET_ROWS=16384 LS_AMD_TILE_CHUNK=2 timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x \
  -k "row_kernel_variants and default or single_locale or chain_24" 2>&1 | tail -3
for cfg in "6 5242880 32" "6 2621440 32" "7 10485760 32" "4 2621440 32" "8 4587520 32" "6 5242880 8"; do
  set -- $cfg
  echo "+ chip-wide sets t=$1 rows=$2 chunk=$3"; LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=$1 LS_AMD_SET_ROWS=$2 LS_AMD_TILE_CHUNK=$3 $B
done
export CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
LS_AMD_TILE_CHUNK=512 bash scripts/gpu_pmc_quick.sh chunk512
LS_AMD_TRANSPOSED=1 LS_AMD_TOP_BITS=6 LS_AMD_SET_ROWS=5242880 LS_AMD_TILE_CHUNK=32 bash scripts/gpu_pmc_quick.sh chipsets

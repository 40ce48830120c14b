# GPU job 23: bucket-directory window look-up of k_tile_pull_idx against the binary search (same tree otherwise)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 1200 python -m pytest tests -m gpu -q -x -k "indexed or symm or single_locale or replicated or ranks_as_threads or bethe or kagome or complex_characters or partitioned" > $OUT/pytest_job23.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job23.log | tail -2
for v in dir nodir dir nodir; do
  cp $P/libls_amd_$v.so $P/libls_amd.so
  for h in 128 512; do
    LS_AMD_PULL_HALO=$h timeout 600 python bench.py --model heisenberg_chain_36_symm --steps 8 --warmup 2 --no-cpu-baseline > $OUT/dir_${v}_36_h$h.json 2>/dev/null
    echo "$v chain_36_symm halo=$h: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/dir_${v}_36_h$h.json | head -1)"
  done
done | tee $OUT/dir_ab.txt
for v in dir nodir; do
  cp $P/libls_amd_$v.so $P/libls_amd.so
  LS_AMD_PULL_HALO=128 timeout 600 python bench.py --model heisenberg_chain_40_symm --steps 4 --warmup 1 --no-cpu-baseline > $OUT/dir_${v}_40.json 2>/dev/null
  echo "$v chain_40_symm halo=128: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/dir_${v}_40.json | head -1)"
done | tee -a $OUT/dir_ab.txt
cp $P/libls_amd_dir.so $P/libls_amd.so

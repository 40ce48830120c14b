# GPU job 24: K4 run length by doubling + refinement against the step-by-step loop (same tree otherwise)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 1200 python -m pytest tests -m gpu -q -x -k "indexed or symm or single_locale or replicated or ranks_as_threads or bethe or kagome or complex_characters or partitioned or k4" > $OUT/pytest_job24.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job24.log | tail -2
cp $P/libls_amd.so /tmp/new.so
for v in new old new old; do
  [ $v = old ] && cp $P/libls_amd_oldk4.so $P/libls_amd.so || cp /tmp/new.so $P/libls_amd.so
  timeout 600 python bench.py --model heisenberg_chain_36_symm --steps 8 --warmup 2 --no-cpu-baseline > $OUT/k4d_${v}_36.json 2>/dev/null
  echo "$v chain_36_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/k4d_${v}_36.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/k4d_${v}_36.json | head -1)"
done | tee $OUT/k4_doubling_ab.txt
for v in new old; do
  [ $v = old ] && cp $P/libls_amd_oldk4.so $P/libls_amd.so || cp /tmp/new.so $P/libls_amd.so
  timeout 600 python bench.py --model heisenberg_chain_40_symm --steps 4 --warmup 1 --no-cpu-baseline > $OUT/k4d_${v}_40.json 2>/dev/null
  echo "$v chain_40_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/k4d_${v}_40.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/k4d_${v}_40.json | head -1)"
done | tee -a $OUT/k4_doubling_ab.txt
cp /tmp/new.so $P/libls_amd.so

# GPU job 30: per-wave packet rings (k_tile_pull_wv, LS_AMD_PULL_WAVE=1) against k_tile_pull_idx: parity, then benches
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
LS_AMD_PULL_WAVE=1 timeout 1200 python -m pytest tests -m gpu -q -x -k "indexed or symm or single_locale or replicated or ranks_as_threads or bethe or kagome or complex_characters or partitioned or k4" > $OUT/pytest_job30.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job30.log | tail -2
for v in 0 1 0 1; do
  for m in 36 40; do
    LS_AMD_PULL_WAVE=$v timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/wv_${v}_$m.json 2>/dev/null
    echo "wave=$v chain_${m}_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/wv_${v}_$m.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/wv_${v}_$m.json | head -1)"
  done
done | tee $OUT/wv_ab.txt

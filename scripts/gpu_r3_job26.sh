# GPU job 26: K4 with the mirrored candidates inside the loop over the runs: micro-benchmark, parity subset, benches
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 600 python scripts/k4_rate.py heisenberg_chain_36_symm 2>&1 | grep variant | tee $OUT/k4_rate_v2.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "indexed or symm or single_locale or replicated or ranks_as_threads or bethe or kagome or complex_characters or partitioned or k4" > $OUT/pytest_job26.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job26.log | tail -2
for m in 36 40; do
  timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 6 --warmup 2 --no-cpu-baseline > $OUT/k4v2_$m.json 2>/dev/null
  echo "chain_${m}_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/k4v2_$m.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/k4v2_$m.json | head -1)"
done | tee $OUT/k4v2_bench.txt

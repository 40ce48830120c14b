# GPU job 1 of round 3: new parity tests (indexed pull mode, two operators on one basis, Bethe pins on the small shapes),
# A/B of the indexed mode on one device, 8-rank loop-back timings of chain_36_symm / chain_40_symm, default bench.
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x \
  -k "indexed or two_operators or replicated_exchange or ranks_as_threads or bethe or kernel_table or staged_kernel or single_locale_matvec_f64" \
  > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
for idx in 0 1; do
  LS_AMD_PULL_INDEXED=$idx timeout 600 python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline --kDisplayTimings \
    > $OUT/bench_36symm_idx$idx.json 2> $OUT/bench_36symm_idx$idx.err
  grep -o '"value": [0-9.]*\|"kernel_ms_avg": [0-9.]*' $OUT/bench_36symm_idx$idx.json | head -2; grep -A8 matrixVectorProduct $OUT/bench_36symm_idx$idx.err | head -9
done
for idx in 1 0; do
  LS_AMD_REPL_INDEXED=$idx timeout 900 python scripts/loopback_bench.py --L 36 --symm --P 8 --mode replicated --steps 5 > $OUT/loopback_36symm_repl_idx$idx.txt 2>&1
  tail -12 $OUT/loopback_36symm_repl_idx$idx.txt
done
LS_AMD_REPL_INDEXED=1 timeout 1500 python scripts/loopback_bench.py --L 40 --symm --P 8 --mode replicated --steps 3 > $OUT/loopback_40symm_repl_idx1.txt 2>&1
tail -12 $OUT/loopback_40symm_repl_idx1.txt
for idx in 0 1; do
  LS_AMD_PULL_INDEXED=$idx timeout 900 python bench.py --model heisenberg_chain_40_symm --steps 4 --warmup 2 --no-cpu-baseline --kDisplayTimings \
    > $OUT/bench_40symm_idx$idx.json 2> $OUT/bench_40symm_idx$idx.err
  grep -o '"value": [0-9.]*\|"kernel_ms_avg": [0-9.]*' $OUT/bench_40symm_idx$idx.json | head -2; grep -A8 matrixVectorProduct $OUT/bench_40symm_idx$idx.err | head -9
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json

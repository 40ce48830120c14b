# GPU job 19: k_chain_t with the near-pair table, adaptive uniform split and base+lane far gathers: suite, then A/B benches
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_job19.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job19.log | tail -3
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_job19_f64_$i.json 2>/dev/null; grep -o '"value": [0-9.]*' $OUT/bench_job19_f64_$i.json | head -1; grep -o '"kernel_ms_avg": [0-9.]*' $OUT/bench_job19_f64_$i.json | head -1
done
timeout 600 python bench.py --dtype c128 --no-cpu-baseline --no-extra > $OUT/bench_job19_c128.json 2>$OUT/bench_job19_c128.err; grep -o '"value": [0-9.]*' $OUT/bench_job19_c128.json | head -1; grep -o '"c128[^}]*' $OUT/bench_job19_f64_1.json | head -3

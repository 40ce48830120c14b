# GPU job 37: the VALU / TCC passes of the headline kernel as well (so that the default bench line carries int_alu), default bench
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3final; mkdir -p $OUT
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r3_chain32_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_merge.py r3_chain32_f64
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/r3_chain32_f64_rocprof_summary.txt profiles/r3_chain32_f64_bench_line.json $OUT/
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; grep -o '"value": [0-9.]*' $OUT/bench_default.json | head -1; grep -o '"int_alu": {[^}]*}' $OUT/bench_default.json

# GPU job 7 of round 3: full GPU suite on the current tree; the generic row kernel on non-chain models (+ rocprof stats)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu_full2.log 2>&1; tail -6 $OUT/pytest_gpu_full2.log
timeout 900 python scripts/nonchain_bench.py > $OUT/nonchain.jsonl 2> $OUT/nonchain.err; cat $OUT/nonchain.jsonl; tail -3 $OUT/nonchain.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/nonchain_prof -o trace -- python $ROOT/scripts/nonchain_bench.py --models square_6x5 --steps 4 > $OUT/nonchain_prof.log 2>&1
cd $ROOT && python3 scripts/rocpd_summary.py $OUT/nonchain_prof > $OUT/nonchain_rocprof_summary.txt 2>&1; rm -rf $OUT/nonchain_prof; head -12 $OUT/nonchain_rocprof_summary.txt | cut -c1-150

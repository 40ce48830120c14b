# GPU job: fabric bytes (FETCH_SIZE, WRITE_SIZE) and VALU instruction count of the dominant kernel of a bench workload, each
# counter group in its own rocprofv3 pass (FETCH_SIZE + WRITE_SIZE together exceed the TCC slots), plus the kernel trace.
#   PASSES=min skips the VALU and TCC passes
#   MODEL=heisenberg_chain_32 DTYPE=f64 KNAME=direct-pull+staged TAG=r2_chain32 bash scripts/gpu_pmc_traffic.sh
export TMPDIR=/tmp
MODEL=${MODEL:-heisenberg_chain_32}; DTYPE=${DTYPE:-f64}; TAG=${TAG:-pmc}; MODE=${MODE:-auto}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
CMD="python $ROOT/bench.py --model $MODEL --dtype $DTYPE --mode $MODE --steps 4 --warmup 2 --no-cpu-baseline --no-extra"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
[ "${PASSES:-all}" = all ] && timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_valu -o pmc -- $CMD > $OUT/pmc_valu.log 2>&1
[ "${PASSES:-all}" = all ] && timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
cd $ROOT
KNAME=${KNAME:-$(grep -o '"kernel": "[^"]*"' $OUT/trace.log | head -1 | cut -d'"' -f4)}
PMC_SOURCE="profiles/${TAG}_rocprof_summary.txt" python3 scripts/pmc_traffic_entry.py $OUT $MODEL $DTYPE "$KNAME" > $OUT/pmc_traffic_entry.json
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/trace.log > $OUT/bench_line.json
rm -rf $OUT/*/*.db $OUT/*/*/*.db
cat $OUT/pmc_traffic_entry.json
head -8 $OUT/summary.txt | cut -c1-150

# GPU job: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE / VALU / TCC counter passes (each in a run of its own) of ANY command.
#   CMD="python scripts/tile_bench.py --L 28 --P 8" TAG=r6_packets_ring bash scripts/gpu_pmc_cmd.sh
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${TAG:-pmc_cmd}
mkdir -p $OUT
cd /tmp
FULL="cd $ROOT && $CMD"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- bash -c "$FULL" > $OUT/trace.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- bash -c "$FULL" > $OUT/pmc_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- bash -c "$FULL" > $OUT/pmc_write.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_valu -o pmc -- bash -c "$FULL" > $OUT/pmc_valu.log 2>&1
timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- bash -c "$FULL" > $OUT/pmc_tcc.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $OUT/pmc_sq2 -o pmc -- bash -c "$FULL" > $OUT/pmc_sq2.log 2>&1
cd $ROOT
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db $OUT/*/*/*.db
cat $OUT/summary.txt | cut -c1-160 | head -${LINES_OUT:-70}

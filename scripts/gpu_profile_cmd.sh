# GPU job: rocprofv3 kernel trace + SQ/TCC PMC passes of an arbitrary command ($CMD), summary via rocpd_summary.py
set -x
export TMPDIR=/tmp
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
if [ -n "$WITH_MEM" ]; then
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
fi
python3 $ROOT/scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db
cut -c1-180 $OUT/summary.txt

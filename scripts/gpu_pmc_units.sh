# GPU job: which unit is busy in the row kernel? Small counter groups, one per pass, each under a timeout.
export TMPDIR=/tmp
TAG=${1:-units}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
cd /tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 70 rocprofv3 --pmc $group -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
GROUPS
python3 $ROOT/scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db
grep -E "k_direct|k_lin|k_chain" $OUT/summary.txt | grep -E "SQ_|TA_|TCP_" | cut -c1-20,60-140

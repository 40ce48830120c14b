# GPU job 11 (round 4): what bounds the cached gather (k_pull_gather)?  Counter passes on chain_36_symm with the slot cache
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r4job11; mkdir -p $OUT
CMD="python $ROOT/scripts/tile_bench.py --L 36 --symm --mode pull --cache --steps 8"
$CMD 2>&1 | tail -1 | tee $OUT/cached36.txt
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $group -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
GROUPS
cd $ROOT
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db $OUT/*/*/*.db
grep -E "k_pull_gather" $OUT/summary.txt | cut -c1-50,80-160

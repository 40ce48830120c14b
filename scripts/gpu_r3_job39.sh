# GPU job 39: SQ activity counters of k_tile_pull_wv on chain_36_symm (two passes)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd /tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $group -d $OUT/sqw_p$i -o pmc -- python $ROOT/bench.py --model heisenberg_chain_36_symm --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/sqw_p$i.log 2>&1 || echo "pass $i failed"
done <<'GROUPS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CU_CYCLES
GROUPS
for k in 1 2; do python3 $ROOT/scripts/rocpd_summary.py $OUT/sqw_p$k; done > $OUT/sqw_summary.txt 2>&1
rm -rf $OUT/sqw_p*/
grep -E "k_tile_pull_wv" $OUT/sqw_summary.txt | grep -E "SQ_|GRBM" | cut -c1-30,60-140

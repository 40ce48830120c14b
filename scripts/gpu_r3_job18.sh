# GPU job 18: VALU issue rates (microbenchmark) + SQ activity counters of k_tile_pull_idx on chain_36_symm and k_chain_t on chain_32
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 120 scripts/tools/valu_rate > $OUT/valu_rate.txt 2>&1; cat $OUT/valu_rate.txt
cd /tmp
i=0
for model in heisenberg_chain_36_symm heisenberg_chain_32; do
while read -r group; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $group -d $OUT/sq_p$i -o pmc -- python $ROOT/bench.py --model $model --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/sq_p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CU_CYCLES
GROUPS
done
for k in 1 2 3 4; do python3 $ROOT/scripts/rocpd_summary.py $OUT/sq_p$k; done > $OUT/sq_summary.txt 2>&1
rm -rf $OUT/sq_p*/*.db $OUT/sq_p*/*/*.db
grep -E "k_tile_pull_idx|k_chain_t" $OUT/sq_summary.txt | grep -E "SQ_|GRBM" | cut -c1-30,60-140

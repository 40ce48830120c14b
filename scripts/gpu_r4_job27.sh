# GPU job 27 (round 4): wait share of the packet producer k_tile_wv (VERDICT r3 #5 asks for the SQ-wait share of the producer)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r4job27; mkdir -p $OUT
CMD="python $ROOT/scripts/tile_bench.py --L 28 --P 8 --steps 5"
cd /tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $group -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS
GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
GROUPS
cd $ROOT
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db $OUT/*/*/*.db
grep -E "k_tile_wv<unsigned long, true, false, true, false>|k_scatter<false>" $OUT/summary.txt | cut -c1-20,60-130

# GPU job 7 (round 4): evidence passes -- (1) per-stage timing tree + kernel trace of the packets path (chain_28 x 8 partitions),
# (2) FETCH / WRITE / VALU / TA / TCC counters of the staged pair kernel (k_pairs_t) next to the generic row kernel (k_direct) on
# square_6x5, (3) fresh PMC entry of chain_40_symm (the committed one predates k_pull_t)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r4job7; mkdir -p $OUT
timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 --tree > $OUT/packets_tree_f64.txt 2>&1; tail -14 $OUT/packets_tree_f64.txt
timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 --tree --dtype c128 > $OUT/packets_tree_c128.txt 2>&1; tail -14 $OUT/packets_tree_c128.txt
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/packets_trace -o trace -- python $ROOT/scripts/tile_bench.py --L 28 --P 8 --steps 5 > $OUT/packets_trace.log 2>&1
NC="python $ROOT/scripts/nonchain_bench.py --models square_6x5 --steps 3"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/nc_trace -o trace -- $NC > $OUT/nc_trace.log 2>&1
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $group -d $OUT/nc_p$i -o pmc -- $NC > $OUT/nc_p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
GROUPS
cd $ROOT
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db $OUT/*/*/*.db
grep -E "k_pairs_t|k_direct|k_tile_wv|k_scatter" $OUT/summary.txt | cut -c1-50,80-160 | head -60
MODEL=heisenberg_chain_40_symm DTYPE=f64 TAG=r4_chain40symm_f64 bash scripts/gpu_pmc_traffic.sh > $OUT/pmc40.log 2>&1; tail -12 $OUT/pmc40.log
python scripts/pmc_traffic_merge.py r4_chain40symm_f64 2>&1 | tail -3
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/r4_chain40symm* $OUT/ 2>/dev/null

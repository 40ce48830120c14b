# GPU job 12 (round 4): slot-cache streams sorted by slot -- parity, then gather-only timings sorted / unsorted on chain_36_symm,
# chain_40_symm and square_6x6, and the address-path counters of the sorted gather
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r4job12; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py -m gpu -q -x -k "slot_cache or replicated_exchange" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_focus.log
( time timeout 600 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "bethe" > $OUT/pytest_bethe.log 2>&1 ) 2>&1 | grep real; tail -2 $OUT/pytest_bethe.log
for L in 36 40; do
  for srt in 1 0; do
    echo -n "L=$L sort=$srt: "; LS_AMD_SLOT_CACHE_SORT=$srt timeout 600 python scripts/tile_bench.py --L $L --symm --mode pull --cache --steps 6 2>&1 | tail -1
  done
done | tee $OUT/cached_sort_ab.txt
timeout 300 python scripts/lattice_bench.py heisenberg_square_6x6 5 2>&1 | tail -1 | cut -c1-60,380-700 | tee -a $OUT/cached_sort_ab.txt
CMD="python $ROOT/scripts/tile_bench.py --L 36 --symm --mode pull --cache --steps 8"
cd /tmp
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $group -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($group) failed rc=$?"
done <<'GROUPS'
FETCH_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
GROUPS
cd $ROOT
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db $OUT/*/*/*.db
grep -E "k_pull_gather|k_pull_sort" $OUT/summary.txt | cut -c1-30,60-130

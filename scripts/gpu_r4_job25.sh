# GPU job 25 (round 4): wave-cooperative k_build_table: the tests that use the search index, set-up time of chain_40_symm
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job25; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "partitioned or replicated_x or complex_characters or hashed_layout or state_index or batched" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -2 $OUT/pytest_focus.log
( time timeout 600 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "eight_partitions or eight_ranks" > $OUT/pytest_big.log 2>&1 ) 2>&1 | grep real; tail -2 $OUT/pytest_big.log
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/tile_bench.py --L 40 --symm --mode pull --steps 2 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py $OUT/trace > $OUT/summary.txt 2>&1; rm -rf $OUT/trace/*.db $OUT/trace/*/*.db
grep -E "k_build_table|k_enum_flags|k_gtab_insert|k_norms" $OUT/summary.txt | cut -c1-140

# GPU job 6 (round 4): packet producer with per-wave rings and a plan-fixed send layout (k_tile_wv) against the block-wide lists
# (k_tile, LS_AMD_PACKETS=block): parity suite with the new default, the multi-partition tests on the old path, timings
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job6; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 > $OUT/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -6 $OUT/pytest_gpu.log
( time LS_AMD_PACKETS=block timeout 900 python -m pytest tests/test_gpu_matvec.py -m gpu -q --maxfail=5 > $OUT/pytest_block.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_block.log
for mode in wave block; do
  for args in "--L 28 --P 8" "--L 28 --P 8 --dtype c128" "--L 28 --P 2" "--L 30 --P 8" "--L 32 --symm --P 8 --mode push" "--L 36 --symm --P 8 --mode push"; do
    echo -n "$mode: "; LS_AMD_PACKETS=$mode timeout 300 python scripts/tile_bench.py $args --steps 5 2>&1 | tail -1
  done
done | tee $OUT/packets_ab.txt

# GPU job 28 (round 4): rank directory (closed-form rank + popcount) instead of the binary search in the packet kernels: parity, A/B
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job28; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py tests/test_gpu_dense_pin.py tests/test_gpu_rccl.py -m gpu -q -x -k "not slot_cache" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_focus.log
( time timeout 600 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "eight_partitions or eight_ranks or chain_32" > $OUT/pytest_big.log 2>&1 ) 2>&1 | grep real; tail -2 $OUT/pytest_big.log
for rd in 1 0; do
  for args in "--L 28 --P 8" "--L 28 --P 8 --dtype c128" "--L 28 --P 2" "--L 30 --P 8"; do
    echo -n "rankdir=$rd $args: "; LS_AMD_RANKDIR=$rd timeout 300 python scripts/tile_bench.py $args --steps 5 --tree 2>&1 | grep -E "matvec=|producers|consumers" | tr '\n' ' ' | sed 's/  */ /g' | cut -c1-330; echo
  done
done | tee $OUT/rankdir_ab.txt

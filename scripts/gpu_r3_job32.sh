# GPU job 32: k_scatter with four interleaved look-ups per thread against one: parity of the packet paths, chain_28 x 8 partitions
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 1200 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py tests/test_gpu_rccl.py -m gpu -q -x > $OUT/pytest_job32.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job32.log | tail -2
for v in sc4 sc1 sc4 sc1; do
  cp $P/libls_amd_$v.so $P/libls_amd.so
  echo "$v: $(timeout 600 python scripts/tile_bench.py --L 28 --P 8 --steps 5 --mode push 2>&1 | tail -2 | tr '\n' ' ')"
done | tee $OUT/scatter_u4_ab.txt
cp $P/libls_amd_sc4.so $P/libls_amd.so

# GPU job 29: k_tile_pull_idx with two groups per LDS list (half the LDS and registers) at 6 / 7 / 8 blocks per CU
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for v in base g2 g2o7 g2o8 g4o7; do
  [ $v = base ] && cp /tmp/base.so $P/libls_amd.so || cp $P/libls_amd_$v.so $P/libls_amd.so
  for m in 36 40; do
    timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/gc_${v}_$m.json 2>/dev/null
    echo "$v chain_${m}_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/gc_${v}_$m.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/gc_${v}_$m.json | head -1)"
  done
done | tee $OUT/gc_ab.txt
cp /tmp/base.so $P/libls_amd.so

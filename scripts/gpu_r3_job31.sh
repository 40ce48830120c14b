# GPU job 31: occupancy / ring-size variants of k_tile_pull_wv (halo 128 so that the window does not cap the blocks)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for v in base wa wb wc wd; do
  [ $v = base ] && cp /tmp/base.so $P/libls_amd.so || cp $P/libls_amd_$v.so $P/libls_amd.so
  for m in 36 40; do
    LS_AMD_PULL_HALO=128 timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/wvv_${v}_$m.json 2>/dev/null
    echo "$v chain_${m}_symm halo 128: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/wvv_${v}_$m.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/wvv_${v}_$m.json | head -1)"
  done
done | tee $OUT/wv_variants.txt
cp /tmp/base.so $P/libls_amd.so

# GPU job 27: k_tile_pull_idx (new K4) at 7 blocks per CU, halo 128, against the shipped build
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for v in base occ7 base occ7; do
  [ $v = occ7 ] && cp $P/libls_amd_occ7.so $P/libls_amd.so || cp /tmp/base.so $P/libls_amd.so
  for m in 36 40; do
    LS_AMD_PULL_HALO=128 timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/occ7b_${v}_$m.json 2>/dev/null
    echo "$v chain_${m}_symm halo 128: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/occ7b_${v}_$m.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/occ7b_${v}_$m.json | head -1)"
  done
done | tee $OUT/occ7b_ab.txt
cp /tmp/base.so $P/libls_amd.so

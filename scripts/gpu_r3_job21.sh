# GPU job 21: second sweep of k_chain_t variants (near batch x far depth for f64; launch bound / near batch for c128)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for v in a b c d e f; do
  cp $P/libls_amd_$v.so $P/libls_amd.so
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 > $OUT/var2_f64_$v.json 2>/dev/null
  timeout 600 python bench.py --dtype c128 --no-cpu-baseline --no-extra --steps 8 > $OUT/var2_c128_$v.json 2>/dev/null
  echo "$v: f64 $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var2_f64_$v.json | head -1) c128 $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var2_c128_$v.json | head -1)"
done | tee $OUT/chain_variants2.txt
cp /tmp/base.so $P/libls_amd.so

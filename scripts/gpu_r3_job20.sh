# GPU job 20: variants of the new k_chain_t (near table on/off and launch bound for c128; near batch / far depth for f64)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for v in base v1 v2 v3; do
  [ $v != base ] && cp $P/libls_amd_$v.so $P/libls_amd.so
  timeout 600 python bench.py --dtype c128 --no-cpu-baseline --no-extra --steps 8 > $OUT/var_c128_$v.json 2>/dev/null
  echo "c128 $v: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var_c128_$v.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/var_c128_$v.json | head -1)"
done | tee $OUT/chain_variants.txt
for v in base f1 f2 f3; do
  cp /tmp/base.so $P/libls_amd.so
  [ $v != base ] && cp $P/libls_amd_$v.so $P/libls_amd.so
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 > $OUT/var_f64_$v.json 2>/dev/null
  echo "f64 $v: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var_f64_$v.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/var_f64_$v.json | head -1)"
done | tee -a $OUT/chain_variants.txt
cp /tmp/base.so $P/libls_amd.so

# GPU job 17: full GPU suite on the tree after the K4 revert, default bench line, then k_tile_pull_idx at 7 blocks/CU
# (libls_amd_occ7.so: __launch_bounds__(256,7) + amdgpu_num_sgpr(94), halo 128 so the window fits 22.8 KB) against the shipped build
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_job17.log 2>&1; tail -3 $OUT/pytest_job17.log
timeout 600 python bench.py > $OUT/bench_default_job17.json 2>$OUT/bench_default_job17.err; cat $OUT/bench_default_job17.json | head -c 600; echo
P=distributed-matvec_amd
cp $P/libls_amd.so /tmp/base.so
for lib in base occ7; do
  [ $lib = occ7 ] && cp $P/libls_amd_occ7.so $P/libls_amd.so
  for m in 36 40; do for h in 128 512; do
    [ $lib = occ7 ] && [ $h = 512 ] && continue
    LS_AMD_PULL_HALO=$h timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 6 --warmup 2 --no-cpu-baseline > $OUT/occ7_${lib}_${m}_h$h.json 2>/dev/null
    echo "$lib chain_${m}_symm halo=$h: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/occ7_${lib}_${m}_h$h.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/occ7_${lib}_${m}_h$h.json | head -1)"
  done; done
done | tee $OUT/occ7_ab.txt
cp /tmp/base.so $P/libls_amd.so

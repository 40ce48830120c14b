export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests -m gpu -q -x -k "indexed" > $OUT/pytest_idx2.log 2>&1; tail -3 $OUT/pytest_idx2.log
for h in 512 384 256 160 64 0; do
  LS_AMD_PULL_HALO=$h timeout 300 python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline > $OUT/halo_36_$h.json 2>/dev/null
  echo "halo $h: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/halo_36_$h.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $OUT/halo_36_$h.json | head -1)"
done

# GPU job 15: ablation of the indexed pull kernel on chain_36_symm (profiling build: make ABLATE=1; wrong results by design)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
for a in 0 1 2 4 6 32 36; do
  LS_AMD_ABLATE=$a timeout 300 python bench.py --model heisenberg_chain_36_symm --steps 8 --warmup 2 --no-cpu-baseline > $OUT/ablate_idx_$a.json 2>/dev/null
  echo "ablate $a: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/ablate_idx_$a.json | head -1)"
done | tee $OUT/ablate_idx_36symm.txt

# GPU job 16: K4 of a packet from the runs of its row (k4_mode 4) -- parity, then A/B against mode 3 on chain_36_symm / chain_40_symm
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 -k "indexed or single_locale or replicated_exchange or ranks_as_threads or bethe or general_k4 or complex_characters or partitioned" > $OUT/pytest_k4.log 2>&1; tail -3 $OUT/pytest_k4.log
for m in 36 40; do for nb in 0 1; do
  LS_AMD_K4_NEIGHBOUR=$nb timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 6 --warmup 2 --no-cpu-baseline > $OUT/k4_${m}_nb$nb.json 2>/dev/null
  echo "chain_${m}_symm neighbour=$nb: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/k4_${m}_nb$nb.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/k4_${m}_nb$nb.json | head -1)"
done; done | tee $OUT/k4_neighbour_ab.txt
for h in 64 128 256 512; do
  LS_AMD_PULL_HALO=$h timeout 300 python bench.py --model heisenberg_chain_36_symm --steps 8 --warmup 2 --no-cpu-baseline > $OUT/k4_halo_$h.json 2>/dev/null
  echo "halo $h: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/k4_halo_$h.json | head -1)"
done | tee -a $OUT/k4_neighbour_ab.txt

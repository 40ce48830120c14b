# GPU job 36: stage A of k_tile_pull_wv without group records for the groups of one exchange run: parity subset, A/B
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "indexed or symm or bethe or kagome or complex_characters" > $OUT/pytest_job36.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job36.log | tail -2
for v in runa base runa base; do
  cp $P/libls_amd_$v.so $P/libls_amd.so
  for m in 36 40; do
    timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/runa_${v}_$m.json 2>/dev/null
    echo "$v chain_${m}_symm: $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/runa_${v}_$m.json | head -1)"
  done
done | tee $OUT/stage_a_run_ab.txt
cp $P/libls_amd_runa.so $P/libls_amd.so

# GPU job 1 (round 4): new indexed pull kernels (k_pull_t fused | resolve + gather) -- parity, then fused vs split timings on
# chain_36_symm / chain_40_symm; tile-order / cache-policy experiments of the staged row kernel on chain_32
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job1; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py -m gpu -q -x -k "indexed or complex_characters or projected or symm or replicated or ranks_as_threads or kagome or issue" > $OUT/pytest_a.log 2>&1 ) 2>&1 | grep real; tail -5 $OUT/pytest_a.log
for m in 36 40; do
  for sp in 0 6000000000; do
    LS_AMD_PULL_SPLIT=$sp timeout 600 python bench.py --model heisenberg_chain_${m}_symm --steps 5 --warmup 2 --no-cpu-baseline --kDisplayTimings > $OUT/bench_${m}symm_split$sp.json 2> $OUT/bench_${m}symm_split$sp.err
    python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_${m}symm_split$sp.json').read().strip().splitlines()[-1])
    print('chain_${m}_symm split=$sp', round(d['ms_per_step'],3), 'ms/matvec', d['roofline']['kernel'], 'kernel_ms', d['roofline']['kernel_ms_avg'])
except Exception as e:
    print('chain_${m}_symm split=$sp NO JSON', e)
PY
    grep -A9 "matrixVectorProduct \[" $OUT/bench_${m}symm_split$sp.err | head -12
  done
done
C=";LS_AMD_TILE_SETS=8;LS_AMD_TILE_SETS=8,LS_AMD_CHAIN_NT=12:24;LS_AMD_TILE_SETS=7,LS_AMD_CHAIN_NT=12:25;LS_AMD_TILE_SETS=6:8,LS_AMD_CHAIN_NT=12:26;LS_AMD_TILE_SETS=10,LS_AMD_CHAIN_NT=12:22;LS_AMD_CHAIN_NT=20:32;LS_AMD_CHAIN_NT=24:32;LS_AMD_TILE_SETS=8:4,LS_AMD_CHAIN_NT=12:24;LS_AMD_CHAIN_TILE=512;LS_AMD_CHAIN_TILE=512,LS_AMD_TILE_SETS=8,LS_AMD_CHAIN_NT=12:24;LS_AMD_CHAIN_TILE=512,LS_AMD_TILE_SETS=9,LS_AMD_CHAIN_NT=12:23;"
timeout 900 python scripts/order_sweep.py --steps 8 --configs "$C" > $OUT/order_sweep.log 2>&1; cut -c1-200 $OUT/order_sweep.log

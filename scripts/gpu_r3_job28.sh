# GPU job 28: k_direct with the span-limited rank shift for exchange groups: parity on the lattice models, non-chain benches
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "not symm and not c_example" > $OUT/pytest_job28.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job28.log | tail -2
timeout 900 python scripts/nonchain_bench.py --models square_6x5,j1j2_30,j1j2_32 > $OUT/nonchain_delta.txt 2>&1; grep -E '"model"|ms' $OUT/nonchain_delta.txt | cut -c1-260

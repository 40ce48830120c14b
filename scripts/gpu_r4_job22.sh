# GPU job 22 (round 4): basis rotation kernel of the thick restart: parity, eigensolver tests, time split
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job22; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_orth.py -m gpu -q 2>&1 | tail -2
( time timeout 900 python -m pytest tests/test_gpu_diagonalize.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "diagonal or bethe or published or distributed_eigensolve" > $OUT/pytest_eig.log 2>&1 ) 2>&1 | grep real; tail -2 $OUT/pytest_eig.log
LS_AMD_LANCZOS_PROFILE=1 timeout 600 python scripts/lanczos_profile.py 40 12 2>&1 | grep -E "chain_|profile after restart 7" | tee $OUT/lanczos_profile.txt
timeout 300 python scripts/lanczos_profile.py 36 16 2>&1 | grep -E "chain_" | tee -a $OUT/lanczos_profile.txt

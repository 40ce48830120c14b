# GPU job 20 (round 4): fused orthogonalisation sweeps in the eigensolver: parity tests that use it, time split before / after
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job20; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_diagonalize.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "diagonal or bethe or ground or lanczos or published or eight_ranks" > $OUT/pytest_eig.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_eig.log
for f in 1 0; do
  LS_AMD_FUSED_ORTH=$f timeout 600 python scripts/lanczos_profile.py 36 16 2>&1 | grep chain_ | sed "s/^/fused_orth=$f /"
  LS_AMD_FUSED_ORTH=$f timeout 900 python scripts/lanczos_profile.py 40 12 2>&1 | grep chain_ | sed "s/^/fused_orth=$f /"
done | tee $OUT/lanczos_profile_fused_orth.txt

# GPU job 21 (round 4): eigensolver time split with the scratch-free orth kernel
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job21; mkdir -p $OUT
python scripts/orth_bench.py 2>&1 | tail -2 | tee $OUT/orth_bench.txt
python scripts/orth_bench.py 63068876 14 2>&1 | tail -2 | tee -a $OUT/orth_bench.txt
timeout 600 python scripts/lanczos_profile.py 36 16 2>&1 | grep chain_ | tee $OUT/lanczos_profile.txt
timeout 900 python scripts/lanczos_profile.py 40 12 2>&1 | grep chain_ | tee -a $OUT/lanczos_profile.txt

# GPU job 19 (round 4): eigensolver time split (matvec vs orthogonalisation); the kernel-table slot cache test
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job19; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "kernel_table or slot_cache" 2>&1 | tail -2
timeout 600 python scripts/lanczos_profile.py 36 16 2>&1 | grep chain_ | tee $OUT/lanczos_profile.txt
timeout 900 python scripts/lanczos_profile.py 40 12 2>&1 | grep chain_ | tee -a $OUT/lanczos_profile.txt

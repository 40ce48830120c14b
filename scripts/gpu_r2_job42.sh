export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2final
timeout 900 python scripts/loopback_bench.py --L 32 --P 8 --steps 3 > gpurun_out/r2final/loopback_chain32_P8.txt 2>&1; tail -30 gpurun_out/r2final/loopback_chain32_P8.txt
timeout 900 python scripts/loopback_bench.py --L 36 --symm --P 8 --steps 3 > gpurun_out/r2final/loopback_chain36symm_P8.txt 2>&1; tail -24 gpurun_out/r2final/loopback_chain36symm_P8.txt | grep -v "^ "

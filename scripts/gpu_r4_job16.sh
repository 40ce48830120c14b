# GPU job 16 (round 4): where the 63 ms of heisenberg_square_6x6 go (profiling build: stage A | + K4 | everything)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job16; mkdir -p $OUT
timeout 600 python scripts/ablate_pull.py heisenberg_square_6x6 0 1 2 64 32 0 2>&1 | grep ablate | tee $OUT/ablate_square6x6.txt

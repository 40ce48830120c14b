export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2; python bench.py --steps 10 --warmup 3 --force-distributed --model heisenberg_chain_28 --no-cpu-baseline > gpurun_out/r2/bench_fd.out 2> gpurun_out/r2/bench_fd.err
tail -5 gpurun_out/r2/bench_fd.err; tail -c 3000 gpurun_out/r2/bench_fd.out

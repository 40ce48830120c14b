# GPU job 9 (round 4): sub-range exchange of the replicated-x mode (unprojected bases): parity over loop-back ranks, the share of
# x a rank receives on chain_28 / chain_32 at 8 ranks with per-stage times; torus_min with per-lane candidate popping on 6x6
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job9; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_loopback.py tests/test_gpu_rccl.py tests/test_gpu_two_process.py -m gpu -q -x > $OUT/pytest_loopback.log 2>&1 ) 2>&1 | grep real; tail -5 $OUT/pytest_loopback.log
( time timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_diagonalize.py -m gpu -q -x -k "replicated or square or lattice or state_info or slot_cache" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_focus.log
timeout 300 python scripts/lattice_bench.py heisenberg_square_6x6 5 2>&1 | tail -1 | tee $OUT/square6x6.jsonl
for L in 28 32; do
  timeout 600 python scripts/loopback_bench.py --L $L --P 8 --mode replicated --steps 3 > $OUT/loopback_chain${L}_reach.txt 2>&1; grep -E "ranks sharing|x received|aggregate" $OUT/loopback_chain${L}_reach.txt
  LS_AMD_REPL_REACH=0 timeout 600 python scripts/loopback_bench.py --L $L --P 8 --mode replicated --steps 3 > $OUT/loopback_chain${L}_whole.txt 2>&1; grep -E "ranks sharing|x received|aggregate" $OUT/loopback_chain${L}_whole.txt
done

# GPU job 10 (round 4): the share of x a rank receives in the sub-range exchange at 2 and 4 ranks (chain_28, loop-back)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job10; mkdir -p $OUT
for P in 2 4; do
  timeout 600 python scripts/loopback_bench.py --L 28 --P $P --mode replicated --steps 3 > $OUT/loopback_chain28_P${P}_reach.txt 2>&1; grep -E "ranks sharing|x received|aggregate" $OUT/loopback_chain28_P${P}_reach.txt
  LS_AMD_REPL_REACH=-14 timeout 600 python scripts/loopback_bench.py --L 28 --P $P --mode replicated --steps 3 > $OUT/loopback_chain28_P${P}_forced.txt 2>&1; grep -E "ranks sharing|x received|aggregate" $OUT/loopback_chain28_P${P}_forced.txt
done

# GPU job 2 (round 4): where the time of k_pull_t goes (ablation build), the split matvec on eight loop-back ranks, VALU count
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job2; mkdir -p $OUT
timeout 600 python scripts/ablate_pull.py heisenberg_chain_36_symm 0 1 2 4 32 64 6 0 2>&1 | grep ablate | tee $OUT/ablate_pull_chain36symm.txt
for sp in 0 32000000000; do
  LS_AMD_PULL_SPLIT=$sp timeout 600 python scripts/loopback_bench.py --L 36 --symm --P 8 --mode replicated --steps 3 > $OUT/loopback_36symm_split$sp.txt 2>&1
  grep -E "ranks sharing|aggregate|rank 0|producers|row kernel|exchange wait|x prep|returned" $OUT/loopback_36symm_split$sp.txt | cut -c1-190
done
LS_AMD_PULL_SPLIT=8000000000 timeout 900 python scripts/loopback_bench.py --L 40 --symm --P 8 --mode replicated --steps 2 > $OUT/loopback_40symm_split8e9.txt 2>&1
grep -E "ranks sharing|aggregate|rank 0|producers|row kernel|exchange wait|x prep|returned|Error|error" $OUT/loopback_40symm_split8e9.txt | cut -c1-190
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r4_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -30

cd $GRAFT_REPO_ROOT
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r2a_chain32_f64 bash scripts/gpu_pmc_traffic.sh 2>&1 | grep -A12 '^{' | head -16
MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r2a_chain32_c128 bash scripts/gpu_pmc_traffic.sh 2>&1 | grep -A12 '^{' | head -16
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r2a_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh 2>&1 | grep -A12 '^{' | head -16
grep -h "k_chain_t\|k_tile_pull" gpurun_out/r2a_*/summary.txt | cut -c1-60,80-140 | sort -u | head -40

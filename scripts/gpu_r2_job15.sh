cd $GRAFT_REPO_ROOT
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r2a_chain32_f64 bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -25

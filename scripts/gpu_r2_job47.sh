export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "staged or row_kernel or block_rows or chain_32_edge or single_locale" 2>&1 | tail -2
for v in 0 1 0 1 0 1; do
  echo "== LS_AMD_CHAIN_KS=$v"; LS_AMD_CHAIN_KS=$v timeout 300 python scripts/order_sweep.py --L 32 --steps 10 --configs ";" 2>&1 | grep staged | head -1 | cut -c1-110
done

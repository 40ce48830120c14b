export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LS_AMD_CHAIN_BIG=1 timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "staged or row_kernel or block_rows or chain_32_edge" 2>&1 | tail -2
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs ";LS_AMD_CHAIN_BIG=1;;LS_AMD_CHAIN_BIG=1;LS_AMD_CHAIN_BIG=1,LS_AMD_TILE_CHUNK=128;LS_AMD_CHAIN_BIG=1,LS_AMD_TILE_CHUNK=64" 2>&1 | grep staged | cut -c1-130

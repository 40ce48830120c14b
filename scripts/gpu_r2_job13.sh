export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "staged or row_kernel or block_rows" 2>&1 | tail -3
LS_AMD_CHAIN_TILE=2 timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "staged or row_kernel or block_rows" 2>&1 | tail -3
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs ";LS_AMD_CHAIN_TILE=2;LS_AMD_CHAIN_TILE=2,LS_AMD_TILE_CHUNK=128;LS_AMD_CHAIN_MAXLO=0;LS_AMD_CHAIN_MAXLO=0,LS_AMD_CHAIN_TILE=2;" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 5 --dtype c128 --configs ";LS_AMD_CHAIN_TILE=2" 2>&1 | grep -v amdgpu.ids | cut -c1-220

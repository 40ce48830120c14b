export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "staged or block_rows" 2>&1 | tail -3
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "LS_AMD_CHAIN_OLD=1;LS_AMD_CHAIN_OLD=0;LS_AMD_CHAIN_OLD=1;LS_AMD_CHAIN_OLD=0" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --dtype c128 --configs "LS_AMD_CHAIN=1" 2>&1 | grep -v amdgpu.ids | cut -c1-220

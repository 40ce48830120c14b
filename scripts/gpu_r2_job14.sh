export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "staged or row_kernel or block_rows or chain_32 or single_locale or replicated" 2>&1 | tail -3
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "LS_AMD_CHAIN_REC=0;LS_AMD_CHAIN_REC=1;LS_AMD_CHAIN_REC=0;LS_AMD_CHAIN_REC=1;LS_AMD_CHAIN_REC=1,LS_AMD_CHAIN_MAXLO=0;LS_AMD_CHAIN_REC=1,LS_AMD_CHAIN_MAXLO=31" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 5 --dtype c128 --configs "LS_AMD_CHAIN_REC=0;LS_AMD_CHAIN_REC=1" 2>&1 | grep -v amdgpu.ids | cut -c1-220

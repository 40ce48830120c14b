export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "staged or chain_32 or single_locale or row_kernel or block_rows or replicated or chain_24" 2>&1 | tail -8
C="LS_AMD_CHAIN_OLD=1;LS_AMD_CHAIN_OLD=0;LS_AMD_CHAIN_OLD=1;LS_AMD_CHAIN_OLD=0;LS_AMD_CHAIN_OLD=0,LS_AMD_TILE_CHUNK=0;LS_AMD_CHAIN_OLD=0,LS_AMD_CHAIN_MAXLO=12;LS_AMD_CHAIN_OLD=0,LS_AMD_CHAIN_MAXLO=20;LS_AMD_CHAIN_OLD=0,LS_AMD_CHAIN_MAXLO=31"
sed -i 's/"LS_AMD_CHAIN_MAXLO")/"LS_AMD_CHAIN_MAXLO", "LS_AMD_CHAIN_OLD", "LS_AMD_CHAIN")/' scripts/order_sweep.py
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "$C" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --dtype c128 --configs "LS_AMD_CHAIN=0;LS_AMD_CHAIN=1;LS_AMD_CHAIN=1,LS_AMD_TILE_CHUNK=0" 2>&1 | grep -v amdgpu.ids | cut -c1-220

export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "LS_AMD_CHAIN_FULLGRID=0;LS_AMD_CHAIN_FULLGRID=1;LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=0;LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=4;LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=256;LS_AMD_CHAIN_FULLGRID=0" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --dtype c128 --configs "LS_AMD_CHAIN_FULLGRID=0;LS_AMD_CHAIN_FULLGRID=1" 2>&1 | grep -v amdgpu.ids | cut -c1-220

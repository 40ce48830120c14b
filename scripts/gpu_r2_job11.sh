export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=""
for c in 32 64 128 256 512 1024 4096 16384; do C="$C;LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=$c"; done
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "${C#;}" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 5 --configs "LS_AMD_CHAIN=0;LS_AMD_CHAIN=0,LS_AMD_CHAIN_FULLGRID=1;LS_AMD_CHAIN=0,LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=256" 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 600 python scripts/order_sweep.py --L 32 --steps 5 --dtype c128 --configs "LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=256;LS_AMD_CHAIN_FULLGRID=1,LS_AMD_TILE_CHUNK=1024;LS_AMD_CHAIN=0,LS_AMD_CHAIN_FULLGRID=1" 2>&1 | grep -v amdgpu.ids | cut -c1-220

export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs ";LS_AMD_CHAIN_SMALL=1;;LS_AMD_CHAIN_SMALL=1;LS_AMD_CHAIN_SMALL=1,LS_AMD_TILE_CHUNK=512" 2>&1 | grep staged | cut -c1-130

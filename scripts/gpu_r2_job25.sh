export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/order_sweep.py --L 32 --steps 10 --configs ";;LS_AMD_HIGH_PAIR=13;LS_AMD_HIGH_PAIR=16" 2>&1 | grep -v amdgpu.ids | cut -c1-200

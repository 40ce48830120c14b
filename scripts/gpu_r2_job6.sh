export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -12 ) 2>&1
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs "LS_AMD_CHAIN_OLD=1;LS_AMD_CHAIN_OLD=0" 2>&1 | grep -v amdgpu.ids | cut -c1-220

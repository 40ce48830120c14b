export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=""
for cfg in "3 32" "4 32" "4 8" "5 16" "6 8" "6 2" "8 2" "8 1" "4 128" "10 1"; do set -- $cfg; C="$C;LS_AMD_TILE_SETS=$1,LS_AMD_TILE_SET_TILES=$2"; done
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs ";${C#;};" 2>&1 | grep -v amdgpu.ids | cut -c1-200

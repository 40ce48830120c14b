export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3; do echo "== SCATTER_ABLATE=$a"; LS_AMD_SCATTER_ABLATE=$a timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L="; done
echo "== full grid scatter"; LS_AMD_SCATTER_GRID=1 timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L="
echo "== P=2"; timeout 300 python scripts/tile_bench.py --L 28 --P 2 --steps 5 2>&1 | grep "L="
echo "== chain32 P=8"; timeout 300 python scripts/tile_bench.py --L 32 --P 8 --steps 2 2>&1 | grep "L="

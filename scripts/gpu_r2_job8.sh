export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for p in 1 0; do
echo "== LS_AMD_TILE_PERSISTENT=$p"
LS_AMD_TILE_PERSISTENT=$p timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 5 2>&1 | grep "L="
LS_AMD_TILE_PERSISTENT=$p timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 3 --mode push 2>&1 | grep "L="
LS_AMD_TILE_PERSISTENT=$p timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L="
LS_AMD_TILE_PERSISTENT=$p timeout 300 python scripts/tile_bench.py --L 36 --symm --P 8 --steps 3 2>&1 | grep "L="
done

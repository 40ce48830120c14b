export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 1 8 16 24; do echo "== LS_AMD_ABLATE=$a"; LS_AMD_ABLATE=$a timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L=" | cut -c60-250; done

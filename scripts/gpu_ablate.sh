set -x
export TMPDIR=/tmp
for a in 0 1 2 4 6; do
  echo "ABLATE=$a"; LS_AMD_ABLATE=$a python scripts/tile_bench.py --L 36 --symm --P 1 --steps 3 2>&1 | grep -E "^L="
done

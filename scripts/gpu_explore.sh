set -x
export TMPDIR=/tmp
python scripts/tile_bench.py --L 28 --P 2
python scripts/tile_bench.py --L 28 --P 8
python scripts/tile_bench.py --L 24 --symm --P 1
python scripts/tile_bench.py --L 32 --symm --P 1
python scripts/tile_bench.py --L 32 --symm --P 8
timeout 600 python scripts/tile_bench.py --L 36 --symm --P 1 --steps 2

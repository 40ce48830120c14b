set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -8
python scripts/tile_bench.py --L 24 --symm --P 1
python scripts/tile_bench.py --L 32 --symm --P 1
python scripts/tile_bench.py --L 32 --symm --P 8
timeout 600 python scripts/tile_bench.py --L 36 --symm --P 1 --steps 2

# GPU job: parity suite + symmetric timing (quick iteration loop)
set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
python scripts/tile_bench.py --L 36 --symm --P 1 --steps 3
python scripts/tile_bench.py --L 32 --symm --P 1 --steps 5

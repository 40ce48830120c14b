# GPU job: parity suite + symmetric benches (quick iteration loop)
set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
python scripts/tile_bench.py --L 36 --symm --P 1 --steps 3
LS_AMD_K4_BRUTE=1 python scripts/tile_bench.py --L 36 --symm --P 1 --steps 3
python scripts/tile_bench.py --L 32 --symm --P 1 --steps 5
python scripts/tile_bench.py --L 36 --symm --P 8 --steps 3
python scripts/tile_bench.py --L 40 --symm --P 1 --steps 2

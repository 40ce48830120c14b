export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_two_process.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 36 --symm --P 8 --steps 3 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 3 --mode push 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 40 --symm --steps 2 2>&1 | grep "L="

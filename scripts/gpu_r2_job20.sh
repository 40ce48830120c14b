export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_two_process.py tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 28 --P 2 --steps 5 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 36 --symm --P 8 --steps 3 2>&1 | grep "L="
timeout 300 python scripts/tile_bench.py --L 32 --P 8 --steps 2 2>&1 | grep "L="

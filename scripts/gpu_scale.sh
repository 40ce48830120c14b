set -x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matvec.py -m gpu -q -k "full_size" 2>&1 | tail -8
timeout 600 python bench.py --model heisenberg_chain_40_symm --steps 2 --warmup 1 --no-cpu-baseline --no-extra

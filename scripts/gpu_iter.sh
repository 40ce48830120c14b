# GPU job: parity suite + headline bench (quick iteration loop)
set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-}

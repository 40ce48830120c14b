# GPU job: parity suite + headline bench A/B (quick iteration loop)
set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
for t in 0 8 10; do
  LS_AMD_HIGH_BITS=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
done
LS_AMD_HIGH_BITS=8 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --dtype c128

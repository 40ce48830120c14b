# GPU job: parity suite + headline A/B
set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
for v in 1 0 1 0; do
  LS_AMD_PULL2=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
done
LS_AMD_PULL2=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --dtype c128
LS_AMD_PULL2=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --dtype c128

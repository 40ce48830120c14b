# GPU job: parity + A/B of the wave-uniform far pairs (LS_AMD_HIGH_PAIR = first pair handled that way)
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for v in 0 10 12 14 16 18; do
  echo "+ HIGH_PAIR=$v"; LS_AMD_HIGH_PAIR=$v $B
done
echo "+ HIGH_PAIR=14 TOP_BITS=0"; LS_AMD_HIGH_PAIR=14 LS_AMD_TOP_BITS=0 $B

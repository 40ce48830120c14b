# GPU job: parity suite + c128 timing with the wave-uniform far pairs
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3
B="timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
echo "+ c128 HIGH_PAIR=0"; LS_AMD_HIGH_PAIR=0 $B --dtype c128
echo "+ c128 default"; $B --dtype c128
echo "+ c128 HIGH_PAIR=12"; LS_AMD_HIGH_PAIR=12 $B --dtype c128
echo "+ f64 default"; $B

# GPU job: parity suite + timing of the staged kernel
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
echo "+ default"; $B
echo "+ BLOCKS=6"; LS_AMD_BLOCKS_PER_CU=6 $B
echo "+ HIGH_PAIR=14"; LS_AMD_HIGH_PAIR=14 $B

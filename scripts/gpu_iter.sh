# GPU job: parity + timing of the staged row kernel (k_chain) with cached ring-closing partners
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
echo "+ CHAIN=0"; LS_AMD_CHAIN=0 $B
echo "+ CHAIN=1"; $B
echo "+ CHAIN=1 HIGH_PAIR=12"; LS_AMD_HIGH_PAIR=12 $B
echo "+ CHAIN=1 BLOCKS=6"; LS_AMD_BLOCKS_PER_CU=6 $B
echo "+ CHAIN=1 BLOCKS=5"; LS_AMD_BLOCKS_PER_CU=5 $B

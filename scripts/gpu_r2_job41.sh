export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
echo "== LS_AMD_HASH_SORTED=$v"
LS_AMD_HASH_SORTED=$v timeout 300 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "chain_24_symm or kagome_12_symm" 2>&1 | tail -1
LS_AMD_HASH_SORTED=$v timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 5 2>&1 | grep "L=" | cut -c60-200
done
LS_AMD_HASH_SORTED=1 timeout 300 python scripts/tile_bench.py --L 40 --symm --steps 3 2>&1 | grep "L=" | cut -c60-200
LS_AMD_HASH_SORTED=0 timeout 300 python scripts/tile_bench.py --L 40 --symm --steps 3 2>&1 | grep "L=" | cut -c60-200

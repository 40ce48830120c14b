# GPU job 13 (round 4): XCD-chunked block -> tile map of the pull kernels (k_pull_t fused, k_pull_gather cached): A/B
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job13; mkdir -p $OUT
for C in 0 4 16 64 256; do
  echo -n "chunk=$C cached 36: "; LS_AMD_PULL_XCD_CHUNK=$C timeout 300 python scripts/tile_bench.py --L 36 --symm --mode pull --cache --steps 8 2>&1 | tail -1 | grep -o "matvec=.*"
  echo -n "chunk=$C fused 36: "; LS_AMD_PULL_XCD_CHUNK=$C timeout 300 python scripts/tile_bench.py --L 36 --symm --mode pull --steps 6 2>&1 | tail -1 | grep -o "matvec=.*"
done | tee $OUT/xcd_chunk_ab.txt
for C in 0 16 64; do
  echo -n "chunk=$C cached 40: "; LS_AMD_PULL_XCD_CHUNK=$C timeout 600 python scripts/tile_bench.py --L 40 --symm --mode pull --cache --steps 4 2>&1 | tail -1 | grep -o "matvec=.*"
  echo -n "chunk=$C fused 40: "; LS_AMD_PULL_XCD_CHUNK=$C timeout 600 python scripts/tile_bench.py --L 40 --symm --mode pull --steps 3 2>&1 | tail -1 | grep -o "matvec=.*"
done | tee -a $OUT/xcd_chunk_ab.txt

# GPU job 23 (round 4): XCD-chunked packet blocks in the consumer k_scatter (full grid + chunk map) -- A/B on chain_28 x 8
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job23; mkdir -p $OUT
for C in 0 1 16 64 256 1024; do
  echo -n "scatter chunk=$C: "; LS_AMD_SCATTER_XCD_CHUNK=$C timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 --tree 2>&1 | grep -E "matvec=|consumers" | tr '\n' ' ' | grep -o "matvec=[0-9.]*ms.*consumers[^m]*ms"
done | tee $OUT/scatter_xcd_chunk_ab.txt

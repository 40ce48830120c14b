# GPU job r2/2: what do the near / middle / far pairs of the staged row kernel cost (ablation by LS_AMD_CHAIN_MAXLO,
# wrong results by construction), and the chunk size of the round-robin tile dealing
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
C=""
for m in 0 12 14 16 18 20 22 24 26 28 31; do C="$C;LS_AMD_CHAIN_MAXLO=$m"; done
for c in 4 8 16 32 64 128; do C="$C;LS_AMD_TILE_CHUNK=$c"; done
for c in 16 32; do C="$C;LS_AMD_TILE_CHUNK=$c,LS_AMD_CHAIN_MAXLO=12"; done
timeout 600 python scripts/order_sweep.py --L 32 --steps 6 --configs ";${C#;}" > gpurun_out/r2/ablate_sweep.log 2>&1
cat gpurun_out/r2/ablate_sweep.log | cut -c1-200

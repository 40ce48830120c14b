# GPU job 38: environment knobs of the new k_chain_t (first uniform pair, tiles per round-robin chunk), chain_32 f64
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3; mkdir -p $OUT
run() { echo "$1: $(env $1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 2>/dev/null | grep -o '"kernel_ms_avg": [0-9.]*' | head -1)"; }
for cfg in "LS_AMD_HIGH_PAIR=12" "LS_AMD_HIGH_PAIR=13" "LS_AMD_HIGH_PAIR=14" "LS_AMD_TILE_CHUNK=128" "LS_AMD_TILE_CHUNK=512" "LS_AMD_TILE_CHUNK=1024" "LS_AMD_HIGH_PAIR=12"; do run "$cfg"; done | tee $OUT/chain_knobs.txt

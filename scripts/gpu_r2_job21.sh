export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export CMD="python $GRAFT_REPO_ROOT/scripts/tile_bench.py --L 28 --P 8 --steps 3"
bash scripts/gpu_profile_cmd.sh r2_pkt28 > /dev/null 2>&1
grep -E "k_tile|k_scatter" gpurun_out/prof_r2_pkt28/summary.txt | cut -c1-50,60-140

# GPU job r2/1: (a) tile-order sweep of the staged row kernel on chain_32 (one process, plan per order);
# (b) baseline of the packet path (k_tile + k_scatter) with 8 logical partitions: kernel trace + fabric counters.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 > gpurun_out/r2/order_sweep.log 2>&1
tail -15 gpurun_out/r2/order_sweep.log
export CMD="python $GRAFT_REPO_ROOT/scripts/tile_bench.py --L 28 --P 8 --steps 3"
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28/trace -o trace -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28_trace.log 2>&1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28/pmc_fetch -o pmc -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28_fetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28/pmc_write -o pmc -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/r2/pkt28_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py gpurun_out/r2/pkt28 > gpurun_out/r2/pkt28_summary.txt 2>&1
rm -rf gpurun_out/r2/pkt28/*/*.db gpurun_out/r2/pkt28/*/*/*.db
grep -h "L=28" gpurun_out/r2/pkt28_trace.log
cut -c1-170 gpurun_out/r2/pkt28_summary.txt | head -60

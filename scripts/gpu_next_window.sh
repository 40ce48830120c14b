# GPU job, first call of the next GPU window (~6 min of box time): everything the near window of k_tile_pull still owes --
# the whole -m gpu suite on the final source, the PMC passes of the windowed kernel (FETCH / WRITE / VALU: the entry in
# profiles/pmc_traffic.json predates the window and is no longer attached), where its time goes (ablation, window on/off),
# a halo sweep, and the default bench line with its traffic attached.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash scripts/gpu_next_window.sh'
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/next; mkdir -p $O
( time timeout 300 python -m pytest tests -m gpu -q --maxfail=5 -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real
tail -2 $O/pytest_gpu.log
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=next_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_merge.py next_chain36symm_f64
cp profiles/pmc_traffic.json $O/pmc_traffic.json
B="timeout 60 python bench.py --model heisenberg_chain_36_symm --steps 6 --warmup 2 --no-cpu-baseline --no-extra"
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', round(d['ms_per_step'],3), 'ms/matvec, kernel', round(r['kernel_ms_avg'],3), 'ms')"; }
# profiling only (wrong results): 1 = no stage B, 2 = K4 but no look-ups, 4 = every key is the row's own (always a window hit)
for a in 0 1 2 4; do for h in 512 0; do
  LS_AMD_ABLATE=$a LS_AMD_PULL_HALO=$h $B > $O/ablate${a}_halo$h.json 2>/dev/null; line $O/ablate${a}_halo$h.json "ablate $a halo $h"
done; done 2>&1 | tee $O/ablation.txt
for h in 64 128 256 384 512; do LS_AMD_PULL_HALO=$h $B > $O/halo$h.json 2>/dev/null; line $O/halo$h.json "halo $h"; done 2>&1 | tee $O/halo_sweep.txt
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', round(d['value'],2), 'matvec/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],3), 'traffic', r['traffic'], 'frac_traffic', r['frac_traffic'], r['traffic_note'])"

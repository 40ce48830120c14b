# GPU job 15 (round 4; re-run as the final evidence job): whole -m gpu suite on the current tree, smoke, fresh PMC passes of the headline kernels (k_pull_t changed:
# XCD-chunked tile map), the default bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job24; mkdir -p $OUT
( time timeout 1700 python -m pytest tests -m gpu -q --maxfail=5 > $OUT/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -6 $OUT/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r4_chain32_f64 bash scripts/gpu_pmc_traffic.sh > $OUT/pmc32.log 2>&1; tail -4 $OUT/pmc32.log
PASSES=min MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r4_chain32_c128 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r4_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_40_symm DTYPE=f64 TAG=r4_chain40symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_merge.py r4_chain32_f64 r4_chain32_c128 r4_chain36symm_f64 r4_chain40symm_f64 2>&1 | tail -3
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/r4_chain*_rocprof_summary.txt profiles/r4_chain*_bench_line.json $OUT/ 2>/dev/null
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; tail -2 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4job24/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['value'],2),'matvec/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'wasted', r.get('wasted_traffic'), 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
for k,v in d['extra'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('matvecs_per_s','ms_per_step','kernel','kernel_ms_avg','error','slot_cache','pmc_note')})
PY
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4job24/bench_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r4job24/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py gpurun_out/r4job24/bench_trace > gpurun_out/r4job24/bench_trace_summary.txt 2>&1
rm -rf gpurun_out/r4job24/bench_trace/*.db gpurun_out/r4job24/bench_trace/*/*.db
head -16 gpurun_out/r4job24/bench_trace_summary.txt | cut -c1-150

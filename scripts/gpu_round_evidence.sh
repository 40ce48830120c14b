# GPU job: final evidence after the c128 window change -- whole -m gpu suite, smoke, PMC traffic of the three headline
# workloads (merged on the box so that the bench lines carry it), bench lines
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2final
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -4 ) 2>&1 | grep -v "^$"
python __graft_entry__.py smoke 2>&1 | tail -2
PASSES=min MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r2_chain32_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
PASSES=min MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r2_chain32_c128 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
PASSES=min MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r2_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_merge.py r2_chain32_f64 r2_chain32_c128 r2_chain36symm_f64
cp profiles/pmc_traffic.json gpurun_out/r2final/pmc_traffic.json
python bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err; tail -2 gpurun_out/r2final/bench_default.err
python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2final/bench_36symm.json 2>/dev/null
python bench.py --model heisenberg_chain_32 --dtype c128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2final/bench_32_c128.json 2>/dev/null
for f in gpurun_out/r2final/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
r=d['roofline']
print('$f'.split('/')[-1], round(d['value'],2), 'matvec/s', round(d['ms_per_step'],3),'ms', r['kernel'], 'frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'frac_traffic', r.get('frac_traffic'))"; done

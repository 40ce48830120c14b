# GPU job: the round's final evidence -- whole -m gpu suite, default bench line, PMC traffic of the three headline workloads
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2final
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -4 ) 2>&1 | grep -v "^$"
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err; tail -2 gpurun_out/r2final/bench_default.err
python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2final/bench_36symm.json 2>/dev/null
python bench.py --model heisenberg_chain_40_symm --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2final/bench_40symm.json 2>/dev/null
python bench.py --model heisenberg_chain_24 --dtype c128 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2final/bench_24_c128.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --force-distributed --model heisenberg_chain_32 --no-cpu-baseline --kDisplayTimings > gpurun_out/r2final/bench_32_one_rank_distributed.json 2> gpurun_out/r2final/bench_32_one_rank_distributed.err
MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r2_chain32_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r2_chain32_c128 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r2_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_32 DTYPE=f64 MODE=push TAG=r2_chain32_f64_push bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
cat gpurun_out/r2_chain*/pmc_traffic_entry.json | grep -E "traffic_bytes|heisenberg|source_sha"
for f in gpurun_out/r2final/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
r=d['roofline']
print('$f'.split('/')[-1], round(d['value'],2), 'matvec/s', round(d['ms_per_step'],3),'ms', r['kernel'], 'frac', round(r['frac'],3), d.get('exchanges') and {k:round(v.get('ms_per_step',0),3) for k,v in d['exchanges'].items()})"; done

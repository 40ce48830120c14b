# GPU job: round-3 evidence on the final tree -- whole -m gpu suite, smoke, PMC traffic of the headline workloads (merged on the box
# so that the bench lines carry it), bench lines of every BASELINE config, the self-launching multi-GPU bench on a one-GPU box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3final; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q --maxfail=5 > $OUT/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -4 $OUT/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2
PASSES=min MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r3_chain32_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
PASSES=min MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r3_chain32_c128 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r3_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
MODEL=heisenberg_chain_40_symm DTYPE=f64 TAG=r3_chain40symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
python scripts/pmc_traffic_merge.py r3_chain32_f64 r3_chain32_c128 r3_chain36symm_f64 r3_chain40symm_f64
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/r3_chain*_rocprof_summary.txt profiles/r3_chain*_bench_line.json $OUT/ 2>/dev/null
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; tail -2 $OUT/bench_default.err
python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_36symm.json 2>/dev/null
python bench.py --model heisenberg_chain_40_symm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_40symm.json 2>/dev/null
python bench.py --model heisenberg_chain_32 --dtype c128 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_32_c128.json 2>/dev/null
python bench.py --model heisenberg_chain_24 --dtype c128 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_24_c128.json 2>/dev/null
python bench.py --force-distributed --steps 5 --warmup 2 --no-cpu-baseline --kDisplayTimings > $OUT/bench_32_one_rank_distributed.json 2> $OUT/bench_32_one_rank_distributed.err
python bench.py --force-distributed --model heisenberg_chain_36_symm --steps 5 --warmup 2 --no-cpu-baseline --kDisplayTimings > $OUT/bench_36symm_one_rank_distributed.json 2> $OUT/bench_36symm_one_rank_distributed.err
for f in $OUT/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
except Exception as e:
    print('$f', 'NO JSON', e); sys.exit(0)
r=d['roofline']
print('$f'.split('/')[-1], round(d['value'],2), 'matvec/s', round(d['ms_per_step'],3),'ms', r['kernel'], 'frac', round(r['frac'],3) if r['frac'] else None, 'traffic', r.get('traffic'), 'frac_traffic', r.get('frac_traffic'), d.get('exchanges') and {k:(v.get('ms_per_step') or v.get('error')) for k,v in d['exchanges'].items()})"; done
timeout 300 python scripts/k4_rate.py heisenberg_chain_36_symm 2>&1 | grep variant > $OUT/k4_rate.txt; cat $OUT/k4_rate.txt
echo "--- self-launch on a one-GPU box:"; python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2

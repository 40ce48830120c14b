export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
tail -3 gpurun_out/r2/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/r2/bench_default.json'))
print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','extra')}, indent=1))"
python bench.py --steps 10 --warmup 3 --force-distributed --model heisenberg_chain_28 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({k:d[k] for k in ('value','ms_per_step','config','exchanges')}, indent=1))"

export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q 2>&1 | grep -E "passed|failed" | head -3
python bench.py --steps 10 --warmup 3 --force-distributed --model heisenberg_chain_32 --no-cpu-baseline --kDisplayTimings 2> gpurun_out/r2final/bench_32_one_rank_distributed.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        open('gpurun_out/r2final/bench_32_one_rank_distributed.json','w').write(line)
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['config']['exchange'], d['exchanges'], d['failed_exchanges'])"
python bench.py --steps 5 --warmup 2 --force-distributed --model heisenberg_chain_36_symm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['config']['exchange'], d['exchanges'], d['failed_exchanges'])"

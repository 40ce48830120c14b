export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_c_example.py -m gpu -q 2>&1 | grep -E "passed|failed" | head -3
python bench.py --steps 10 --warmup 3 --force-distributed --exchange replicated --model heisenberg_chain_32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['exchanges'], d['failed_exchanges'])"

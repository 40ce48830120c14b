# GPU job 17 (round 4): the N > 1 code path of bench.py on one rank over RCCL (--force-distributed): both exchanges, x in-bytes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job17; mkdir -p $OUT
( time python bench.py --force-distributed --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_one_rank_distributed.json 2> $OUT/bench_one_rank_distributed.err ) 2>&1 | grep real; tail -3 $OUT/bench_one_rank_distributed.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4job17/bench_one_rank_distributed.json').read().strip().splitlines()[-1])
print(d['value'], d['config'].get('exchange'), json.dumps(d['exchanges'])[:900], d.get('failed_exchanges'))
PY
( time python bench.py --force-distributed --model heisenberg_chain_36_symm --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_one_rank_distributed_36symm.json 2> $OUT/bench_one_rank_distributed_36symm.err ) 2>&1 | grep real; tail -3 $OUT/bench_one_rank_distributed_36symm.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4job17/bench_one_rank_distributed_36symm.json').read().strip().splitlines()[-1])
print(d['value'], d['config'].get('exchange'), json.dumps(d['exchanges'])[:900], d.get('failed_exchanges'))
PY

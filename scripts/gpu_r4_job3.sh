# GPU job 3 (round 4): the reference's benchmark model (heisenberg_square_6x6) -- enumeration, matvec with K4 as
# translations x point-group cosets against the element loop, E0 against the published value; the touched GPU tests; the default
# bench line with the projected-basis extras
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job3; mkdir -p $OUT
timeout 600 python scripts/lattice_bench.py heisenberg_square_6x6 5 2>&1 | grep model | tee $OUT/square6x6_cosets.json
LS_AMD_K4=brute timeout 900 python scripts/lattice_bench.py heisenberg_square_6x6 2 2>&1 | grep model | tee $OUT/square6x6_element_loop.json
( time timeout 1500 python -m pytest tests/test_gpu_diagonalize.py tests/test_gpu_matvec.py tests/test_gpu_loopback.py tests/test_hdf5_io.py -m gpu -q -x > $OUT/pytest_b.log 2>&1 ) 2>&1 | grep real; tail -5 $OUT/pytest_b.log
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4job3/bench_default.json').read().strip().splitlines()[-1])
print(round(d['value'],2),'matvec/s', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
for k,v in d['extra'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('matvecs_per_s','ms_per_step','kernel','kernel_ms_avg','error','requests_64B_per_s')})
PY

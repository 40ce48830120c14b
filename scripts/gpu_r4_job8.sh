# GPU job 8 (round 4): torus row table in K4 mode 4 (square lattices) and the slot cache (ls_amd_plan_cache_slots): parity first,
# then timings -- the reference's benchmark model, the default bench line with the cached legs of chain_36/40_symm
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job8; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py tests/test_gpu_diagonalize.py -m gpu -q -x -k "slot_cache or indexed or replicated_exchange or square or state_info or lattice" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -5 $OUT/pytest_focus.log
( time timeout 600 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "bethe" > $OUT/pytest_bethe.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_bethe.log
timeout 300 python scripts/lattice_bench.py heisenberg_square_6x6 5 2>&1 | tail -1 | tee $OUT/square6x6.jsonl
timeout 300 python scripts/lattice_bench.py heisenberg_square_4x4 5 2>&1 | tail -1 | tee -a $OUT/square6x6.jsonl
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4job8/bench_default.json').read().strip().splitlines()[-1])
print(round(d['value'],2),'matvec/s', round(d['ms_per_step'],3), 'ms')
for k,v in d['extra'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('matvecs_per_s','ms_per_step','kernel','kernel_ms_avg','error','slot_cache')})
PY

# GPU job: near window of the staged pull kernel (k_tile_pull) -- the whole GPU suite with the window on (the default),
# then the A/B against the table-only path on chain_36_symm (and chain_40_symm when time is left)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/window
timeout 140 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/window/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/window/pytest.log
for m in heisenberg_chain_36_symm heisenberg_chain_40_symm; do
  for h in 512 0; do
    LS_AMD_PULL_HALO=$h timeout 40 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/window/bench_${m}_halo$h.json 2>/dev/null
    python -c "
import json; d=json.loads(open('gpurun_out/window/bench_${m}_halo$h.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$m halo $h', round(d['ms_per_step'],3), 'ms/matvec kernel', round(r['kernel_ms_avg'],3), 'ms')"
  done
done

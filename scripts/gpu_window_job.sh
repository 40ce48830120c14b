# GPU job: near window of the staged pull kernel (k_tile_pull) -- parity of everything that runs it, A/B against the
# table-only path on chain_36_symm, then the PMC traffic of the three headline workloads for this source
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/window
timeout 200 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py tests/test_gpu_loopback.py -m gpu -q -x -p no:cacheprovider \
  -k "single_locale or replicated_x_block_rows or chain_36_symm or complex_char or golden or (24_symm and replicated)" > gpurun_out/window/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/window/pytest.log | tail -3
B="timeout 100 python bench.py --model heisenberg_chain_36_symm --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for h in 0 512; do
  LS_AMD_PULL_HALO=$h $B > gpurun_out/window/bench_36symm_halo$h.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/window/bench_36symm_halo$h.json').read().strip().splitlines()[-1]); r=d['roofline']
print('halo $h', round(d['ms_per_step'],3), 'ms/matvec kernel', round(r['kernel_ms_avg'],3), 'ms')"
done
PASSES=min MODEL=heisenberg_chain_32 DTYPE=f64 TAG=r2_chain32_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
PASSES=min MODEL=heisenberg_chain_36_symm DTYPE=f64 TAG=r2_chain36symm_f64 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
PASSES=min MODEL=heisenberg_chain_32 DTYPE=c128 TAG=r2_chain32_c128 bash scripts/gpu_pmc_traffic.sh > /dev/null 2>&1
cat gpurun_out/r2_chain*/pmc_traffic_entry.json | grep -E "traffic_bytes|heisenberg|source_sha"

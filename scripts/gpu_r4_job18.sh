# GPU job 18 (round 4): prefix slot cache (one-partition plans keep the streams of as many rows as fit): parity, the 40-site
# Lanczos / Bethe test with it, timing of a 50 % prefix on chain_40_symm
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job18; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py -m gpu -q -x -k "slot_cache or replicated_exchange_indexed" > $OUT/pytest_focus.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_focus.log
( time timeout 900 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q -x -k "chain_40_symm_properties or bethe" > $OUT/pytest_40.log 2>&1 ) 2>&1 | grep real; tail -3 $OUT/pytest_40.log
python - <<'PY' 2>&1 | tail -4 | tee gpurun_out/r4job18/prefix_cache_chain40symm.txt
import time, torch
import distributed_matvec_amd as D
from distributed_matvec_amd import config
basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(40, symm=True), hamiltonian=True)
reps, masks = D.enumerateStates(basis, 1)
x = [D.fillRandom(reps[0], 42, torch.float64)]; y = [torch.zeros_like(x[0])]
n = int(reps[0].numel())
for budget in (0, 22 << 30, 44 << 30, 66 << 30, 100 << 30):
    pl = D.MatvecPlan(h, reps, torch.float64)
    rows = pl.cache_slots(budget) if budget else 0
    pl.matvec(x, y)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): pl.matvec(x, y, check=False)
    pl.check(); torch.cuda.synchronize()
    print(f"chain_40_symm budget {budget / 2**30:.0f} GiB: cached rows {rows} ({rows / n:.2f}), {pl.slot_cache[1] / 1e9:.1f} GB, {(time.perf_counter() - t) / 3 * 1e3:.1f} ms per matvec", flush=True)
    pl.destroy(); torch.cuda.empty_cache()
PY

#!/usr/bin/env python3
"""One lattice model of tests/golden/models.json (e.g. the reference's benchmark model heisenberg_square_6x6) on one GPU:
enumeration time, matvec time of the default plan, a JSON line.   usage: lattice_bench.py <model> [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "heisenberg_square_6x6"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"][name]["config"]
basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
torch.cuda.synchronize()
t = time.perf_counter()
reps, masks = D.enumerateStates(basis, 1)
torch.cuda.synchronize()
t_enum = time.perf_counter() - t
n = int(reps[0].numel())
x = [D.fillRandom(reps[0], 42, torch.float64)]
y = [torch.zeros_like(x[0])]
t = time.perf_counter()
pl = D.MatvecPlan(h, reps, torch.float64)
torch.cuda.synchronize()
t_plan = time.perf_counter() - t
pl.enable_timing(256)
for _ in range(2):
    pl.matvec(x, y, check=False)
pl.check()
pl.kernel_times_ms()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    pl.matvec(x, y, check=False)
pl.check()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
ks = pl.kernel_times_ms()
pp = D.MatvecPlan(h, reps, torch.float64, mode="push")
nnz = pp.nnz
pp.destroy()
# the same plan with the slot cache (opt-in, not matrix-free): resolve once, then only gather
cached = None
if pl.cache_slots(0) > 0:
    pl.matvec(x, y)
    pl.kernel_times_ms()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        pl.matvec(x, y, check=False)
    pl.check()
    torch.cuda.synchronize()
    cached = {"ms_per_matvec": (time.perf_counter() - t) / steps * 1e3, "hbm_bytes": pl.slot_cache[1], "kernel": pl.kernel}
print(json.dumps({"model": name, "sites": basis.numberSites(), "group_order": basis.groupOrder(), "spin_inversion": basis.spinInversion(),
                  "states": n, "nnz": nnz, "enumerate_s": t_enum, "plan_s": t_plan, "kernel": pl.kernel, "ms_per_matvec": dt * 1e3,
                  "kernel_ms_avg": sum(ks) / max(1, len(ks)), "matvecs_per_s": 1.0 / dt, "gnnz_per_s": nnz / dt / 1e9,
                  "slot_cache": cached,
                  "k4_env": {k: os.environ[k] for k in ("LS_AMD_K4",) if k in os.environ}}), flush=True)

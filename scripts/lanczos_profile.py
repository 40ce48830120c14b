#!/usr/bin/env python3
"""Where the time of the device-resident eigensolver goes: matvec vs orthogonalisation, with and without the slot cache.
usage: lanczos_profile.py L [max_basis]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402
from distributed_matvec_amd.diagonalize import LocalOperator, lanczos_smallest  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 36
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(L, symm=True), hamiltonian=True)
reps, masks = D.enumerateStates(basis, 1)
n = int(reps[0].numel())
for cache in (0, 1):
    free, _ = torch.cuda.mem_get_info()
    budget = max(0, int(free) - (mb + 6) * n * 8 - (8 << 30)) if cache else 0
    op = LocalOperator(h, reps, torch.float64, slot_cache_bytes=budget)
    t_mv = [0.0]
    inner = op.matvec

    def timed(x, y, inner=inner):
        torch.cuda.synchronize()
        t = time.perf_counter()
        inner(x, y)
        torch.cuda.synchronize()
        t_mv[0] += time.perf_counter() - t

    op.matvec = timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = lanczos_smallest(op, num_evals=1, eps=1e-7, max_basis=mb, max_restarts=200)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"chain_{L}_symm max_basis={mb} cache_rows={op.cached_rows}: E0={res.eigenvalues[0]:.10f} matvecs={res.matvecs} total {total:.2f} s, "
          f"matvec {t_mv[0]:.2f} s ({1e3 * t_mv[0] / res.matvecs:.1f} ms each), rest {total - t_mv[0]:.2f} s ({1e3 * (total - t_mv[0]) / res.matvecs:.1f} ms per step)", flush=True)
    op.plan.destroy()
    del op, res
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Per-rank compute cost of the replicated-x mode with block rows (what ReplicatedOperator runs), emulating
rank P // 2 of P on one device: no communication, kernel time only."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32)
ap.add_argument("--P", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(args.L), hamiltonian=True)
reps, masks = D.enumerateStates(basis, 1)
reps_global = reps[0]
n = reps_global.numel()
xg = D.fillRandom(reps_global, 42, torch.float64)
for P in args.P:
    p = P // 2
    n0, n1 = n * p // P, n * (p + 1) // P
    pl = D.ReplicatedPlan(h, reps_global[n0:n1], reps_global, torch.float64, P, p)
    y = torch.zeros(n1 - n0, dtype=torch.float64, device="cuda")
    pl.enable_timing(1024)
    for _ in range(args.steps + 2):
        pl.matvec(xg, y)
    torch.cuda.synchronize()
    ms = pl.kernel_times_ms()[2:]
    print(f"P={P} rank {p}: rows {n1 - n0}, kernel {pl.kernel}: {sum(ms) / len(ms):.3f} ms per launch", flush=True)
    del pl, y

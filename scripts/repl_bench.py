#!/usr/bin/env python3
"""Exploration: per-rank compute cost of the replicated-x mode, emulating rank 0 of P on one device
(no communication): plan creation, the gather permutation pass, the pull kernel."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32)
ap.add_argument("--symm", action="store_true")
ap.add_argument("--P", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(args.L, symm=args.symm), hamiltonian=True)
for P in args.P:
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    my = reps[0].clone()
    del reps
    torch.cuda.empty_cache()
    xg = D.fillRandom(reps_global, 42, torch.float64)
    y = torch.zeros(my.numel(), dtype=torch.float64, device="cuda")
    t = time.perf_counter()
    pl = D.ReplicatedPlan(h, my, reps_global, torch.float64, P, 0)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t
    pl.enable_timing(1024)
    pl.matvec(xg, y)
    pl.kernel_times_ms()
    perm = torch.randperm(xg.numel(), device="cuda").to(torch.int32)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.steps):
        torch.index_select(xg, 0, perm)
    torch.cuda.synchronize()
    t_perm = (time.perf_counter() - t) / args.steps
    t = time.perf_counter()
    for _ in range(args.steps):
        pl.matvec(xg, y, check=False)
    pl.check()
    dt = (time.perf_counter() - t) / args.steps
    ks = pl.kernel_times_ms()
    print(f"L={args.L} symm={args.symm} P={P} rows={my.numel()} kernel={pl.kernel} plan={t_plan:.3f}s "
          f"matvec={dt*1e3:.3f}ms kernel={sum(ks)/len(ks):.3f}ms random-permute-of-x={t_perm*1e3:.3f}ms", flush=True)
    del pl, xg, y, perm, reps_global, my
    torch.cuda.empty_cache()

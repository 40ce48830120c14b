#!/usr/bin/env python3
"""Exploration bench (not the contract bench): times enumeration, plan creation and the matvec of any
chain config with P logical partitions on ONE device (the in-process ls_amd_matvec path)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=28)
ap.add_argument("--symm", action="store_true")
ap.add_argument("--P", type=int, default=1)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--mode", default="auto")
ap.add_argument("--tree", action="store_true", help="per-stage timing tree (HIP events around every stage)")
ap.add_argument("--cache", action="store_true", help="slot cache (ls_amd_plan_cache_slots): time the gather-only matvec")
args = ap.parse_args()

cfg = config.heisenberg_chain_config(args.L, symm=args.symm)
basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
torch.cuda.synchronize()
t = time.perf_counter()
reps, masks = D.enumerateStates(basis, args.P)
torch.cuda.synchronize()
t_enum = time.perf_counter() - t
n = int(masks.numel())
td = torch.float64 if args.dtype == "f64" else torch.complex128
x = [D.fillRandom(r, 42, td) for r in reps]
y = [torch.zeros_like(v) for v in x]
t = time.perf_counter()
pl = D.MatvecPlan(h, reps, td, mode=args.mode)
torch.cuda.synchronize()
t_plan = time.perf_counter() - t
if args.cache:
    assert pl.cache_slots(0) > 0, "nothing to cache for this plan"
pl.enable_timing(4096)
if args.tree:
    pl.enable_stage_timing(65536)
pl.matvec(x, y)
pl.kernel_times_ms()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(args.steps):
    pl.matvec(x, y, check=False)
pl.check()
dt = (time.perf_counter() - t) / args.steps
ks = pl.kernel_times_ms()
print(f"L={args.L} symm={args.symm} P={args.P} {args.dtype} N={n} kernel={pl.kernel} rounds={pl.num_rounds} nnz={pl.nnz} "
      f"enum={t_enum:.3f}s plan={t_plan:.3f}s matvec={dt*1e3:.3f}ms ({1/dt:.2f}/s) dominant-kernel-total={sum(ks)/args.steps:.3f}ms "
      f"launches/step={len(ks)//args.steps}", flush=True)
if args.tree:
    print(pl.timing_report(), flush=True)

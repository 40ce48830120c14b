#!/usr/bin/env python3
"""Exploration (not the contract bench): one process, one basis, many tile orders of the staged row kernel.
The tile map is plan data (lsk_tilemap), read from the environment at plan creation, so every configuration is a
fresh plan on the same sigma / x / y.  Prints one line per configuration: kernel ms (HIP events), max |dy| vs the
default order (parity of the order itself)."""
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
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--configs", default="")
args = ap.parse_args()

KEYS = ("LS_AMD_TILE_CHUNK", "LS_AMD_ROW_KERNEL")
# the first configuration of a process runs 5-9 % faster than the later ones (clock / power state): compare variants in
# separate processes (--configs ";" = one default run), or read the trailing {} against the leading one
DEFAULT_CONFIGS = [
    {},
    {"LS_AMD_TILE_CHUNK": "32"},
    {"LS_AMD_TILE_CHUNK": "512"},
    {"LS_AMD_ROW_KERNEL": "pairs"},
    {"LS_AMD_ROW_KERNEL": "generic"},
    {},
]
configs = DEFAULT_CONFIGS
if args.configs:
    configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in args.configs.split(";")]

cfg = config.heisenberg_chain_config(args.L)
basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
reps, masks = D.enumerateStates(basis, 1)
td = torch.float64 if args.dtype == "f64" else torch.complex128
x = [D.fillRandom(reps[0], 42, td)]
y = [torch.zeros_like(x[0])]
y_ref = None
for c in configs:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(c)
    t = time.perf_counter()
    pl = D.MatvecPlan(h, reps, td)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t
    pl.enable_timing(4096)
    pl.matvec(x, y)
    pl.matvec(x, y)
    pl.kernel_times_ms()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.steps):
        pl.matvec(x, y, check=False)
    pl.check()
    dt = (time.perf_counter() - t) / args.steps
    ks = pl.kernel_times_ms()
    if y_ref is None:
        y_ref = y[0].clone()
        err = 0.0
    else:
        err = float((y[0] - y_ref).abs().max())
    print(f"{pl.kernel} {c} plan={t_plan:.2f}s wall={dt * 1e3:.3f}ms kernel_avg={sum(ks) / max(1, len(ks)):.3f}ms "
          f"min={min(ks):.3f} max|dy|={err:.1e}", flush=True)
    pl.destroy()
    del pl

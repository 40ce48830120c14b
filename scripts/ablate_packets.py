#!/usr/bin/env python3
"""Profiling (never shipped): where the time of the packet producer of the projected bases (k_tile_wv) goes.  Loads the ABLATE build
(make -C distributed-matvec_amd/csrc ablate) and times the producers / consumers of chain_L_symm over P logical partitions with
stages switched off through LS_AMD_ABLATE: 256 no K4, 512 no packet stores, 1024 own-partition packets dropped, 2048 no hash.
Results are WRONG by construction; only the times mean anything.   usage: ablate_packets.py [L] [P] [masks...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import _lib, config  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "distributed-matvec_amd", "libls_amd_ablate.so")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 36
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
masks_ = [int(a) for a in sys.argv[3:]] or [0, 256, 512, 1024, 2048, 256 + 512 + 1024 + 2048, 0]
for m in masks_:
    os.environ["LS_AMD_ABLATE"] = str(m)
    basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(L, symm=True), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    x = [D.fillRandom(r, 42, torch.float64) for r in reps]
    y = [torch.zeros_like(v) for v in x]
    pl = D.MatvecPlan(h, reps, torch.float64)
    for _ in range(2):
        pl.matvec(x, y, check=False)
    pl.enable_stage_timing() if hasattr(pl, "enable_stage_timing") else None
    torch.cuda.synchronize()
    import time

    t = time.perf_counter()
    for _ in range(3):
        pl.matvec(x, y, check=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"chain_{L}_symm x {P} ablate={m:5d} kernel {pl.kernel}: {dt * 1e3:8.3f} ms per matvec", flush=True)
    pl.destroy()
    del pl, h, basis, reps, x, y
    torch.cuda.empty_cache()

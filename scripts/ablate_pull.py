#!/usr/bin/env python3
"""Profiling (never shipped): where the time of the indexed pull kernel of the projected bases goes.  Loads the ABLATE build of
the library (make -C distributed-matvec_amd/csrc ablate -> libls_amd_ablate.so) and times k_pull_t with stages switched
off through LS_AMD_ABLATE: 1 stage A only, 4 no K4, 2 K4 but no look-ups / accumulation, 32 no near window, 64 no value load.
Results are WRONG by construction; only the times mean anything.   usage: ablate_pull.py heisenberg_chain_36_symm [masks...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import _lib, config  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "distributed-matvec_amd", "libls_amd_ablate.so")
name = sys.argv[1] if len(sys.argv) > 1 else "heisenberg_chain_36_symm"
masks_ = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 4, 32, 64, 0]
if name.startswith("heisenberg_chain_"):
    L = int(name.split("_")[2])
    cfg = config.heisenberg_chain_config(L, symm=name.endswith("_symm"))
else:  # a lattice model of tests/golden/models.json (heisenberg_square_6x6, ...)
    import json

    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"][name]["config"]
for m in masks_:
    os.environ["LS_AMD_ABLATE"] = str(m)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)  # the ablation mask is read when the basis goes to the device
    reps, masks = D.enumerateStates(basis, 1)
    x = [D.fillRandom(reps[0], 42, torch.float64)]
    y = [torch.zeros_like(x[0])]
    pl = D.MatvecPlan(h, reps, torch.float64)
    pl.enable_timing(64)
    for _ in range(2):
        pl.matvec(x, y, check=False)
    pl.kernel_times_ms()
    for _ in range(4):
        pl.matvec(x, y, check=False)
    ks = pl.kernel_times_ms()
    print(f"{name} ablate={m:3d} kernel {pl.kernel}: avg {sum(ks) / len(ks):8.3f} ms  min {min(ks):8.3f} ms", flush=True)
    pl.destroy()
    del pl, h, basis, reps, x, y
    torch.cuda.empty_cache()

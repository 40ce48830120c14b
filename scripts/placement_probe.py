#!/usr/bin/env python3
"""Why is the first plan of a process ~5-9 % faster than later ones on the same vectors?  Probe: shift the addresses the
plan's hipMalloc calls return by holding pads of various sizes, and time the staged kernel for each."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import _lib, config  # noqa: E402

L = _lib.load()
basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(32), hamiltonian=True)
reps, masks = D.enumerateStates(basis, 1)
x = [D.fillRandom(reps[0], 42, torch.float64)]
y = [torch.zeros_like(x[0])]
print("x", hex(x[0].data_ptr()), "y", hex(y[0].data_ptr()), "reps", hex(reps[0].data_ptr()), flush=True)


def run(tag):
    pl = D.MatvecPlan(h, reps, torch.float64)
    pl.enable_timing(256)
    for _ in range(3):
        pl.matvec(x, y, check=False)
    pl.check()
    pl.kernel_times_ms()
    for _ in range(8):
        pl.matvec(x, y, check=False)
    pl.check()
    ks = pl.kernel_times_ms()
    print(f"{tag}: kernel_avg={sum(ks) / len(ks):.3f} ms min={min(ks):.3f}", flush=True)
    pl.destroy()


run("first plan")
run("second plan (same process)")
for pad in (0, 1 << 21, 1 << 25, 1 << 28, 1 << 30, 3 << 30, (1 << 32) + (1 << 21)):
    p = C.c_void_p()
    if pad:
        _lib.check(L.ls_amd_malloc(C.byref(p), pad))
    run(f"pad {pad >> 20} MiB at {hex(p.value or 0)}")
    if pad:
        L.ls_amd_free(p)
# a fresh pair of vectors allocated AFTER everything else
x2 = [x[0].clone()]
y2 = [torch.zeros_like(x2[0])]
x, y = x2, y2
print("x2", hex(x[0].data_ptr()), "y2", hex(y[0].data_ptr()), flush=True)
run("fresh vectors")

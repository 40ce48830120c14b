#!/usr/bin/env python3
"""ls_amd_orth_pass alone: achieved HBM rate of the two sweeps (pass 1: read m + 1 vectors; pass 2: read m + 1, write 1) next to
the torch.mv form they replace.  usage: orth_bench.py [n] [m]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from distributed_matvec_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 861725794
m = int(sys.argv[2]) if len(sys.argv) > 2 else 9
lib = _lib.load()
V = torch.randn((m, n), dtype=torch.float64, device="cuda")
w = torch.randn(n, dtype=torch.float64, device="cuda")
h = torch.zeros(m, dtype=torch.float64, device="cuda")
out = torch.zeros(m + 1, dtype=torch.float64, device="cuda")


def run(hin):
    assert lib.ls_amd_orth_pass(m, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(w.data_ptr()),
                                C.c_void_p(hin.data_ptr()) if hin is not None else None, C.c_void_p(out.data_ptr()), None) == 0


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


t1 = timed(lambda: run(None))
t2 = timed(lambda: run(h))
b1, b2 = (m + 1) * n * 8, (m + 2) * n * 8
print(f"n={n} m={m}: pass 1 {t1 * 1e3:.2f} ms = {b1 / t1 / 1e12:.2f} TB/s, pass 2 {t2 * 1e3:.2f} ms = {b2 / t2 / 1e12:.2f} TB/s (HBM bytes; the block is read twice from cache level)")
ta = timed(lambda: torch.mv(V, w))
tb = timed(lambda: w.sub_(torch.mv(V.t(), h)))
print(f"torch: mv {ta * 1e3:.2f} ms = {b1 / ta / 1e12:.2f} TB/s, update {tb * 1e3:.2f} ms = {b2 / tb / 1e12:.2f} TB/s; CGS twice = {2 * (ta + tb) * 1e3:.1f} ms vs fused two sweeps {(t1 + t2) * 1e3:.1f} ms")

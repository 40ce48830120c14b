#!/usr/bin/env python3
"""Attainable streaming rates of the box: copy (read + write) and read-only, 16 bytes per lane, over a 4.8 GB vector."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_matvec_amd import _lib  # noqa: E402

L = _lib.load()
n = 601080390
x = torch.rand(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
nb = n * 8
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


t = timed(lambda: _lib.check(L.ls_amd_stream_copy(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), nb, st)))
print(f"copy       : {2 * nb / t / 1e9:8.1f} GB/s (read + write), {t * 1e3:.3f} ms")
for per in (1, 2, 4, 8, 16):
    t = timed(lambda: _lib.check(L.ls_amd_stream_read(C.c_void_p(x.data_ptr()), nb, per, C.c_void_p(sink.data_ptr()), st)))
    print(f"read x{per:<2d}   : {nb / t / 1e9:8.1f} GB/s, {t * 1e3:.3f} ms")

#!/usr/bin/env python3
"""K4 (orbit minimum, trivial sector, dihedral fast path) priced alone: every representative of a symmetric ring with each adjacent
pair flipped (L packets per row, every lane busy), variants: 2 = the packets only, 1 = the two run searches, 0 = the whole thing.
Prints ns per packet-lane and what the chain's real packet count (half the pairs anti-aligned) would cost at that rate."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import _lib, config  # noqa: E402

L = _lib.load()
for name in sys.argv[1:] or ["heisenberg_chain_36_symm"]:
    sites = int(name.split("_")[2])
    basis = D.loadConfigFromDict(config.heisenberg_chain_config(sites, symm=True))
    reps, masks = D.enumerateStates(basis, 1)
    r = reps[0] if isinstance(reps, (list, tuple)) else reps
    n = int(r.numel())
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for variant in (2, 1, 0):
        def run():
            _lib.check(L.ls_amd_bench_k4(sites, 1, 1, variant, n, C.c_void_p(r.data_ptr()), C.c_void_p(out.data_ptr()), st))
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        pk = n * sites
        print(f"{name} variant {variant}: {ms:8.3f} ms for {pk:.3e} packets = {ms * 1e6 / pk * 1e3:.2f} ps per packet; "
              f"the matvec's {pk // 2:.3e} packets at that rate: {ms / 2:.2f} ms", flush=True)

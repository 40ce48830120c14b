#!/usr/bin/env python3
"""What a byte per row costs the headline kernel (profiling build, never shipped): k_chain_t on heisenberg_chain_<L> with and without
ONE MORE 8-byte stream per row (LS_AMD_ABLATE & 128, make -C distributed-matvec_amd/csrc ablate), prefetched exactly like the 8-byte
sigma | partner records.  The difference bounds what computing sigma (and the ring partner) instead of loading them could save:
removing a stream cannot gain more than adding the same stream costs.   usage: chain_stream_cost.py [L] [steps]"""
import os
import sys

os.environ["LS_AMD_LIB"] = "libls_amd_ablate.so"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = config.heisenberg_chain_config(L)
rows = {}
y_ref = None
for mask in [int(m) for m in os.environ.get("LS_AMD_ABLATE_MASKS", "0,128,0,128").split(",")]:
    os.environ["LS_AMD_ABLATE"] = str(mask)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)  # the mask is read when the basis goes to the device
    reps, masks = D.enumerateStates(basis, 1)
    x = [D.fillRandom(reps[0], 42, torch.float64)]
    y = [torch.zeros_like(x[0])]
    pl = D.MatvecPlan(h, reps, torch.float64)
    pl.enable_timing(256)
    for _ in range(3):
        pl.matvec(x, y, check=False)
    pl.check()
    pl.kernel_times_ms()
    for _ in range(steps):
        pl.matvec(x, y, check=False)
    pl.check()
    ks = pl.kernel_times_ms()
    n = int(reps[0].numel())
    if y_ref is None:
        y_ref = y[0].clone()
    same = bool(torch.equal(y[0], y_ref))
    rows.setdefault(mask, []).append(sum(ks) / len(ks))
    print(f"chain_{L} ablate={mask:3d} kernel {pl.kernel}: avg {sum(ks) / len(ks):7.3f} ms  min {min(ks):7.3f} ms  y == first run: {same}", flush=True)
    pl.destroy()
    del x, y, reps
    torch.cuda.empty_cache()
if 128 not in rows:  # other masks (LS_AMD_ABLATE_MASKS): the table above is the result
    sys.exit(0)
base, extra = min(rows[0]), min(rows[128])
n_bytes = 8 * n
print(f"one more 8-byte stream per row ({n_bytes / 1e9:.2f} GB): {base:.3f} -> {extra:.3f} ms = +{extra - base:.3f} ms, i.e. {1e3 * (extra - base) / 8:.1f} us per byte per row; "
      f"removing the 8-byte record stream can gain at most as much: >= {base - (extra - base):.3f} ms = {1e3 / (base - (extra - base)):.1f} matvec/s "
      f"(4 bytes -- sigma alone: {1e3 / (base - (extra - base) / 2):.1f})")

#!/usr/bin/env python3
"""Models the staged CHAIN kernel does not take -- the staged kernel for arbitrary exchange pairs (k_pairs_t, round 4) against the
generic row kernel (k_direct) and the push formulation: a periodic
square lattice (Lx x Ly sites, half filling, no symmetries) and the J1-J2 ring, at sizes that leave every cache.
Prints one JSON line per (model, mode): ms per matvec (HIP events), matvec/s, non-zeros, compulsory bytes and rates."""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402


def heis(sites):
    return {"name": "H", "terms": [{"expression": e, "sites": sites} for e in ("σˣ₀ σˣ₁", "σʸ₀ σʸ₁", "σᶻ₀ σᶻ₁")]}


def square(lx, ly):
    idx = lambda x, y: (x % lx) + lx * (y % ly)  # noqa: E731
    bonds = []
    for y in range(ly):
        for x in range(lx):
            bonds += [[idx(x, y), idx(x + 1, y)], [idx(x, y), idx(x, y + 1)]]
    return {"basis": {"number_spins": lx * ly, "hamming_weight": lx * ly // 2, "symmetries": []}, "hamiltonian": heis(bonds)}, len(bonds)


def j1j2(L):
    bonds = [[i, (i + 1) % L] for i in range(L)] + [[i, (i + 2) % L] for i in range(L)]
    return {"basis": {"number_spins": L, "hamming_weight": L // 2, "symmetries": []}, "hamiltonian": heis(bonds)}, len(bonds)


ap = argparse.ArgumentParser()
ap.add_argument("--models", default="square_6x5,j1j2_30,j1j2_32")
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
for name in args.models.split(","):
    kind, size = name.split("_")
    if kind == "square":
        lx, ly = (int(v) for v in size.split("x"))
        cfg, nb = square(lx, ly)
    else:
        cfg, nb = j1j2(int(size))
    L = cfg["basis"]["number_spins"]
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    n = int(masks.numel())
    nnz = n * nb * L // (2 * (L - 1))  # every bond is anti-aligned in a fraction L / (2 (L - 1)) of the half-filling states
    x = [D.fillRandom(reps[0], 42, torch.float64)]
    y = [torch.zeros_like(x[0])]
    y_ref = None
    for mode, rk in (("pull", "auto"), ("pull", "generic"), ("push", "auto")):
        os.environ["LS_AMD_ROW_KERNEL"] = rk  # auto: the staged kernel for arbitrary exchange pairs; generic: k_direct
        pl = D.MatvecPlan(h, reps, torch.float64, mode=mode)
        pl.enable_timing(256)
        pl.matvec(x, y)
        pl.matvec(x, y)
        pl.kernel_times_ms()
        for _ in range(args.steps):
            pl.matvec(x, y, check=False)
        pl.check()
        ks = pl.kernel_times_ms()
        ms = sum(ks) / len(ks)
        if y_ref is None:
            y_ref = y[0].clone()
        err = float((y[0] - y_ref).abs().max() / y_ref.abs().max())
        alg = n * (pl.row_bytes + 16) if mode == "pull" else n * 16 + nnz * 16
        print(json.dumps({"model": name, "sites": L, "bonds": nb, "flip_mask_groups": h.numberOffDiagTerms(), "states": n, "nnz": nnz,
                          "mode": mode, "kernel": pl.kernel, "kernel_ms": ms, "matvecs_per_s": 1e3 / ms, "gnnz_per_s": nnz / ms / 1e6,
                          "algorithmic_GB": alg / 1e9, "algorithmic_GBps": alg / ms / 1e6,
                          "max_rel_diff_vs_first": err}), flush=True)
        pl.destroy()
    del x, y, reps, masks
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Round 6, VERDICT r5 #6: what the operators OUTSIDE the round-5 fast paths get now.
  chain_<L>_inv<s>   heisenberg ring in a spin-inversion sector without permutations (staged chain kernel vs generic row kernel vs push)
  hop_<L>            NON-Hermitian ring: sigma^+_i sigma^-_{i+1} (one direction) + zz (pull of a non-Hermitian operator vs push)
  square_<X>x<Y>_w<k> unsymmetrised periodic square lattice at weight k (> 32 sites: off the 32-bit pairs kernel)
  j1j2_<L>_w<k>       ring with first and second neighbours at weight k
Prints one JSON line per (model, variant): kernel, ms per matvec (HIP events inside the library), non-zeros (the push plan's count
pass), G non-zeros / s, max relative difference against the first variant."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402


def heis(sites):
    return {"name": "H", "terms": [{"expression": e, "sites": sites} for e in ("σˣ₀ σˣ₁", "σʸ₀ σʸ₁", "σᶻ₀ σᶻ₁")]}


def model(name):
    parts = name.split("_")
    if parts[0] == "chain":
        L, s = int(parts[1]), int(parts[2].replace("inv", ""))
        bonds = [[i, (i + 1) % L] for i in range(L)]
        return {"basis": {"number_spins": L, "hamming_weight": L // 2, "spin_inversion": s, "symmetries": []}, "hamiltonian": heis(bonds)}
    if parts[0] == "hop":
        L = int(parts[1])
        bonds = [[i, (i + 1) % L] for i in range(L)]
        return {"basis": {"number_spins": L, "hamming_weight": L // 2, "symmetries": []},
                "hamiltonian": {"name": "H", "terms": [{"expression": "σ⁺₀ σ⁻₁", "sites": bonds}, {"expression": "σᶻ₀ σᶻ₁", "sites": bonds}]}}
    if parts[0] == "j1j2":  # ring with first and second neighbours at weight k
        L, k = int(parts[1]), int(parts[2][1:])
        bonds = [[i, (i + 1) % L] for i in range(L)] + [[i, (i + 2) % L] for i in range(L)]
        return {"basis": {"number_spins": L, "hamming_weight": k, "symmetries": []}, "hamiltonian": heis(bonds)}
    lx, ly = (int(v) for v in parts[1].split("x"))
    k = int(parts[2][1:])
    idx = lambda x, y: (x % lx) + lx * (y % ly)  # noqa: E731
    bonds = []
    for y in range(ly):
        for x in range(lx):
            bonds += [[idx(x, y), idx(x + 1, y)], [idx(x, y), idx(x, y + 1)]]
    return {"basis": {"number_spins": lx * ly, "hamming_weight": k, "symmetries": []}, "hamiltonian": heis(bonds)}


ap = argparse.ArgumentParser()
ap.add_argument("--models", default="chain_32_inv1,hop_30,square_6x6_w6")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--variants", default="push:auto,auto:auto,pull:generic", help="mode:LS_AMD_ROW_KERNEL, comma-separated")
args = ap.parse_args()
for name in args.models.split(","):
    cfg = model(name)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    n = int(masks.numel())
    x = [D.fillRandom(reps[0], 42, torch.float64)]
    y = [torch.zeros_like(x[0])]
    y_ref, nnz = None, None
    for mode, rk in (v.split(":") for v in args.variants.split(",")):
        os.environ["LS_AMD_ROW_KERNEL"] = rk
        try:
            pl = D.MatvecPlan(h, reps, torch.float64, mode=mode)
        except D.LsAmdError as e:
            print(json.dumps({"model": name, "mode": mode, "row_kernel": rk, "error": str(e)[:200]}), flush=True)
            continue
        if mode == "push":
            nnz = pl.nnz
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
        print(json.dumps({"model": name, "sites": basis.numberSites(), "states": n, "hermitian": bool(h.isHermitian), "nnz": nnz, "mode": mode, "row_kernel": rk,
                          "kernel": pl.kernel, "kernel_ms": ms, "gnnz_per_s": (nnz or 0) / ms / 1e6, "max_rel_diff_vs_push": err}), flush=True)
        pl.destroy()
    del x, y, reps, masks
    torch.cuda.empty_cache()

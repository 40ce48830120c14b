"""Host mirror of ``ls_hs_load_yaml_config`` (/root/reference/src/ForeignTypes.chpl:261-288,
/root/reference/src/FFI.chpl:205): YAML -> basis description + non-branching terms.

The reference delegates this to lattice-symmetries-haskell; here the (new-schema) YAML subset of
/root/reference/data/*.yaml is compiled symbolically: every expression is a
monomial of single-site operators, each of which maps a basis state to at most one basis state, so a
monomial applied to a site tuple yields a handful of terms
    coefficient v, projector (m, r), flip mask x, sign mask s
(include/ls_hs.h).  Merging / cancellation / grouping by flip mask happens in C
(ls_hs_create_operator_from_terms).  No numerics on the hot path happen here.

Conventions: site i <-> bit i; bit 0 = spin up (sigma^z = +1); S^a = sigma^a / 2.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field

_SUPER = {"ˣ": "x", "ʸ": "y", "ᶻ": "z", "⁺": "+", "⁻": "-"}
_SUB = {chr(0x2080 + d): d for d in range(10)}

# single-site operators as {input bit: (output bit, coefficient)}
_SITE_OPS = {
    "x": {0: (1, 1.0 + 0j), 1: (0, 1.0 + 0j)},
    "y": {0: (1, 1j), 1: (0, -1j)},
    "z": {0: (0, 1.0 + 0j), 1: (1, -1.0 + 0j)},
    "+": {1: (0, 1.0 + 0j)},  # |up><down|
    "-": {0: (1, 1.0 + 0j)},
    "I": {0: (0, 1.0 + 0j), 1: (1, 1.0 + 0j)},
}


def parse_expression(expr: str):
    """'0.8 × σˣ₀ σˣ₁' -> (0.8+0j, [('x', 0, 1.0), ('x', 1, 1.0)])."""
    scalar = 1.0 + 0j
    factors = []
    for tok in expr.replace("×", " ").replace("*", " ").split():
        if tok[0] in ("σ", "S"):
            pref = 1.0 if tok[0] == "σ" else 0.5
            if len(tok) < 3 or tok[1] not in _SUPER:
                raise ValueError(f"cannot parse operator {tok!r} in {expr!r}")
            idx = 0
            for ch in tok[2:]:
                if ch not in _SUB:
                    raise ValueError(f"cannot parse site index in {tok!r}")
                idx = idx * 10 + _SUB[ch]
            factors.append((_SUPER[tok[1]], idx, pref))
        else:
            scalar *= complex(tok)
    if not factors:
        raise ValueError(f"expression {expr!r} has no operators")
    return scalar, factors


def _compose(first, second):
    """site operator `second` applied after `first`."""
    out = {}
    for b, (a, c) in first.items():
        if a in second:
            a2, c2 = second[a]
            out[b] = (a2, c * c2)
    return out


def _site_alternatives(op):
    """A single-site non-branching operator as mutually exclusive alternatives
    (needs_projector, r_bit, flip, sign, coeff): Pauli-like operators need no projector."""
    if 0 in op and 1 in op:
        (a0, c0), (a1, c1) = op[0], op[1]
        flip0, flip1 = a0 != 0, a1 != 1
        if flip0 == flip1 and c1 == c0:
            return [(False, 0, flip0, False, c0)]
        if flip0 == flip1 and c1 == -c0:
            return [(False, 0, flip0, True, c0)]
    alts = []
    for b, (a, c) in op.items():
        alts.append((True, b, a != b, False, c))
    return alts


def monomial_terms(expr: str, sites):
    """Terms (v, m, r, x, s) of one monomial on one tuple of global site indices."""
    scalar, factors = parse_expression(expr)
    per_site = {}
    order = []
    # operators written left-to-right act right-to-left on a ket
    for kind, idx, pref in reversed(factors):
        op = {b: (a, c * pref) for b, (a, c) in _SITE_OPS[kind].items()}
        if idx in per_site:
            per_site[idx] = _compose(per_site[idx], op)
        else:
            per_site[idx] = op
            order.append(idx)
    k = 1 + max(per_site)
    if len(sites) != k:
        raise ValueError(f"expression {expr!r} needs {k} sites, got {sites}")
    if len(set(sites)) != len(sites):
        raise ValueError(f"repeated site in {sites}")
    alts_per_site = [(idx, _site_alternatives(per_site[idx])) for idx in order]
    terms = []
    for combo in itertools.product(*[alts for _, alts in alts_per_site]):
        v = scalar
        m = r = x = s = 0
        for (idx, _), (need, rbit, flip, sign, c) in zip(alts_per_site, combo):
            bit = 1 << int(sites[idx])
            v *= c
            if need:
                m |= bit
                if rbit:
                    r |= bit
            if flip:
                x |= bit
            if sign:
                s |= bit
        if v != 0:
            terms.append((complex(v), m, r, x, s))
    return terms


@dataclass
class BasisSpec:
    number_sites: int
    hamming_weight: int = -1  # -1: unrestricted
    spin_inversion: int = 0
    permutations: list = field(default_factory=list)
    sectors: list = field(default_factory=list)


@dataclass
class OperatorSpec:
    terms: list  # [(v complex, m, r, x, s)]


def parse_basis(cfg: dict) -> BasisSpec:
    b = cfg["basis"]
    hw = b.get("hamming_weight", None)
    inv = b.get("spin_inversion", None)
    syms = b.get("symmetries", None) or []
    particle = b.get("particle", "spin-1/2")
    if particle != "spin-1/2":
        raise ValueError(f"only spin-1/2 bases are supported, got {particle!r}")
    return BasisSpec(
        number_sites=int(b["number_spins"]),
        hamming_weight=-1 if hw is None else int(hw),
        spin_inversion=0 if inv is None else int(inv),
        permutations=[[int(v) for v in s["permutation"]] for s in syms],
        sectors=[int(s["sector"]) for s in syms],
    )


def parse_operator(section: dict) -> OperatorSpec:
    terms = []
    for t in section["terms"]:
        if "expression" not in t:
            raise ValueError("only the `expression:` schema is supported (data/*.yaml); "
                             "old-schema `matrix:` files are inputs of input_for_matvec.py only")
        for sites in t["sites"]:
            terms.extend(monomial_terms(t["expression"], [int(q) for q in sites]))
    return OperatorSpec(terms)


def heisenberg_chain_config(L: int, symm: bool = False, spin_inversion=None) -> dict:
    """The reference's chain inputs, generated: identical content to
    /root/reference/data/heisenberg_chain_{L}[_symm].yaml (checked in tests when the reference is
    mounted).  Needed because nothing may read /root/reference at run time on the GPU box."""
    basis = {"number_spins": L, "hamming_weight": L // 2}
    if symm:
        basis["spin_inversion"] = 1
        basis["symmetries"] = [
            {"permutation": [(i + 1) % L for i in range(L)], "sector": 0},
            {"permutation": [L - 1 - i for i in range(L)], "sector": 0},
        ]
    else:
        if spin_inversion is not None:
            basis["spin_inversion"] = spin_inversion
        basis["symmetries"] = []
    lattice = [[i, (i + 1) % L] for i in range(L)]
    terms = [{"expression": e, "sites": lattice} for e in ("σˣ₀ σˣ₁", "σʸ₀ σʸ₁", "σᶻ₀ σᶻ₁")]
    return {"basis": basis, "hamiltonian": {"name": "Heisenberg Hamiltonian", "terms": terms}}

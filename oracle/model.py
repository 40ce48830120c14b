"""CPU oracle, part 1 (numpy): model loading, term tables, symmetry groups, dense checks.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
(``distributed-matvec_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED (by reference artefacts): the reference's per-row arithmetic lives in the
un-vendored ``lattice-symmetries-haskell`` @ 14e7319 (/root/reference/.github/workflows/ci.yml:6,27-31)
and its golden HDF5 vectors are downloaded artefacts (/root/reference/Makefile:128-146), both
absent offline.  What *is* pinned, and checked in tests/test_oracle_model.py:
  * the explicit two-site matrices of the old YAML schema
    (/root/reference/data/old/heisenberg_chain_10.yaml:9-12) == what this module derives from the
    ``expression:`` strings of the new schema (/root/reference/data/heisenberg_chain_10.yaml:9-14);
  * the 13-representative fixture of /root/reference/v1/error.chpl:21;
  * the golden *inputs* recipe of /root/reference/input_for_matvec.py:8,31,49-75.
Two independent formulations are cross-checked here instead: (i) "non-branching term" tables
(what the external kernels ``ls_internal_operator_apply_{diag,off_diag}_x1`` consume,
/root/reference/src/FFI.chpl:219-225) and (ii) dense Kronecker-product matrices with an explicit
symmetry projector.

Conventions (unpinned by any in-tree config, all of which are invariant under them):
  * site i <-> bit i of a uint64 (/root/reference/src/BatchedOperator.chpl:140);
  * bit value 0 = spin up (sigma^z = +1), 1 = spin down;
  * a permutation p acts as (g.s)[i] = s[p[i]]  ("output bit i = input bit p_i").
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------------------
# Expression parsing: the subset of the new YAML schema used by /root/reference/data/*.yaml
# --------------------------------------------------------------------------------------

_SUPER = {"ˣ": "x", "ʸ": "y", "ᶻ": "z", "⁺": "+", "⁻": "-"}
_SUB = {chr(0x2080 + d): str(d) for d in range(10)}

_PAULI = {
    "x": np.array([[0, 1], [1, 0]], dtype=complex),
    "y": np.array([[0, -1j], [1j, 0]], dtype=complex),
    "z": np.array([[1, 0], [0, -1]], dtype=complex),
    "+": np.array([[0, 1], [0, 0]], dtype=complex),  # |up><down|
    "-": np.array([[0, 0], [1, 0]], dtype=complex),
}


def parse_expression(expr: str):
    """'0.8 x sigma^x_0 sigma^x_1' -> (0.8, [('x', 0, 1.0), ('x', 1, 1.0)]).

    Each factor is (pauli kind, local site index, prefactor) where prefactor is 1 for
    sigma and 1/2 for S.  Grammar = products only (all in-tree expressions are monomials:
    /root/reference/data/issue_01.yaml:12-23, heisenberg_kagome_12.yaml:28-33).
    """
    s = expr.replace("×", " ").replace("*", " ")
    toks = s.split()
    scalar = 1.0 + 0j
    factors = []
    for tok in toks:
        if tok[0] in ("σ", "S"):
            pref = 1.0 if tok[0] == "σ" else 0.5
            kind = _SUPER[tok[1]]
            idx = int("".join(_SUB[c] for c in tok[2:]))
            factors.append((kind, idx, pref))
        else:
            scalar *= complex(tok)
    return scalar, factors


def local_matrix(expr: str):
    """Dense 2^k x 2^k matrix of a monomial on its k local sites.

    Index convention: local pattern value = sum_q bit_q << q with q the *local* site index,
    and single-site vector index == bit value (0 = up).
    Returns (k, matrix).
    """
    scalar, factors = parse_expression(expr)
    k = 1 + max(idx for _, idx, _ in factors)
    per_site = [np.eye(2, dtype=complex) for _ in range(k)]
    for kind, idx, pref in factors:
        per_site[idx] = per_site[idx] @ (pref * _PAULI[kind])
    # kron with site 0 as the least significant index
    m = np.array([[1.0 + 0j]])
    for q in range(k):
        m = np.kron(per_site[q], m)
    return k, scalar * m


# --------------------------------------------------------------------------------------
# Non-branching terms  (semantics: SURVEY Appendix A; usage spec
# /root/reference/src/BatchedOperator.chpl:11-36)
#   term active on alpha iff (alpha & m) == r ; beta = alpha ^ x ;
#   coefficient = v * (-1)^{popcount(alpha & s)}
# --------------------------------------------------------------------------------------


@dataclass
class Terms:
    v: np.ndarray  # complex128 [n]
    m: np.ndarray  # uint64 [n]
    r: np.ndarray  # uint64 [n]
    x: np.ndarray  # uint64 [n]
    s: np.ndarray  # uint64 [n]

    def __len__(self):
        return len(self.v)

    @staticmethod
    def empty():
        z = np.zeros(0, dtype=np.uint64)
        return Terms(np.zeros(0, dtype=complex), z, z.copy(), z.copy(), z.copy())

    def select(self, mask):
        return Terms(self.v[mask], self.m[mask], self.r[mask], self.x[mask], self.s[mask])


def terms_from_local_matrices(entries, tol=0.0):
    """entries: iterable of (matrix 2^k x 2^k, sites tuple of k global site indices).

    Matrix-element decomposition: every nonzero <a|M|b> becomes one projector-style term
    (m = the k sites, r = pattern b, x = a ^ b, s = 0).  Terms with equal (m, r, x) are merged.
    """
    acc: dict[tuple[int, int, int], complex] = {}
    for mat, sites in entries:
        k = len(sites)
        assert mat.shape == (1 << k, 1 << k)
        mmask = 0
        for q in sites:
            mmask |= 1 << q
        for a in range(1 << k):
            for b in range(1 << k):
                val = mat[a, b]
                if val == 0:
                    continue
                rr = 0
                xx = 0
                for q, site in enumerate(sites):
                    if (b >> q) & 1:
                        rr |= 1 << site
                    if ((a ^ b) >> q) & 1:
                        xx |= 1 << site
                key = (mmask, rr, xx)
                acc[key] = acc.get(key, 0) + val
    keys = [k for k, val in acc.items() if abs(val) > tol]
    keys.sort(key=lambda t: (t[2], t[0], t[1]))
    n = len(keys)
    return Terms(
        np.array([acc[k] for k in keys], dtype=complex).reshape(n),
        np.array([k[0] for k in keys], dtype=np.uint64).reshape(n),
        np.array([k[1] for k in keys], dtype=np.uint64).reshape(n),
        np.array([k[2] for k in keys], dtype=np.uint64).reshape(n),
        np.zeros(n, dtype=np.uint64),
    )


# --------------------------------------------------------------------------------------
# Model (basis + hamiltonian) from a parsed YAML dict (Appendix C of SURVEY.md)
# --------------------------------------------------------------------------------------


@dataclass
class Group:
    """Closure of the YAML generators, with 1-D characters.  Spin inversion is NOT included
    here; it is kept as a separate Z2 factor (``Model.spin_inversion``)."""

    perms: np.ndarray  # int32 [order, L]; element 0 is the identity
    chars: np.ndarray  # complex128 [order]


def _perm_order(p):
    n = 1
    q = list(p)
    ident = list(range(len(p)))
    while q != ident:
        q = [q[i] for i in p]  # compose
        n += 1
    return n


def compose(p, q):
    """(p after q): state -> q applied first, then p, under (g.s)[i] = s[g[i]]:
    ((p.q).s)[i] = (q.s)[p[i]] = s[q[p[i]]]."""
    return tuple(q[i] for i in p)


def close_group(number_sites, generators, sectors):
    """BFS closure.  The character of a generator of order n in sector k is exp(-2 pi i k / n)
    ([upstream-memory] for the sign; real in every in-tree config)."""
    ident = tuple(range(number_sites))
    if not generators:
        return Group(np.array([ident], dtype=np.int32), np.ones(1, dtype=complex))
    gens = [tuple(int(v) for v in g) for g in generators]
    gchars = []
    for g, k in zip(gens, sectors):
        n = _perm_order(g)
        ph = np.exp(-2j * np.pi * (k % n) / n)
        # snap exact values
        ph = complex(round(ph.real, 15), round(ph.imag, 15))
        gchars.append(ph)
    elems = {ident: 1.0 + 0j}
    frontier = [ident]
    while frontier:
        nxt = []
        for e in frontier:
            for g, ch in zip(gens, gchars):
                ne = compose(g, e)
                nch = elems[e] * ch
                if ne in elems:
                    if abs(elems[ne] - nch) > 1e-9:
                        raise ValueError("sectors are incompatible with the group structure")
                else:
                    elems[ne] = nch
                    nxt.append(ne)
        frontier = nxt
    items = sorted(elems.items(), key=lambda kv: (kv[0] != ident, kv[0]))
    perms = np.array([k for k, _ in items], dtype=np.int32)
    chars = np.array([v for _, v in items], dtype=complex)
    return Group(perms, chars)


@dataclass
class Model:
    number_sites: int
    hamming_weight: int  # -1 = unrestricted
    spin_inversion: int  # 0 = none, +1 / -1
    group: Group
    diag: Terms
    offdiag: Terms
    raw_terms: list = field(default_factory=list)  # [(expr or matrix, sites)]

    @property
    def has_permutations(self):
        return self.group.perms.shape[0] > 1

    @property
    def requires_projection(self):
        # mirrors ls_hs_basis.requires_projection as *used* at
        # /root/reference/src/BatchedOperator.chpl:89,119
        return self.has_permutations or self.spin_inversion != 0

    @property
    def state_index_is_identity(self):
        return self.hamming_weight < 0 and not self.requires_projection

    @property
    def mask(self):
        return (1 << self.number_sites) - 1

    @property
    def max_off_diag(self):
        """number of distinct flip masks == ls_hs_operator_max_number_off_diag
        (/root/reference/src/ForeignTypes.chpl:228-229)."""
        return len(set(int(v) for v in self.offdiag.x))


def model_from_config(cfg: dict) -> Model:
    b = cfg["basis"]
    L = int(b["number_spins"])
    hw = b.get("hamming_weight", None)
    hw = -1 if hw is None else int(hw)
    inv = b.get("spin_inversion", None)
    inv = 0 if inv is None else int(inv)
    syms = b.get("symmetries", []) or []
    group = close_group(L, [s["permutation"] for s in syms], [int(s["sector"]) for s in syms])
    entries = []
    raw = []
    ham = cfg.get("hamiltonian")
    if ham is not None:
        for t in ham["terms"]:
            if "expression" in t:
                k, mat = local_matrix(t["expression"])
                raw.append((t["expression"], t["sites"]))
            else:  # old schema: explicit matrix, /root/reference/data/old/*.yaml
                mat = np.array(t["matrix"], dtype=complex)
                k = int(round(math.log2(mat.shape[0])))
                # old-schema matrices index the FIRST site as the most significant factor
                mat = _reverse_site_order(mat, k)
                raw.append(("matrix", t["sites"]))
            for sites in t["sites"]:
                assert len(sites) == k, (sites, k)
                entries.append((mat, tuple(int(q) for q in sites)))
    allt = terms_from_local_matrices(entries, tol=1e-14)
    is_diag = allt.x == 0
    return Model(L, hw, inv, group, allt.select(is_diag), allt.select(~is_diag), raw)


def _reverse_site_order(mat, k):
    idx = [int(format(a, f"0{k}b")[::-1], 2) if k > 0 else 0 for a in range(1 << k)]
    return mat[np.ix_(idx, idx)]


def load_yaml(path: str) -> Model:
    import yaml

    with open(path, "r", encoding="utf-8") as f:
        return model_from_config(yaml.safe_load(f))


# --------------------------------------------------------------------------------------
# Bit-level helpers (pure Python ints; small cases only)
# --------------------------------------------------------------------------------------


def next_state_fixed_hamming(v: int) -> int:
    """/root/reference/src/StatesEnumeration.chpl:31-34 (Gosper's hack)."""
    t = v | (v - 1)
    ctz = (v & -v).bit_length() - 1
    return ((t + 1) | (((~t & (t + 1)) - 1) >> (ctz + 1))) & 0xFFFFFFFFFFFFFFFF


def hash64_01(x: int) -> int:
    """/root/reference/src/StatesEnumeration.chpl:122-127 (splitmix64 finaliser)."""
    M = 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def locale_idx_of(x: int, num_locales: int) -> int:
    """/root/reference/src/StatesEnumeration.chpl:133-136."""
    return hash64_01(x) % num_locales


def apply_perm(p, s: int) -> int:
    out = 0
    for i, src in enumerate(p):
        out |= ((s >> int(src)) & 1) << i
    return out


def state_info(model: Model, alpha: int):
    """Semantics of ls_hs_state_info as used at /root/reference/src/BatchedOperator.chpl:184-203:
    returns (representative, character, norm) with
      representative = min over the full group (permutations x optional spin flip),
      character      = conj(chi(g0)) for a g0 with g0(alpha) = representative,
      norm           = sqrt( (1/|G|) sum_{g in Stab(alpha)} chi(g) ).
    H~[r', r] = c * character * norm(r') / norm(r)  (derivation in DESIGN.md)."""
    g = model.group
    inv = model.spin_inversion
    order = g.perms.shape[0] * (2 if inv != 0 else 1)
    best = None
    best_ch = None
    stab = 0j
    for p, ch in zip(g.perms, g.chars):
        t = apply_perm(p, alpha)
        cands = [(t, ch)]
        if inv != 0:
            cands.append((t ^ model.mask, ch * inv))
        for tt, cc in cands:
            if tt == alpha:
                stab += cc
            if best is None or tt < best:
                best, best_ch = tt, cc
    n2 = stab.real / order
    if abs(stab.imag) > 1e-9:
        raise AssertionError("stabiliser character sum must be real")
    norm = math.sqrt(n2) if n2 > 1e-12 else 0.0
    return best, np.conj(best_ch), norm


def enumerate_representatives(model: Model):
    """Ascending list of basis states (what enumerateStates produces for numLocales == 1,
    /root/reference/src/StatesEnumeration.chpl:158-224): orbit minima with non-zero norm."""
    L, hw = model.number_sites, model.hamming_weight
    if hw >= 0:
        if hw == 0:
            cands = [0]
        else:
            cands = []
            v = (1 << hw) - 1
            top = ((1 << hw) - 1) << (L - hw)
            while True:
                cands.append(v)
                if v == top:
                    break
                v = next_state_fixed_hamming(v)
    else:
        cands = range(1 << L)
    if not model.requires_projection:
        return np.array(list(cands), dtype=np.uint64)
    out = []
    for s in cands:
        rep, _, norm = state_info(model, s)
        if rep == s and norm > 0:
            out.append(s)
    return np.array(out, dtype=np.uint64)


# --------------------------------------------------------------------------------------
# Term-table matvec in pure Python (tiny cases) -- mirrors localMatrixVector semantics:
# y[i] = d(alpha_i) x[i]  (assign), then y[idx(beta)] += c x[i]
# (/root/reference/src/DistributedMatrixVector.chpl:36-127,1055-1070)
# --------------------------------------------------------------------------------------


def _apply_terms(terms: Terms, alpha: int):
    out = {}
    for v, m, r, x, s in zip(terms.v, terms.m, terms.r, terms.x, terms.s):
        if (alpha & int(m)) == int(r):
            sign = -1.0 if bin(alpha & int(s)).count("1") & 1 else 1.0
            beta = alpha ^ int(x)
            out[beta] = out.get(beta, 0) + sign * v
    return out


def matvec_terms_python(model: Model, reps, x):
    reps = [int(r) for r in reps]
    index = {r: i for i, r in enumerate(reps)}
    y = np.zeros(len(reps), dtype=np.result_type(x.dtype, np.float64) if not np.iscomplexobj(x) else complex)
    ycomplex = np.zeros(len(reps), dtype=complex)
    norms = None
    if model.requires_projection and model.has_permutations:
        norms = [state_info(model, r)[2] for r in reps]
    for i, a in enumerate(reps):
        d = _apply_terms(model.diag, a)
        ycomplex[i] += d.get(a, 0) * x[i]
    for i, a in enumerate(reps):
        for beta, c in _apply_terms(model.offdiag, a).items():
            if c == 0:
                continue
            c = c * x[i]
            if not model.requires_projection:
                j = index[beta]
            elif not model.has_permutations:
                inv = beta ^ model.mask
                if inv < beta:
                    beta = inv
                    c = c * model.spin_inversion
                j = index[beta]
            else:
                rep, ch, nb = state_info(model, beta)
                c = c * ch * nb / norms[i]
                if c == 0:
                    continue
                j = index[rep]
            ycomplex[j] += c
    if np.iscomplexobj(x):
        return ycomplex
    assert np.abs(ycomplex.imag).max(initial=0.0) < 1e-12
    return ycomplex.real.copy()


# --------------------------------------------------------------------------------------
# Dense / Kronecker oracle: completely independent of the term tables
# --------------------------------------------------------------------------------------


def dense_hamiltonian_full(cfg: dict):
    """Sparse 2^L x 2^L matrix straight from the YAML expressions via explicit single-site
    Pauli matrices (L <= 16)."""
    import scipy.sparse as sp

    L = int(cfg["basis"]["number_spins"])
    assert L <= 16
    dim = 1 << L
    states = np.arange(dim, dtype=np.int64)
    rows, cols, vals = [], [], []
    for t in cfg["hamiltonian"]["terms"]:
        if "expression" in t:
            k, mat = local_matrix(t["expression"])
        else:
            mat = np.array(t["matrix"], dtype=complex)
            k = int(round(math.log2(mat.shape[0])))
            mat = _reverse_site_order(mat, k)
        for sites in t["sites"]:
            # local input pattern of every state
            b = np.zeros(dim, dtype=np.int64)
            for q, site in enumerate(sites):
                b |= ((states >> site) & 1) << q
            rest = states.copy()
            for site in sites:
                rest &= ~(1 << site)
            for a in range(1 << k):
                coeff = mat[a, b]  # <a|M|b> for every column state
                nz = coeff != 0
                if not nz.any():
                    continue
                out = rest.copy()
                for q, site in enumerate(sites):
                    if (a >> q) & 1:
                        out |= 1 << site
                rows.append(out[nz])
                cols.append(states[nz])
                vals.append(coeff[nz])
    H = sp.coo_matrix(
        (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(dim, dim)
    ).tocsr()
    H.sum_duplicates()
    return H


def dense_sector_matrix(cfg: dict):
    """H restricted to the symmetry sector, built with an explicit projector
    P = (1/|G|) sum_g conj(chi(g)) U_g on the fixed-Hamming subspace; columns of the isometry
    are P|r>/||P|r>|| for every representative r with non-zero norm.  Returns (reps, Hsector)."""
    model = model_from_config(cfg)
    H = dense_hamiltonian_full(cfg)
    L, hw = model.number_sites, model.hamming_weight
    if hw >= 0:
        sector_states = np.array(
            [s for s in range(1 << L) if bin(s).count("1") == hw], dtype=np.int64
        )
    else:
        sector_states = np.arange(1 << L, dtype=np.int64)
    pos = {int(s): i for i, s in enumerate(sector_states)}
    Hs = H[sector_states][:, sector_states]
    n = len(sector_states)
    if not model.requires_projection:
        return sector_states.astype(np.uint64), np.asarray(Hs.todense())
    g = model.group
    elems = []
    for p, ch in zip(g.perms, g.chars):
        elems.append((p, False, ch))
        if model.spin_inversion != 0:
            elems.append((p, True, ch * model.spin_inversion))
    order = len(elems)
    reps = []
    cols = []
    seen = set()
    for s in sector_states:
        s = int(s)
        if s in seen:
            continue
        vec = np.zeros(n, dtype=complex)
        orbit = set()
        for p, flip, ch in elems:
            t = apply_perm(p, s)
            if flip:
                t ^= model.mask
            orbit.add(t)
            vec[pos[t]] += np.conj(ch) / order
        seen |= orbit
        nrm = np.linalg.norm(vec)
        if nrm > 1e-9:
            r = min(orbit)
            # phase convention: column = P|r>/||P|r>|| with r the orbit minimum
            vec_r = np.zeros(n, dtype=complex)
            for p, flip, ch in elems:
                t = apply_perm(p, r)
                if flip:
                    t ^= model.mask
                vec_r[pos[t]] += np.conj(ch) / order
            reps.append(r)
            cols.append(vec_r / np.linalg.norm(vec_r))
    srt = np.argsort(reps)
    reps = np.array(reps, dtype=np.uint64)[srt]
    B = np.array(cols)[srt].T  # n x nreps
    Hd = np.asarray(Hs.todense())
    return reps, B.conj().T @ Hd @ B


# --------------------------------------------------------------------------------------
# Model library: the in-tree reference configurations, regenerated (no reference files needed)
# --------------------------------------------------------------------------------------


def heisenberg_chain_config(L: int, symm: bool = False, spin_inversion=None):
    """Same content as /root/reference/data/heisenberg_chain_{L}[_symm].yaml (periodic ring,
    sigma.sigma on every bond, half filling); equality with the reference files is asserted in
    tests/test_oracle_model.py when /root/reference is present."""
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
    terms = [
        {"expression": "σˣ₀ σˣ₁", "sites": lattice},
        {"expression": "σʸ₀ σʸ₁", "sites": lattice},
        {"expression": "σᶻ₀ σᶻ₁", "sites": lattice},
    ]
    return {"basis": basis, "hamiltonian": {"name": "Heisenberg Hamiltonian", "terms": terms}}


def binomial(n, k):
    return math.comb(n, k) if 0 <= k <= n else 0

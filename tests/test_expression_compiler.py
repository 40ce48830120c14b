"""Differential test of the C expression compiler (csrc/yaml.c: `expression:` + `sites:` -> non-branching terms, the tables every
kernel reads) against a construction that shares nothing with it: the operator as an explicit sum of Kronecker products of 2 x 2
matrices on the full 2^L space.  The reference's inputs only ever write "c x A_0 B_1" with Pauli matrices (tests/test_dense_pin.py
covers those through the oracle); the grammar the loader accepts is wider -- sigma / S, x y z + -, several factors on one site,
products of scalars, imaginary scalars, one to four sites per tuple -- and a maintainer's own YAML may use all of it
(ls_hs_load_yaml_config, /root/reference/src/FFI.chpl:208, ForeignTypes.chpl:261-288).

Convention (lattice-symmetries): site i is bit i of the state, bit value 0 is spin up; a term (v, m, r, x, s) of the ABI acts as
|alpha> -> v (-1)^popcount(alpha & s) |alpha ^ x> when alpha & m == r (/root/reference/src/FFI.chpl:107-119)."""
import numpy as np
import pytest

import distributed_matvec_amd as D
from distributed_matvec_amd import _lib
from helpers import apply_terms_python, product_terms

SUB = "₀₁₂₃₄₅₆₇₈₉"
KINDS = {"x": ("ˣ", np.array([[0, 1], [1, 0]], dtype=complex)),
         "y": ("ʸ", np.array([[0, -1j], [1j, 0]], dtype=complex)),
         "z": ("ᶻ", np.array([[1, 0], [0, -1]], dtype=complex)),
         "+": ("⁺", np.array([[0, 1], [0, 0]], dtype=complex)),   # |up><down|, up = bit value 0
         "-": ("⁻", np.array([[0, 0], [1, 0]], dtype=complex))}


def random_term(rs, L):
    """(expression text, site tuples, dense matrix on 2^L)"""
    arity = int(rs.randint(1, 5))
    n_factors = int(rs.randint(arity, arity + 4))
    local = list(range(arity)) + [int(rs.randint(arity)) for _ in range(n_factors - arity)]  # every local index at least once
    rs.shuffle(local)
    pieces, factors, scalar = [], [], 1.0 + 0j
    for _ in range(int(rs.randint(0, 3))):
        v = float(rs.choice([2.0, -1.0, 0.5, 0.25, -3.0]))
        if rs.rand() < 0.3:
            pieces.append(f"{v}j")
            scalar *= 1j * v
        else:
            pieces.append(repr(v))
            scalar *= v
        if rs.rand() < 0.7:
            pieces.append("×")
    for idx in local:
        kind = str(rs.choice(list(KINDS)))
        spin = rs.rand() < 0.3
        pieces.append(("S" if spin else "σ") + KINDS[kind][0] + SUB[idx])
        factors.append((idx, KINDS[kind][1] * (0.5 if spin else 1.0)))
    expr = " ".join(pieces)
    tuples = [[int(s) for s in rs.choice(L, size=arity, replace=False)] for _ in range(int(rs.randint(1, 4)))]
    dim = 1 << L
    H = np.zeros((dim, dim), dtype=complex)
    for sites in tuples:
        prod = np.eye(dim, dtype=complex)
        for idx, mat in factors:  # written left to right = matrix product in that order
            site = sites[idx]
            full = np.array([[1.0]], dtype=complex)
            for q in range(L - 1, -1, -1):  # site 0 = least significant bit = last Kronecker factor
                full = np.kron(full, mat if q == site else np.eye(2))
            prod = prod @ full
        H += scalar * prod
    return expr, tuples, H


@pytest.mark.parametrize("seed", range(12))
def test_c_expression_compiler_equals_kronecker_products(seed):
    rs = np.random.RandomState(1000 + seed)
    L = int(rs.randint(4, 7))
    lib = _lib.load()
    for _ in range(25):
        n_terms = int(rs.randint(1, 4))
        terms = [random_term(rs, L) for _ in range(n_terms)]
        text = f"basis:\n  number_spins: {L}\nhamiltonian:\n  terms:\n"
        for expr, tuples, _ in terms:
            text += f'    - expression: "{expr}"\n      sites: {tuples}\n'
        H = sum(t[2] for t in terms)
        conf = lib.ls_amd_load_yaml_config_from_string(text.encode("utf-8"))
        if not conf:  # the only admissible refusal: the sum has no term left at all
            assert np.abs(H).max() < 1e-14, (text, lib.ls_amd_last_error())
            continue
        try:
            op = D.Operator(conf.contents.hamiltonian, owning=False)
            diag, off = product_terms(op)
            # the host mirror the tests and bench.py build their operators with (config.py -> ls_hs_create_operator_from_terms)
            # compiles the same tables from the same text
            cfg = {"basis": {"number_spins": L}, "hamiltonian": {"terms": [{"expression": e, "sites": t} for e, t, _ in terms]}}
            _basis, mirror = D.loadConfigFromDict(cfg, hamiltonian=True)
            assert product_terms(mirror) == (diag, off), text
            got = np.zeros_like(H)
            for alpha in range(1 << L):
                for beta, c in apply_terms_python(diag + off, alpha).items():
                    got[beta, alpha] += c
            np.testing.assert_allclose(got, H, rtol=0, atol=1e-12, err_msg=text)
            assert op.isHermitian == bool(np.abs(H - H.conj().T).max() < 1e-12), text
            assert op.isReal == bool(np.abs(H.imag).max() < 1e-12) or np.abs(H).max() < 1e-14, text
            # the diagonal / off-diagonal split of the ABI: no flip mask among the diagonal terms, none empty among the others
            assert all(x == 0 for _v, _m, _r, x, _s in diag) and all(x != 0 for _v, _m, _r, x, _s in off), text
        finally:
            lib.ls_hs_destroy_yaml_config(conf)


def _conserving_term(rs, L):
    """like random_term, but the product conserves the number of up spins: as many raising as lowering factors, any number of z"""
    arity = int(rs.randint(2, 5))
    pairs = int(rs.randint(1, 3))
    kinds = ["+"] * pairs + ["-"] * pairs + ["z"] * int(rs.randint(0, 3))
    rs.shuffle(kinds)
    local = [int(rs.randint(arity)) for _ in kinds]
    for k in range(arity):  # every local index at least once
        if k not in local:
            kinds.append("z")
            local.append(k)
    scalar = complex(float(rs.choice([1.0, -0.5, 2.0])), 0.0) * (1j if rs.rand() < 0.3 else 1.0)
    text = (f"{scalar.imag}j" if scalar.real == 0 else repr(scalar.real)) + " × " + " ".join(
        ("S" if s else "σ") + KINDS[k][0] + SUB[i] for k, i, s in zip(kinds, local, [rs.rand() < 0.3 for _ in kinds]))
    # rebuild the factors from the text just written, so that the S / sigma choice is the one in the text
    factors = []
    for piece in text.split(" × ")[1].split(" "):
        mat = KINDS[{v[0]: k for k, v in KINDS.items()}[piece[1]]][1] * (0.5 if piece[0] == "S" else 1.0)
        factors.append((SUB.index(piece[2]), mat))
    tuples = [[int(s) for s in rs.choice(L, size=arity, replace=False)] for _ in range(int(rs.randint(1, 4)))]
    dim = 1 << L
    H = np.zeros((dim, dim), dtype=complex)
    for sites in tuples:
        prod = np.eye(dim, dtype=complex)
        for idx, mat in factors:
            full = np.array([[1.0]], dtype=complex)
            for q in range(L - 1, -1, -1):
                full = np.kron(full, mat if q == sites[idx] else np.eye(2))
            prod = prod @ full
        H += scalar * prod
    return text, tuples, H


@pytest.mark.parametrize("seed", range(8))
def test_oracle_matvec_equals_kronecker_products_on_random_operators(seed):
    """the CHECKER on operators the reference's inputs never contain (non-Hermitian, complex, non-exchange, up to four sites per
    term): oracle/ls_oracle.c + oracle/model.py (a third expression compiler) against the same explicit construction, on the full
    space and -- for operators that conserve the number of up spins -- inside a fixed-weight sector (combinadic ranks, DMV:73-127)"""
    from oracle import c_oracle as CO
    from oracle import model as M

    rs = np.random.RandomState(2000 + seed)
    L = int(rs.randint(4, 8))
    for _ in range(12):
        sector = rs.rand() < 0.5
        terms = [(_conserving_term if sector else random_term)(rs, L) for _ in range(int(rs.randint(1, 4)))]
        H = sum(t[2] for t in terms)
        hw = int(rs.randint(1, L)) if sector else None
        cfg = {"basis": {"number_spins": L, "hamming_weight": hw},
               "hamiltonian": {"name": "random", "terms": [{"expression": e, "sites": t} for e, t, _ in terms]}}
        o = CO.COracle(M.model_from_config(cfg))
        reps = o.enumerate()
        states = np.arange(1 << L, dtype=np.uint64)
        want_reps = states if hw is None else states[np.bitwise_count(states) == hw]
        assert np.array_equal(reps, want_reps)
        idx = reps.astype(np.int64)
        Hs = H[np.ix_(idx, idx)]
        if hw is not None:  # the operator conserves the weight: nothing leaves the sector
            mask = np.ones(1 << L, dtype=bool)
            mask[idx] = False
            assert np.abs(H[np.ix_(mask, ~mask)]).max(initial=0.0) < 1e-14
        x = (rs.rand(len(reps)) - 0.5) + 1j * (rs.rand(len(reps)) - 0.5)
        np.testing.assert_allclose(o.local_matvec(reps, x), Hs @ x, rtol=0, atol=1e-12, err_msg=str(cfg))
        if np.abs(Hs.imag).max() < 1e-14:  # a real operator also takes real vectors (the f64 entry points)
            xr = rs.rand(len(reps)) - 0.5
            np.testing.assert_allclose(o.local_matvec(reps, xr), Hs.real @ xr, rtol=0, atol=1e-12, err_msg=str(cfg))

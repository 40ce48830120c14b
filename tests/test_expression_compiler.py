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

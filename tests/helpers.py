"""Shared helpers of the test-suite: model library (tests/golden/models.json), oracle handles."""
import ctypes as C
import functools
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SMALL_MODELS = [
    "heisenberg_chain_4", "heisenberg_chain_6", "heisenberg_chain_8", "heisenberg_chain_10",
    "heisenberg_chain_12", "heisenberg_chain_16", "heisenberg_kagome_12", "heisenberg_kagome_12_symm",
    "heisenberg_kagome_16", "heisenberg_square_4x4", "issue_01",
]
# the reference's `make check` matvec matrix (/root/reference/Makefile:88-125) that the oracle finishes
# in seconds
CHECK_MODELS = SMALL_MODELS + ["heisenberg_chain_20", "heisenberg_chain_24_symm"]


@functools.lru_cache(maxsize=None)
def golden():
    with open(os.path.join(HERE, "golden", "models.json"), encoding="utf-8") as f:
        return json.load(f)


@functools.lru_cache(maxsize=None)
def golden_vectors():
    return np.load(os.path.join(HERE, "golden", "vectors.npz"))


def model_config(name):
    return golden()["models"][name]["config"]


@functools.lru_cache(maxsize=None)
def oracle_for(name):
    from oracle import c_oracle as CO
    from oracle import model as M

    m = M.model_from_config(model_config(name))
    return CO.COracle(m)


@functools.lru_cache(maxsize=None)
def oracle_reps(name):
    return oracle_for(name).enumerate()


def complex_translation_config(L, sector):
    """a chain with a complex character (not exercised by any in-tree reference config)."""
    from oracle import model as M

    c = M.heisenberg_chain_config(L)
    c["basis"]["symmetries"] = [{"permutation": [(i + 1) % L for i in range(L)], "sector": sector}]
    return c


def approx_equal(a, b, atol=1e-14, rtol=1e-12):
    """/root/reference/test/TestMatrixVectorProduct.chpl:15-20"""
    a = np.asarray(a)
    b = np.asarray(b)
    return np.abs(a - b) <= np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b)))


def product_terms(op):
    """(diag, off) term tables of a product Operator as python lists (v, m, r, x, s), read through
    the ls_hs_nonbranching_terms ABI."""
    def read(nbt_ptr):
        if not nbt_ptr:
            return []
        t = nbt_ptr.contents
        n = t.number_terms
        v = np.frombuffer((C.c_double * (2 * n)).from_address(t.v), dtype=np.float64).reshape(n, 2)
        arr = lambda p: np.frombuffer((C.c_uint64 * n).from_address(p), dtype=np.uint64)  # noqa: E731
        m, r, x, s = arr(t.m), arr(t.r), arr(t.x), arr(t.s)
        return [(complex(v[i, 0], v[i, 1]), int(m[i]), int(r[i]), int(x[i]), int(s[i])) for i in range(n)]

    c = op.payload.contents
    return read(c.diag_terms), read(c.off_diag_terms)


def apply_terms_python(terms, alpha):
    out = {}
    for v, m, r, x, s in terms:
        if (alpha & m) == r:
            sign = -1.0 if bin(alpha & s).count("1") & 1 else 1.0
            out[alpha ^ x] = out.get(alpha ^ x, 0) + sign * v
    return {k: v for k, v in out.items() if v != 0}

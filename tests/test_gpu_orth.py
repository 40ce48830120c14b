"""ls_amd_orth_pass (csrc/orth.hip): the fused Gram-Schmidt sweep of the eigensolver callers against plain PyTorch f64 -- the
update w <- w - V^T h, the overlaps <V[k], w> over the UPDATED vector and its squared norm, in one pass; odd lengths, a row stride
larger than the length, one row and the maximum number of rows, and the solver-level property (orthogonal to rounding after two
sweeps)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    if not t.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the product has no CPU fallback)")
    t.cuda.set_device(0)
    return t


def _pass(torch, lib, V, w, h):
    m, n = V.shape
    out = torch.full((m + 1,), 7.0, dtype=torch.float64, device="cuda")
    rc = lib.ls_amd_orth_pass(m, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(w.data_ptr()),
                              C.c_void_p(h.data_ptr()) if h is not None else None, C.c_void_p(out.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("m,n,pad", [(1, 1000, 0), (5, 100001, 0), (12, 1 << 20, 6), (32, 300007, 2), (3, 7, 0), (8, 513, 1)])
def test_orth_pass_matches_torch(torch, m, n, pad):
    from distributed_matvec_amd import _lib

    lib = _lib.load()
    assert lib.ls_amd_orth_max_rows() == 32
    g = torch.Generator(device="cuda").manual_seed(100 * m + pad)
    store = torch.randn((m, n + pad), dtype=torch.float64, device="cuda", generator=g)
    V = store[:, :n]  # row stride n + pad
    w = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    # pass 1: overlaps and norm, w untouched
    w1 = w.clone()
    out = _pass(torch, lib, V, w1, None)
    assert torch.equal(w1, w)
    want_h = torch.mv(V, w)
    scale = float(want_h.abs().max()) + float(torch.dot(w, w))
    assert float((out[:m] - want_h).abs().max()) <= 1e-12 * scale
    assert abs(float(out[m]) - float(torch.dot(w, w))) <= 1e-12 * scale
    # pass 2: apply coefficients, overlaps of the updated vector, its norm
    h = torch.randn(m, dtype=torch.float64, device="cuda", generator=g)
    w2 = w.clone()
    out = _pass(torch, lib, V, w2, h)
    want_w = w - torch.mv(V.t(), h)
    assert float((w2 - want_w).abs().max()) <= 1e-12 * float(want_w.abs().max())
    want_h2 = torch.mv(V, want_w)
    scale = float(want_h2.abs().max()) + float(torch.dot(want_w, want_w))
    assert float((out[:m] - want_h2).abs().max()) <= 1e-12 * scale
    assert abs(float(out[m]) - float(torch.dot(want_w, want_w))) <= 1e-12 * scale
    assert lib.ls_amd_orth_pass(33, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(w.data_ptr()), None, C.c_void_p(out.data_ptr()), None) != 0


def test_two_sweeps_orthogonalise_to_rounding(torch):
    """what lanczos_smallest relies on: against an orthonormal block, sweep 1 + sweep 2 leave overlaps at rounding level and the
    second sweep reports them and the norm of the result"""
    from distributed_matvec_amd import _lib

    lib = _lib.load()
    m, n = 16, 1 << 18
    g = torch.Generator(device="cuda").manual_seed(5)
    Q, _ = torch.linalg.qr(torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g))
    V = Q.t().contiguous()
    w = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) + 50.0 * V[3]
    out = _pass(torch, lib, V, w, None)
    h = out[:m].clone()
    out = _pass(torch, lib, V, w, h)
    left, nrm = out[:m], float(out[m]) ** 0.5
    assert float(left.abs().max()) <= 1e-12 * nrm * 60
    assert abs(nrm - float(torch.linalg.vector_norm(w))) <= 1e-12 * nrm
    assert float(torch.mv(V, w).abs().max()) <= 1e-12 * nrm * 60


@pytest.mark.parametrize("m_in,m_out,n,pad", [(12, 8, 100003, 0), (32, 32, 4099, 3), (5, 1, 1 << 18, 2), (2, 2, 3, 0)])
def test_basis_rotate_matches_torch(torch, m_in, m_out, n, pad):
    """ls_amd_basis_rotate (thick restart): V[:m_out] <- S^T V[:m_in] in place == torch.mm on a copy; the rows behind m_out stay"""
    from distributed_matvec_amd import _lib

    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(7 * m_in + m_out)
    store = torch.randn((m_in + 1, n + pad), dtype=torch.float64, device="cuda", generator=g)
    V = store[:, :n]
    S = torch.randn((m_in, m_out), dtype=torch.float64, device="cuda", generator=g)
    want = torch.mm(S.t(), V[:m_in])
    tail = V[m_out:].clone()
    assert lib.ls_amd_basis_rotate(m_in, m_out, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(S.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert float((V[:m_out] - want).abs().max()) <= 1e-12 * float(want.abs().max())
    assert torch.equal(V[m_out:], tail)
    assert lib.ls_amd_basis_rotate(m_out, m_in + 1, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(S.data_ptr()), None) != 0


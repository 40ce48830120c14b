"""The HIP path against the Kronecker-product construction of tests/test_dense_pin.py directly (not through the oracle): enumeration
bit-exact, y element by element, f64 and c128, pull and push, one and three locales -- on the reference's small models, incl. a
-1 character, a spin-inversion sector and a non-cyclic lattice group."""
import numpy as np
import pytest

from helpers import model_config
from test_dense_pin import dense_projected

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    if not t.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the product has no CPU fallback)")
    t.cuda.set_device(0)
    return t


@pytest.mark.parametrize("name", ["heisenberg_chain_10", "heisenberg_chain_12", "heisenberg_kagome_12", "heisenberg_kagome_12_symm",
                                  "issue_01", "heisenberg_kagome_16", "heisenberg_square_4x4"])
def test_hip_matvec_equals_kronecker_construction(torch, name):
    import scipy.sparse as sp

    import distributed_matvec_amd as D

    cfg = model_config(name)
    reps, H = dense_projected(cfg)
    H = (H.toarray() if sp.issparse(H) else np.asarray(H)).real
    rs = np.random.RandomState(7)
    x = rs.rand(len(reps)) - 0.5
    xc = x + 1j * (rs.rand(len(reps)) - 0.5)
    for P in (1, 3):
        basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
        parts, masks = D.enumerateStates(basis, P)
        got_reps = D.arrFromHashedToBlock(parts, masks).cpu().numpy().view(np.uint64)
        assert np.array_equal(got_reps, reps), name
        for vec in (x, xc):
            for mode in (("pull", "push") if P == 1 else ("auto",)):
                xb = torch.from_numpy(np.ascontiguousarray(vec)).cuda()
                xs = D.arrFromBlockToHashed(xb, masks, P)
                ys = [torch.zeros_like(v) for v in xs]
                D.matrixVectorProduct(h, xs, ys, parts, mode=mode)
                got = D.arrFromHashedToBlock(ys, masks).cpu().numpy()
                want = H @ vec
                assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (name, P, mode, vec.dtype)

"""The caller of the hot path in BASELINE config 5: eigensolve through the device-resident driver
(stands in for /root/reference/src/Diagonalize.chpl + PRIMME), checked against known energies."""
import numpy as np
import pytest

from helpers import model_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    if not t.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    t.cuda.set_device(0)
    return t


def test_chain_10_ground_state(torch, tmp_path):
    """E0 = -18.061785417968 in the (h=5, inversion -1) sector."""
    from distributed_matvec_amd.diagonalize import diagonalize

    out = str(tmp_path / "ed.npz")
    r = diagonalize(model_config("heisenberg_chain_10"), num_evals=2, eps=1e-10, output=out)
    assert r.converged
    assert abs(r.eigenvalues[0] - (-18.061785417968)) < 1e-8
    d = np.load(out)
    v = d["hamiltonian/eigenvectors"][0]
    assert abs(np.linalg.norm(v) - 1) < 1e-10
    assert len(d["basis/representatives"]) == 126


def test_symmetric_sector_contains_the_ground_state(torch):
    """E0 of chain_24 in the fully symmetric sector (tile kernel + projection) equals E0 of the
    unsymmetrised chain_24 (fused pull kernel): two different kernel families, one number."""
    from distributed_matvec_amd.diagonalize import diagonalize

    r_full = diagonalize(model_config("heisenberg_chain_24"), num_evals=1, eps=1e-9)
    r_symm = diagonalize(model_config("heisenberg_chain_24_symm"), num_evals=1, eps=1e-9)
    r_symm_p3 = diagonalize(model_config("heisenberg_chain_24_symm"), num_evals=1, eps=1e-9, num_partitions=3)
    assert r_full.converged and r_symm.converged and r_symm_p3.converged
    assert abs(r_full.eigenvalues[0] - r_symm.eigenvalues[0]) < 1e-6
    assert abs(r_symm.eigenvalues[0] - r_symm_p3.eigenvalues[0]) < 1e-6
    # independent number: ARPACK on the oracle's (CPU) symmetric-sector matvec
    import scipy.sparse.linalg as spla

    from helpers import oracle_for, oracle_reps

    o, reps = oracle_for("heisenberg_chain_24_symm"), oracle_reps("heisenberg_chain_24_symm")
    A = spla.LinearOperator((len(reps), len(reps)), matvec=lambda v: o.local_matvec(reps, np.ascontiguousarray(v, dtype=np.float64)), dtype=np.float64)
    e0 = spla.eigsh(A, k=1, which="SA", tol=1e-10)[0][0]
    assert abs(r_symm.eigenvalues[0] - e0) < 1e-6
    assert -0.4447 < r_full.eigenvalues[0] / (4 * 24) < -0.4444  # finite-size Heisenberg ring, S.S units per site


def test_chain_12_symmetric_known_energy(torch):
    """Appendix B: chain_12, h=6, translation+reflection sector 0, inversion +1: E0 = -21.549563669781."""
    from distributed_matvec_amd import config
    from distributed_matvec_amd.diagonalize import diagonalize

    r = diagonalize(config.heisenberg_chain_config(12, symm=True), num_evals=1, eps=1e-10, dtype=torch.complex128)
    assert r.converged and abs(r.eigenvalues[0] - (-21.549563669781)) < 1e-8

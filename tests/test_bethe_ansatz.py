"""Third-party pins of the parity chain ("Pin parity with something the builder did not write").

1. The Bethe-ansatz ground-state energy of the periodic Heisenberg ring (oracle/bethe.py: L/2 coupled transcendental
   equations, no term table, no projector) against (a) SURVEY.md Appendix B's known answers, (b) dense diagonalisation of
   explicit Kronecker-product matrices, (c) the C oracle's matvec (the thing every GPU parity test compares with) driven
   by scipy's Lanczos -- in the full S^z = 0 sector and in the reference's symmetry-projected sector (chain_24_symm).
   The GPU tests then hold the HIP path's Lanczos E0 of chain_{24,32,36,40}_symm against the same numbers.
2. tests/golden/vectors.npz against the numbers SURVEY.md Appendix B recorded in the survey session (scratch-verified
   there with an independent dense construction, not by the builder of this repository).
"""
import numpy as np
import pytest

from helpers import golden_vectors, model_config, oracle_for, oracle_reps
from oracle import bethe


def test_bethe_energy_known_answers():
    # SURVEY.md Appendix B (sigma.sigma units): chain_10 E0 = -18.061785417968, chain_12 E0 = -21.549563669781;
    # L = 4: singlet of the 4-ring, E0 = -2 J = -8 in sigma units; textbook S.S value for L = 10: -4.515446354
    assert abs(bethe.ground_state_energy_sigma(10) - (-18.061785417968)) < 5e-12
    assert abs(bethe.ground_state_energy_sigma(12) - (-21.549563669781)) < 5e-12
    assert abs(bethe.ground_state_energy_sigma(4) - (-8.0)) < 1e-13
    assert abs(bethe.ground_state_energy_SS(10) - (-4.515446354)) < 1e-9
    # Hulthen: E0 / L -> 1/4 - ln 2 from below with the -pi^2 / (12 L^2) finite-size term
    for L in (24, 32, 36, 40, 64):
        per_site = bethe.ground_state_energy_SS(L) / L
        assert per_site < bethe.E_INFINITY_PER_SITE_SS
        corr = per_site - bethe.E_INFINITY_PER_SITE_SS
        assert abs(corr / (-np.pi ** 2 / (12 * L * L)) - 1.0) < 0.12, (L, corr)


@pytest.mark.parametrize("L", [4, 6, 8, 10, 12])
def test_bethe_energy_equals_dense_diagonalisation(L):
    """explicit 2^L Kronecker-product Hamiltonian (Pauli matrices; nothing of the product or the C oracle)"""
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sy = np.array([[0, -1j], [1j, 0]])
    sz = np.diag([1.0, -1.0]).astype(complex)

    def site_op(o, i):
        m = np.array([[1.0 + 0j]])
        for k in range(L):
            m = np.kron(m, o if k == i else np.eye(2))
        return m

    H = np.zeros((2 ** L, 2 ** L), dtype=complex)
    for i in range(L):
        j = (i + 1) % L
        for o in (sx, sy, sz):
            H += site_op(o, i) @ site_op(o, j)
    e0 = np.linalg.eigvalsh(H)[0]
    assert abs(e0 - bethe.ground_state_energy_sigma(L)) < 1e-10 * abs(e0)


def _oracle_e0(name):
    from scipy.sparse.linalg import LinearOperator, eigsh

    o = oracle_for(name)
    reps = oracle_reps(name)
    n = len(reps)
    op = LinearOperator((n, n), matvec=lambda v: o.local_matvec(reps, np.ascontiguousarray(v, dtype=np.float64)), dtype=np.float64)
    return float(eigsh(op, k=1, which="SA", tol=1e-12)[0][0])


@pytest.mark.parametrize("name,L", [("heisenberg_chain_12", 12), ("heisenberg_chain_16", 16), ("heisenberg_chain_24_symm", 24)])
def test_c_oracle_ground_state_equals_bethe(name, L):
    """the checker of the GPU parity tests (oracle/ls_oracle.c: term tables, and for _symm the projection
    c conj(chi) n(r')/n(r) over the 4L-element group) reproduces a number it had no part in: 1e-10 relative"""
    e0 = _oracle_e0(name)
    want = bethe.ground_state_energy_sigma(L)
    assert abs(e0 - want) <= 1e-10 * abs(want), (e0, want)


def test_golden_vectors_match_survey_appendix_b():
    """SURVEY.md Appendix B, heisenberg_chain_10 with the golden-recipe x (the survey session's own dense construction):
    x[:3], y[:3], ||y||_2, sum(y)"""
    v = golden_vectors()
    x, y, reps = v["heisenberg_chain_10/x"], v["heisenberg_chain_10/y"], v["heisenberg_chain_10/representatives"]
    assert len(x) == 126 and reps[:5].tolist() == [31, 47, 55, 59, 61] and int(reps[-1]) == 496
    assert np.allclose(x[:3], [0.02273282938199406, -0.07245898164145037, -0.4745808732559048], rtol=0, atol=1e-16)
    assert np.allclose(y[:3], [-0.5015037972269847, -0.8684407077991885, -1.9586230522118682], rtol=1e-13, atol=0)
    assert abs(np.linalg.norm(y) - 17.014435846369427) < 1e-12
    assert abs(y.sum() - (-15.391509897327113)) < 1e-12
    # fresh-seed variant of the same table: x = RandomState(42).rand(126) - 0.5 through the C oracle
    x2 = np.random.RandomState(42).rand(126) - 0.5
    y2 = oracle_for("heisenberg_chain_10").local_matvec(oracle_reps("heisenberg_chain_10"), x2)
    assert np.allclose(y2[:3], [-0.6435132739429792, 0.8575238407016947, 2.0282782433657394], rtol=1e-13, atol=0)
    assert abs(np.linalg.norm(y2) - 19.418360955956267) < 1e-12 and abs(x2 @ y2 - (-1.8478819708706284)) < 1e-12


def test_hash64_01_and_partition_sizes_match_survey_appendix_b():
    from oracle import c_oracle as CO

    states = np.array([0x1, 0x1F, 0x1F0, 0x155, 0xFFFF, 0xFFFF0000FFFF], dtype=np.uint64)
    assert CO.locale_idx_of(states, 8).tolist() == [5, 5, 4, 2, 5, 7]
    assert CO.locale_idx_of(states, 3).tolist() == [1, 0, 2, 2, 2, 1]


def test_c_oracle_square_4x4_equals_the_published_energy():
    """A two-dimensional pin: the 4 x 4 periodic square-lattice Heisenberg antiferromagnet has E0 / N = -0.7017802 J in S.S units
    (Schulz, Ziman, Poilblanc, PRB 54, 12946 (1996), table of finite-cluster energies; the textbook exact-diagonalisation
    benchmark), i.e. E0 = -11.228483 J = -44.913933 in the sigma.sigma units of data/heisenberg_square_4x4.yaml.  Lanczos on
    the C oracle's matvec in the symmetry-adapted sector the file names (107 representatives of a non-cyclic lattice group;
    vertical and wrap-around bonds: exchange groups no chain exercises)."""
    e0 = _oracle_e0("heisenberg_square_4x4")
    assert abs(e0 / (4 * 16) - (-0.7017802)) < 5e-8, e0

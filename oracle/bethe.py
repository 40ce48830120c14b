"""Bethe-ansatz ground-state energy of the periodic spin-1/2 Heisenberg ring -- TEST INFRASTRUCTURE (oracle/).

A pin of the parity tests that owes nothing to this repository's term tables, symmetry projectors or oracles: Bethe's
1931 solution of H = J sum_i S_i . S_{i+1} (Hulthen 1938 for the antiferromagnetic ground state).  For even L and
M = L/2 overturned spins the ground state is the solution of the M coupled Bethe equations in the rapidities lambda_j

    L * 2 arctan(2 lambda_j) = 2 pi I_j + sum_{k != j} 2 arctan(lambda_j - lambda_k),
    I_j = -(M - 1)/2, -(M - 1)/2 + 1, ..., (M - 1)/2          (the symmetric, densest set of quantum numbers)

and  E / J = L/4 - sum_j 2 / (4 lambda_j^2 + 1).  The reference's Hamiltonian is sum_bonds sigma.sigma = 4 sum S.S
(data/heisenberg_chain_*.yaml), so E_sigma = 4 E.  For L a multiple of 4 this state has momentum 0,
is even under reflection and under the global spin flip, i.e. it lives in the trivial sector that the reference's
heisenberg_chain_{24,32,36,40}_symm.yaml files select, and is the lowest level there.

Known values (checked by tests/test_bethe_ansatz.py against dense diagonalisation for L <= 16 and against SURVEY.md
Appendix B): L = 4 -> -8, L = 10 -> -18.06178541796..., L = 12 -> -21.5495636697...
"""
import numpy as np


def bethe_roots(L: int, tol: float = 1e-15, max_iter: int = 100000) -> np.ndarray:
    """ground-state rapidities of the L-site ring (L even), by damped fixed-point iteration followed by Newton steps"""
    assert L % 2 == 0 and L >= 2
    M = L // 2
    I = np.arange(M, dtype=np.float64) - (M - 1) / 2.0
    lam = 0.5 * np.tan(np.pi * I / L)  # non-interacting start
    for _ in range(max_iter):
        d = lam[:, None] - lam[None, :]
        phase = (np.pi * I + np.arctan(d).sum(axis=1)) / L
        new = 0.5 * np.tan(phase)
        step = np.abs(new - lam).max()
        lam = 0.5 * (lam + new)
        if step < 1e-10:
            break
    # polish with Newton on F_j = L * 2 atan(2 l_j) - 2 pi I_j - sum_k 2 atan(l_j - l_k)
    for _ in range(50):
        d = lam[:, None] - lam[None, :]
        F = 2 * L * np.arctan(2 * lam) - 2 * np.pi * I - 2 * np.arctan(d).sum(axis=1)
        K = 2.0 / (1.0 + d * d)
        J = K.copy()
        np.fill_diagonal(J, 0.0)
        J = np.diag(4.0 * L / (1.0 + 4.0 * lam * lam) - (K.sum(axis=1) - 2.0)) + J
        dl = np.linalg.solve(J, -F)
        lam = lam + dl
        if np.abs(dl).max() < tol:
            break
    d = lam[:, None] - lam[None, :]
    F = 2 * L * np.arctan(2 * lam) - 2 * np.pi * I - 2 * np.arctan(d).sum(axis=1)
    assert np.abs(F).max() < 1e-11, "Bethe equations did not converge"
    return lam


def ground_state_energy_SS(L: int) -> float:
    """E0 of J sum S_i . S_{i+1}, J = 1, periodic, L even"""
    lam = bethe_roots(L)
    return L / 4.0 - float(np.sum(2.0 / (4.0 * lam * lam + 1.0)))


def ground_state_energy_sigma(L: int) -> float:
    """E0 of sum sigma_i . sigma_{i+1} (the reference's units)"""
    return 4.0 * ground_state_energy_SS(L)


# Hulthen's thermodynamic limit: E0 / (J L) -> 1/4 - ln 2
E_INFINITY_PER_SITE_SS = 0.25 - np.log(2.0)

if __name__ == "__main__":
    for L in (4, 6, 8, 10, 12, 16, 24, 32, 36, 40):
        print(L, repr(ground_state_energy_sigma(L)))

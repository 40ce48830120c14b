"""The property the staged pull kernel for projected bases exploits (k_tile_pull's near window, csrc/k_pull.hip): in the sorted
array of representatives the partners rep(beta) of a row cluster around the row itself.  Measured with the oracle (this is
where the numbers quoted in DESIGN.md section 3 come from; chain_32_symm / chain_36_symm take minutes and are run by hand:
`python tests/test_partner_locality.py 32`)."""
import sys

import numpy as np
import pytest


def window_hit_fractions(L, tile=256, halos=(0, 256, 512, 1024), samples=120, seed=1):
    from oracle import c_oracle as CO
    from oracle import model as M

    o = CO.COracle(M.model_from_config(M.heisenberg_chain_config(L, symm=True)))
    reps = o.enumerate()
    n = len(reps)
    rng = np.random.RandomState(seed)
    starts = rng.randint(0, max(1, n // tile - 1), size=samples) * tile
    hits = {h: 0 for h in halos}
    total = 0
    for s in starts:
        betas, _, offs = o.apply_off_diag(reps[s:s + tile])  # raw alpha ^ flip, as the extern returns them
        rep, _, norms = o.state_info(betas)                  # K4: orbit minimum; zero-norm orbits contribute nothing
        rep = rep[norms > 0]
        j = np.searchsorted(reps, rep)
        assert np.array_equal(reps[j], rep)
        total += len(j)
        for h in halos:
            hits[h] += int(((j >= s - h) & (j < s + tile + h)).sum())
    return n, total / (len(starts) * tile), {h: hits[h] / total for h in halos}


@pytest.mark.parametrize("L,least", [(20, 0.55), (24, 0.5)])
def test_half_of_the_partners_lie_in_the_tile_neighbourhood(L, least):
    n, per_row, frac = window_hit_fractions(L, samples=40)
    assert frac[512] >= least, (n, per_row, frac)
    assert frac[0] <= frac[256] <= frac[512] <= frac[1024]


if __name__ == "__main__":
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for L in [int(a) for a in sys.argv[1:]] or [28]:
        print(L, window_hit_fractions(L))

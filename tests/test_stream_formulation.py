"""The sorted-stream packet formulation (csrc/k_packets.hip k_tile_st / k_window) restated in numpy and held
against the oracle on the CPU -- the three facts the HIP kernels rely on, none of which needs a GPU to be checked:

  1. ORDER.  On an unprojected fixed-weight basis every off-diagonal group of an exchange operator is a pair (i, j); a packet
     exists iff alpha is anti-aligned on the pair, and for a fixed pattern of alpha on it (01 or 10) beta = alpha ^ x differs
     from alpha by a CONSTANT.  The rows of a source partition ascend, the states of a destination ascend, so the packets of one
     STREAM = (pair, pattern) written in row order carry strictly ascending indices at every destination
     (the reference's packets carry the state and are looked up one by one: DistributedMatrixVector.chpl:73-127).
  2. CONSUMPTION.  A consumer that owns a window of W rows of y finds the sub-run of every stream by binary search on the keys
     and adds it: summed over the windows that is y = H x (DistributedMatrixVector.chpl:1055-1093) -- no packet is lost or
     counted twice at the window seams, whatever W.
  3. INDEX WITHOUT A SEARCH.  The destination index of beta is prefix(word of rank(beta), owner) + popcount(owner bits below)
     (the all-destinations rank directory), and for a pair on ADJACENT sites (lo, lo + 1) the colex rank of beta is the rank
     of alpha +- C(lo, k), k = set bits of alpha below lo: one binomial per packet instead of the rank sum.

Test infrastructure only: the product path is the HIP library (tests/test_gpu_matvec.py, tests/test_gpu_loopback.py hold
`tile+streams` against the atomic consumers and against this same oracle on the GPU)."""
import math

import numpy as np
import pytest

from helpers import model_config, oracle_for, oracle_reps

# (of the reference's small inputs the ones the streams are eligible for -- unprojected, fixed weight, no inversion -- plus two
# that are not: they must be skipped for the stated reason, not silently)
MODELS = ["heisenberg_chain_6", "heisenberg_chain_8", "heisenberg_chain_16", "heisenberg_kagome_12", "heisenberg_kagome_16",
          "heisenberg_chain_10", "heisenberg_chain_12"]


def _partition(name, P):
    from oracle import c_oracle as CO

    reps = oracle_reps(name)
    masks = CO.locale_idx_of(reps, P)
    return reps, masks, [np.ascontiguousarray(reps[masks == p]) for p in range(P)]


def _streams_of_source(o, reps_src, x_src, parts, P):
    """the packets one source partition writes: {(destination, pair mask, pattern)} -> (keys, values) in ROW order, keys = index of
    beta in the destination's (ascending) representatives"""
    from oracle import c_oracle as CO

    betas, cs, offs = o.apply_off_diag(reps_src, x_src)
    rows = np.repeat(np.arange(len(reps_src)), np.diff(offs))
    alphas = reps_src[rows]
    flip = alphas ^ betas
    assert np.all(np.bitwise_count(flip) == 2), "every off-diagonal term of these models is an exchange of one pair"
    lower = flip & (~flip + np.uint64(1))  # lowest set bit of the pair
    up = (alphas & lower) != 0  # the lower site's bit moves up
    dest = CO.locale_idx_of(betas, P)
    out = {}
    for d in range(P):
        sel_d = dest == d
        if not sel_d.any():
            continue
        idx = CO.state_index(parts[d], betas[sel_d])
        assert np.all(idx >= 0)
        f, u, v = flip[sel_d], up[sel_d], cs[sel_d]
        for pair in np.unique(f):
            for pattern in (False, True):
                s = (f == pair) & (u == pattern)
                if s.any():
                    out[(d, int(pair), bool(pattern))] = (idx[s], v[s])  # boolean selection keeps the row order
    return out


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("P", [2, 3, 8])
def test_streams_are_sorted_and_windows_add_up_to_the_matvec(name, P):
    o = oracle_for(name)
    cfg = model_config(name)
    if cfg["basis"].get("hamming_weight") is None or cfg["basis"].get("symmetries") or cfg["basis"].get("spin_inversion"):
        pytest.skip("the streams are for unprojected fixed-weight bases (ls_amd_internal_streams_eligible)")
    reps, masks, parts = _partition(name, P)
    rng = np.random.RandomState(7)
    x_parts = [rng.rand(len(p)) - 0.5 for p in parts]
    want = o.matvec_partitioned(parts, x_parts)
    # y starts as the diagonal (localDiagonal assigns, DMV:1062-1063), the windows add the packets
    y_parts = [o.apply_diag(parts[p], x_parts[p]) for p in range(P)]
    streams = [_streams_of_source(o, parts[q], x_parts[q], parts, P) for q in range(P)]
    n_packets = 0
    for q in range(P):
        for (d, pair, pattern), (keys, vals) in streams[q].items():
            # 1. ORDER: strictly ascending (hence distinct: a window of W rows holds at most W packets of a stream)
            assert np.all(np.diff(keys.astype(np.int64)) > 0), (name, P, q, d, hex(pair), pattern)
            n_packets += len(keys)
    assert n_packets > 0
    # window seams everywhere / in odd places / nowhere (one row per window is quadratic in python: small bases only)
    for W in ((1, 7, 64, 1 << 20) if len(reps) <= 1000 else (61, 2048, 1 << 20)):
        got = [v.copy() for v in y_parts]
        for d in range(P):
            n = len(parts[d])
            mine = [kv for q in range(P) for (dd, _pair, _pattern), kv in streams[q].items() if dd == d]
            for w0 in range(0, n, W):
                w1 = min(w0 + W, n)
                acc = np.zeros(w1 - w0)
                for keys, vals in mine:
                    lo, hi = np.searchsorted(keys, w0, side="left"), np.searchsorted(keys, w1, side="left")
                    acc[keys[lo:hi] - w0] += vals[lo:hi].real  # distinct keys inside a run: a plain indexed add
                got[d][w0:w1] += acc
        for d in range(P):
            np.testing.assert_allclose(got[d], want[d], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["heisenberg_chain_8", "heisenberg_chain_16", "heisenberg_kagome_16"])
@pytest.mark.parametrize("P", [2, 5, 8])
def test_destination_index_from_the_rank_directory(name, P):
    """index at the owner = states of that owner among the ranks below the word + owner bits below inside the word"""
    from oracle import c_oracle as CO

    reps, masks, parts = _partition(name, P)
    ranks = CO.fixed_hamming_ranks(reps)
    np.testing.assert_array_equal(ranks, np.arange(len(reps)))  # the whole fixed-weight sector, ascending = colex order
    n_words = (len(reps) + 63) // 64
    word = ranks >> 6
    prefix = np.zeros((n_words + 1, P), dtype=np.int64)  # prefix[w][d] = states of partition d with rank < 64 w
    np.add.at(prefix, (word + 1, masks), 1)
    prefix = np.cumsum(prefix, axis=0)
    bits = np.zeros((n_words, P), dtype=np.uint64)  # owner bits of every word
    np.bitwise_or.at(bits, (word, masks), np.uint64(1) << (ranks & 63).astype(np.uint64))
    o = oracle_for(name)
    betas, _cs, _offs = o.apply_off_diag(reps[:: max(1, len(reps) // 3000)])
    r = CO.fixed_hamming_ranks(betas)
    d = CO.locale_idx_of(betas, P)
    below = bits[r >> 6, d] & ((np.uint64(1) << (r & 63).astype(np.uint64)) - np.uint64(1))
    idx = prefix[r >> 6, d] + np.bitwise_count(below)
    for p in range(P):
        sel = d == p
        np.testing.assert_array_equal(idx[sel], CO.state_index(parts[p], betas[sel]))


@pytest.mark.parametrize("L,weight", [(12, 6), (16, 8), (14, 5), (20, 3)])
def test_rank_of_beta_from_the_rank_of_alpha_by_one_binomial(L, weight):
    """exchange on adjacent sites (lo, lo + 1): rank(beta) = rank(alpha) + C(lo, k) when the lower site's bit moves up,
    - C(lo, k) when it moves down, k = set bits of alpha below lo"""
    from oracle import c_oracle as CO
    from oracle import model as M

    states = []
    s = (1 << weight) - 1
    while s < (1 << L):
        states.append(s)
        s = M.next_state_fixed_hamming(s)
    states = np.array(states, dtype=np.uint64)
    if len(states) > 20000:
        states = states[:: len(states) // 20000]
    ga = CO.fixed_hamming_ranks(states)
    for lo in range(L - 1):
        x = np.uint64(3 << lo)
        act = np.bitwise_count(states & x) == 1
        a = states[act]
        up = (a >> np.uint64(lo)) & np.uint64(1)
        k = np.bitwise_count(a & np.uint64((1 << lo) - 1)).astype(np.int64)
        c = np.array([math.comb(lo, int(kk)) for kk in k], dtype=np.int64)
        want = CO.fixed_hamming_ranks(a ^ x)
        np.testing.assert_array_equal(np.where(up == 1, ga[act] + c, ga[act] - c), want)

"""Parity at the BASELINE.json shapes that have no reference golden offline ("next round" item 1):

  config 3  heisenberg_chain_32, full size: >= 1e5 rows recomputed by the oracle, chosen to hit every tile-edge class of the
            staged row kernel (row mod 1024 in {0, 1, 511, 512, 1023}), the first / last 2048 rows (window halo clipped at
            the ends of x) and waves that straddle two high parts; ranks by the oracle's own combinadic rank.
  config 4  heisenberg_chain_36_symm hash-partitioned into 8 partitions, packets path, against the one-partition result and
            against oracle-recomputed rows.
  config 5  heisenberg_chain_40_symm: representative count (Burnside), push == pull, Hermiticity,
            >= 1e4 oracle-recomputed rows, and a Lanczos ground state whose residual is checked with the OTHER kernel family.
  fixture   the HIP enumeration reproduces /root/reference/v1/error.chpl:21 (the one in-tree artefact of the reference
            that pins representatives) directly, not only through the oracle.

Oracle rows use the pull form of the reference's row expansion: with M[rep(beta_j), alpha] = c_j character_j n(beta_j) / n(alpha)
(BatchedOperator.chpl:184-203) and H Hermitian,  y[alpha] = d(alpha) x[alpha] + sum_j conj(M[rep(beta_j), alpha]) x[rep(beta_j)].
"""
import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t

    if not t.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    t.cuda.set_device(0)
    return t


def oracle_rows(torch, o, reps_t, rows, x_t, projected, rank_fn=None):
    """y at `rows` (indices into the ascending device array reps_t) recomputed on the CPU by the oracle."""
    rows_t = torch.from_numpy(np.asarray(rows, dtype=np.int64)).cuda()
    alphas = reps_t[rows_t].cpu().numpy().view(np.uint64)
    betas, cs, offs = o.apply_off_diag(alphas)
    if projected:
        reps_b, chars_b, norms_b = o.state_info(betas)
        _, _, norms_a = o.state_info(alphas)
        counts = np.diff(offs)
        na = np.repeat(norms_a, counts)
        coef = np.conj(cs * chars_b) * norms_b / na
        keep = norms_b > 0
    else:
        reps_b, coef, keep = betas, np.conj(cs), np.ones(len(betas), dtype=bool)
    if rank_fn is not None:
        idx = rank_fn(reps_b)
    else:
        idx = torch.searchsorted(reps_t, torch.from_numpy(reps_b.view(np.int64)).cuda()).cpu().numpy()
        idx = np.minimum(idx, reps_t.numel() - 1)
    idx_t = torch.from_numpy(idx.astype(np.int64)).cuda()
    found = reps_t[idx_t].cpu().numpy().view(np.uint64) == reps_b
    assert bool(np.all(found | ~keep)), "oracle generated a state outside the basis"
    xb = x_t[idx_t].cpu().numpy()
    terms = np.where(keep, coef * xb, 0.0)
    # np.add.reduceat misbehaves on empty segments: use a cumulative sum
    csum = np.concatenate([[0.0], np.cumsum(terms)])
    off_part = csum[offs[1:len(alphas) + 1]] - csum[offs[:len(alphas)]]
    want = o.apply_diag(alphas) * x_t[rows_t].cpu().numpy() + off_part
    return rows_t, (want.real if not x_t.is_complex() else want)


def assert_rows(got, want, what, rtol=1e-12):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale, f"{what}: max |dy| = {err:.3e} (scale {scale:.3e})"


def test_hip_enumeration_equals_reference_fixture(torch):
    """/root/reference/v1/error.chpl:21: the 13 representatives of the 10-site chain (hamming weight 5, translation +
    reflection + spin inversion), produced by k_enum_flags / k_enum_write."""
    import distributed_matvec_amd as D
    from oracle import model as M

    want = golden()["fixtures"]["v1_error_chpl_21_representatives"]
    assert want == [31, 47, 55, 87, 91, 93, 103, 107, 155, 171, 173, 179, 341]
    basis = D.loadConfigFromDict(M.heisenberg_chain_config(10, symm=True))
    basis = basis[0] if isinstance(basis, tuple) else basis
    for P in (1, 2, 3):
        reps, masks = D.enumerateStates(basis, P)
        got = D.arrFromHashedToBlock(reps, masks).cpu().numpy().view(np.uint64)
        assert got.tolist() == want
        for p in range(P):  # every partition ascending and owned by hash64_01 % P
            part = reps[p].cpu().numpy().view(np.uint64).tolist()
            assert part == [s for s in want if D.localeIdxOf(s, P) == p]
    # ls_hs_basis_build -> the registered enumerate_states kernel -> the same HIP enumeration
    basis.build()
    assert basis.representatives().tolist() == want


def test_chain_32_edge_targeted_rows(torch):
    """BASELINE config[2] at full size: 1e5+ oracle rows aimed at the staged kernel's seams."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from oracle import c_oracle as CO
    from oracle import model as M

    cfg = config.heisenberg_chain_config(32)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    r = reps[0]
    n = r.numel()
    assert n == 601080390
    u = D.fillRandom(r, 1, torch.float64)
    y = torch.empty_like(u)
    pl = D.matrixVectorProduct(h, [u], [y], reps, mode="pull")
    assert pl.kernel == "direct-pull+staged"
    rs = np.random.RandomState(2024)
    tiles = rs.randint(0, n // 1024, size=20000).astype(np.int64)
    rows = [tiles * 1024 + k for k in (0, 1, 511, 512, 1023)]
    rows.append(np.arange(0, 2048, dtype=np.int64))
    rows.append(np.arange(n - 2048, n, dtype=np.int64))
    # waves (64 consecutive rows) whose first and last state differ above bit 12: the per-lane fallback of the far pairs
    first, last = r[0:n - 63:64], r[63:n:64]
    straddle = ((first >> 12) != (last >> 12)).nonzero(as_tuple=True)[0]
    assert straddle.numel() > 1000
    pick = straddle[torch.from_numpy(rs.randint(0, straddle.numel(), size=600)).cuda()].cpu().numpy() * 64
    rows.append((pick[:, None] + np.arange(64)[None, :]).ravel())
    # tiles at the boundary between two XCD lists / the last (partial) tile
    last_tile = (n // 1024) * 1024
    rows.append(np.arange(last_tile - 1024, n, dtype=np.int64))
    rows = np.unique(np.concatenate(rows))
    rows = rows[(rows >= 0) & (rows < n)]
    assert len(rows) >= 100000
    o = CO.COracle(M.model_from_config(cfg))
    rows_t, want = oracle_rows(torch, o, r, rows, u, projected=False, rank_fn=CO.fixed_hamming_ranks)
    assert_rows(y[rows_t].cpu().numpy(), want, f"chain_32 {len(rows)} edge-targeted rows")
    # the same rows through the generic row kernel and through the push (atomics) formulation
    import os

    os.environ["LS_AMD_ROW_KERNEL"] = "generic"
    try:
        y2 = torch.empty_like(u)
        pl2 = D.MatvecPlan(h, reps, torch.float64, mode="pull")
        assert pl2.kernel.startswith("direct-pull") and "staged" not in pl2.kernel
        pl2.matvec([u], [y2])
        pl2.destroy()
    finally:
        del os.environ["LS_AMD_ROW_KERNEL"]
    assert float((y - y2).abs().max()) <= 1e-12 * float(y.abs().max())
    # the rows around every point where the bits above the low 12 change (where the wave-uniform far-pair split of the staged
    # kernel changes inside a wave) -- a sample of them against the oracle
    hi = r >> 12
    edges = (hi[1:] != hi[:-1]).nonzero(as_tuple=True)[0]
    pick = edges[torch.from_numpy(rs.randint(0, edges.numel(), size=8000)).cuda()].cpu().numpy()
    rows = np.unique(np.concatenate([pick - 1, pick, pick + 1, pick + 2]))
    rows = rows[(rows >= 0) & (rows < n)]
    rows_t, want = oracle_rows(torch, o, r, rows, u, projected=False, rank_fn=CO.fixed_hamming_ranks)
    assert_rows(y[rows_t].cpu().numpy(), want, f"chain_32 {len(rows)} block-edge rows")


def test_chain_36_symm_eight_partitions_packets(torch):
    """BASELINE config[3]: hash-partitioned over 8 (logical) partitions, (sigma, c x) packets, vs one partition and
    vs the oracle."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from oracle import c_oracle as CO
    from oracle import model as M

    cfg = config.heisenberg_chain_config(36, symm=True)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    P = 8
    reps8, masks = D.enumerateStates(basis, P)
    r_block = D.arrFromHashedToBlock(reps8, masks)
    n = r_block.numel()
    assert n == 63068876
    assert bool((r_block[1:] > r_block[:-1]).all())
    for p in range(P):  # ownership: every state sits in partition hash64_01 % 8 (sampled exactly on the host)
        part = reps8[p]
        assert bool((part[1:] > part[:-1]).all())
        sample = part[:: max(1, part.numel() // 2000)].cpu().numpy().view(np.uint64)
        assert np.all(CO.locale_idx_of(sample, P) == p)
    x_block = D.fillRandom(r_block, 5, torch.float64)
    x8 = D.arrFromBlockToHashed(x_block, masks, P)
    y8 = [torch.full_like(v, 7.0) for v in x8]
    pl = D.matrixVectorProduct(h, x8, y8, reps8)
    assert pl.kernel in ("tile", "tile+streams")
    got = D.arrFromHashedToBlock(y8, masks)
    y1 = torch.empty_like(x_block)
    pl1 = D.matrixVectorProduct(h, [x_block], [y1], [r_block], mode="pull")
    assert pl1.kernel == "tile-pull+indexed"
    scale = float(y1.abs().max())
    assert float((got - y1).abs().max()) <= 1e-12 * scale
    o = CO.COracle(M.model_from_config(cfg))
    rs = np.random.RandomState(36)
    rows = np.unique(np.concatenate([rs.randint(0, n, size=20000), np.arange(256), np.arange(n - 256, n)]))
    rows_t, want = oracle_rows(torch, o, r_block, rows, x_block, projected=True)
    assert_rows(got[rows_t].cpu().numpy(), want, "chain_36_symm P=8 packets vs oracle rows")


def test_chain_40_symm_properties_and_ground_state(torch):
    """BASELINE config[4]: the 40-site ring in its fully symmetric sector feeding the eigensolver
    (/root/reference/src/Diagonalize.chpl:258-332)."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from distributed_matvec_amd.diagonalize import LocalOperator, lanczos_smallest
    from oracle import c_oracle as CO
    from oracle import model as M

    cfg = config.heisenberg_chain_config(40, symm=True)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    r = reps[0]
    n = r.numel()
    assert n == 861725794  # Burnside count, SURVEY.md Appendix B
    assert bool((r[1:] > r[:-1]).all())
    o = CO.COracle(M.model_from_config(cfg))
    rs = np.random.RandomState(40)
    sample = np.unique(rs.randint(0, n, size=5000))
    flags, norms = o.is_representative(r[torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64))
    assert np.all(flags == 1) and np.all(norms > 0)
    u = D.fillRandom(r, 3, torch.float64)
    a = torch.empty_like(u)
    pull = D.MatvecPlan(h, reps, torch.float64, mode="pull")
    assert pull.kernel == "tile-pull+indexed"
    pull.matvec([u], [a])
    rows = np.unique(np.concatenate([rs.randint(0, n, size=12000), np.arange(256), np.arange(n - 256, n)]))
    rows_t, want = oracle_rows(torch, o, r, rows, u, projected=True)
    assert_rows(a[rows_t].cpu().numpy(), want, "chain_40_symm pull vs oracle rows")
    # the slot cache at this shape (88.5 GB of packet streams): the gather-only matvec == the matrix-free one on ALL rows, for the
    # vector that resolved the streams and for another one
    cached = D.MatvecPlan(h, reps, torch.float64, mode="pull")
    if cached.cache_slots(120 << 30) == n:
        assert cached.kernel == "tile-pull+indexed+cached" and cached.slot_cache[1] <= 100e9
        w = D.fillRandom(r, 5, torch.float64)
        c = torch.empty_like(u)
        cached.matvec([w], [c])  # resolves
        cached.matvec([u], [c])  # gathers only
        assert float((c - a).abs().max()) <= 1e-12 * float(a.abs().max())
        del w, c
    cached.destroy()
    torch.cuda.empty_cache()
    # Hermiticity <v, H u> == <H v, u>
    v = D.fillRandom(r, 4, torch.float64)
    c = torch.empty_like(u)
    pull.matvec([v], [c])
    lhs, rhs = float(torch.dot(v, a)), float(torch.dot(c, u))
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs))
    del c, v
    # push (the reference's formulation: packets + atomics) == pull
    b = torch.zeros_like(u)
    push = D.MatvecPlan(h, reps, torch.float64, mode="push")
    assert push.kernel in ("tile", "tile+streams")
    push.matvec([u], [b])
    assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())
    del a, b, u
    torch.cuda.empty_cache()
    # eigensolver: thick-restart Lanczos on the pull kernel; the eigenpair's residual is then measured with the PUSH kernel
    pull.destroy()
    # (what diagonalize() does: whatever HBM is left next to the Krylov basis holds resolved packet streams -- every row of this
    # basis needs 88.5 GB; a prefix of the rows if less is free, the others stay matrix-free)
    free_bytes, _ = torch.cuda.mem_get_info()
    budget = max(0, int(free_bytes) - (12 + 6) * n * 8 - (8 << 30))
    op = LocalOperator(h, reps, torch.float64, mode="pull", slot_cache_bytes=budget)
    assert op.cached_rows % 256 == 0 or op.cached_rows == n
    res = lanczos_smallest(op, num_evals=1, eps=1e-6, max_basis=12, max_restarts=60)
    e0 = res.eigenvalues[0]
    vec = res.eigenvectors[0]
    hv = torch.zeros_like(vec)
    push.matvec([vec], [hv])
    resid = float((hv - e0 * vec).norm()) / float(vec.norm())
    assert res.converged, res.history[-3:]
    assert resid <= 1e-4 * abs(e0), (e0, resid)
    rq = float(torch.dot(vec, hv)) / float(torch.dot(vec, vec))
    assert abs(rq - e0) <= 1e-8 * abs(e0)
    # the Bethe-ansatz ground-state energy of the 40-site ring (oracle/bethe.py: 20 coupled Bethe equations, independent of
    # every term table, projector and oracle of this repository) -- E0 of the trivial sector, 1e-8 relative
    from oracle import bethe

    want = bethe.ground_state_energy_sigma(40)
    assert abs(e0 - want) <= 1e-8 * abs(want), (e0, want)
    # hand the HBM back before the next test: the plans' buffers are the library's own allocations, the Krylov basis sits in
    # torch's cache, and the eight-rank tests below allocate through both
    op.plan.destroy()
    push.destroy()
    del op, res, vec, hv
    torch.cuda.empty_cache()


def test_chain_32_and_36_symm_complex_vectors(torch):
    """the north-star dtype (complex128) at the full BASELINE shapes: the c128 instantiation of the staged kernel on
    chain_32 and the projected-basis pull kernel on chain_36_symm against oracle-recomputed rows (the operator is real, so
    this is also H(x_re) + i H(x_im) against the f64 kernels)"""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from oracle import c_oracle as CO
    from oracle import model as M

    for L, symm, kernel in ((32, False, "direct-pull+staged"), (36, True, "tile-pull+indexed")):
        cfg = config.heisenberg_chain_config(L, symm=symm)
        basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
        reps, masks = D.enumerateStates(basis, 1)
        r = reps[0]
        n = r.numel()
        xc = D.fillRandom(r, 21, torch.complex128)
        yc = torch.empty_like(xc)
        pl = D.MatvecPlan(h, reps, torch.complex128, mode="pull")
        assert pl.kernel == kernel
        pl.matvec([xc], [yc])
        pl.destroy()
        o = CO.COracle(M.model_from_config(cfg))
        rs = np.random.RandomState(L)
        rows = np.unique(np.concatenate([rs.randint(0, n, size=20000), np.arange(1024), np.arange(n - 1024, n),
                                         (rs.randint(0, n // 512, size=2000) * 512)]))
        rows = rows[rows < n]
        rows_t, want = oracle_rows(torch, o, r, rows, xc, projected=symm, rank_fn=None if symm else CO.fixed_hamming_ranks)
        assert_rows(yc[rows_t].cpu().numpy(), want, f"chain_{L}{'_symm' if symm else ''} c128 rows")
        # real and imaginary parts through the f64 kernels
        xr, xi = xc.real.contiguous(), xc.imag.contiguous()
        yr, yi = torch.empty_like(xr), torch.empty_like(xi)
        plr = D.MatvecPlan(h, reps, torch.float64, mode="pull")
        plr.matvec([xr], [yr])
        plr.matvec([xi], [yi])
        plr.destroy()
        scale = float(yc.abs().max())
        assert float((yc.real - yr).abs().max()) <= 1e-12 * scale and float((yc.imag - yi).abs().max()) <= 1e-12 * scale
        del xc, yc, xr, xi, yr, yi
        torch.cuda.empty_cache()


def test_chain_40_symm_complex_vectors(torch):
    """the north-star dtype on BASELINE config[4]/[5]: complex128 vectors on all 861 725 794 representatives of the 40-site ring --
    oracle-recomputed rows, and H(x_re) + i H(x_im) through the f64 kernel on every row"""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from oracle import c_oracle as CO
    from oracle import model as M

    cfg = config.heisenberg_chain_config(40, symm=True)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    r = reps[0]
    n = r.numel()
    xc = D.fillRandom(r, 23, torch.complex128)
    yc = torch.empty_like(xc)
    pl = D.MatvecPlan(h, reps, torch.complex128, mode="pull")
    assert pl.kernel == "tile-pull+indexed"
    pl.matvec([xc], [yc])
    pl.destroy()
    o = CO.COracle(M.model_from_config(cfg))
    rs = np.random.RandomState(41)
    rows = np.unique(np.concatenate([rs.randint(0, n, size=4000), np.arange(256), np.arange(n - 256, n)]))
    rows_t, want = oracle_rows(torch, o, r, rows, xc, projected=True)
    assert_rows(yc[rows_t].cpu().numpy(), want, "chain_40_symm c128 rows")
    plr = D.MatvecPlan(h, reps, torch.float64, mode="pull")
    scale = float(yc.abs().max())
    for part in ("real", "imag"):
        xp = getattr(xc, part).contiguous()
        yp = torch.empty_like(xp)
        plr.matvec([xp], [yp])
        assert float((getattr(yc, part) - yp).abs().max()) <= 1e-12 * scale, part
        del xp, yp
    plr.destroy()
    del xc, yc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["packets", "replicated"])
def test_chain_36_symm_eight_ranks(torch, mode):
    """BASELINE config[3] as it will run on a node: heisenberg_chain_36_symm hash-partitioned over EIGHT ranks, the exchange
    inside the C host (ls_amd_dist_matvec: all-to-all-v of packets; ls_amd_repl_matvec: blocks of x).  One GPU here, so the
    ranks are host threads over the loop-back transport (tests/test_gpu_loopback.py); result vs the one-partition kernel and
    vs oracle-recomputed rows."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator
    from oracle import c_oracle as CO
    from oracle import model as M
    from test_gpu_loopback import _run_ranks

    cfg = config.heisenberg_chain_config(36, symm=True)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    P = 8
    reps, masks = D.enumerateStates(basis, P)
    r_block = D.arrFromHashedToBlock(reps, masks)
    n = r_block.numel()
    assert n == 63068876
    xs = [D.fillRandom(reps[p], 5, torch.float64) for p in range(P)]
    ys = [torch.full_like(v, 2.0) for v in xs]

    def body(rank, comm):
        if mode == "packets":
            op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm, num_rounds=2)
        else:
            op = RcclReplicatedOperator(h, r_block, masks, torch.float64, comm=comm)
        op.matvec(xs[rank], ys[rank], check=True)
        op.dm.destroy() if mode == "packets" else op.rm.destroy()

    comms = _run_ranks(P, body)
    for c in comms:
        c.destroy()
    got = D.arrFromHashedToBlock(ys, masks)
    x_block = D.arrFromHashedToBlock(xs, masks)
    y1 = torch.empty_like(x_block)
    pl1 = D.MatvecPlan(h, [r_block], torch.float64, mode="pull")
    pl1.matvec([x_block], [y1])
    pl1.destroy()
    assert float((got - y1).abs().max()) <= 1e-12 * float(y1.abs().max())
    o = CO.COracle(M.model_from_config(cfg))
    rows = np.unique(np.random.RandomState(8).randint(0, n, size=10000))
    rows_t, want = oracle_rows(torch, o, r_block, rows, x_block, projected=True)
    assert_rows(got[rows_t].cpu().numpy(), want, f"chain_36_symm, 8 ranks, {mode}")


def test_chain_40_symm_eight_ranks_packets(torch):
    """BASELINE config[4] shape across eight ranks (loop-back transport): one matvec of heisenberg_chain_40_symm through
    ls_amd_dist_matvec == the one-partition pull kernel"""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from distributed_matvec_amd.distributed import RcclDistributedOperator
    from test_gpu_loopback import _run_ranks

    torch.cuda.empty_cache()  # (the library allocates outside torch's cache)
    basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(40, symm=True), hamiltonian=True)
    P = 8
    reps, masks = D.enumerateStates(basis, P)
    assert int(masks.numel()) == 861725794
    xs = [D.fillRandom(reps[p], 9, torch.float64) for p in range(P)]
    ys = [torch.full_like(v, -1.0) for v in xs]
    rounds = [None] * P

    def body(rank, comm):
        op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm)
        rounds[rank] = op.num_rounds
        op.matvec(xs[rank], ys[rank], check=True)
        op.dm.destroy()

    for c in _run_ranks(P, body):
        c.destroy()
    assert len(set(rounds)) == 1 and rounds[0] >= 7  # 108 M rows per rank at 2^24 rows per round
    got = D.arrFromHashedToBlock(ys, masks)
    del ys
    x_block = D.arrFromHashedToBlock(xs, masks)
    del xs
    r_block = D.arrFromHashedToBlock(reps, masks)
    del reps
    torch.cuda.empty_cache()
    y1 = torch.empty_like(x_block)
    pl1 = D.MatvecPlan(h, [r_block], torch.float64, mode="pull")
    pl1.matvec([x_block], [y1])
    pl1.destroy()
    assert float((got - y1).abs().max()) <= 1e-12 * float(y1.abs().max())


def test_distributed_eigensolve_eight_ranks(torch):
    """the caller of config[4] (/root/reference/src/Diagonalize.chpl:134-225): thick-restart Lanczos in lock-step on eight
    ranks -- matvec = ls_amd_dist_matvec, dot products = globalSumReal over the communicator -- on heisenberg_chain_32_symm;
    E0 equals the single-device solve."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from distributed_matvec_amd.diagonalize import LocalOperator, RankOperator, lanczos_smallest
    from distributed_matvec_amd.distributed import RcclDistributedOperator
    from test_gpu_loopback import _run_ranks

    basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(32, symm=True), hamiltonian=True)
    P = 8
    reps, masks = D.enumerateStates(basis, P)
    assert int(masks.numel()) == 4707969
    e0 = [None] * P

    def body(rank, comm):
        op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm)
        res = lanczos_smallest(RankOperator(op, reps[rank], torch.float64), num_evals=1, eps=1e-9, max_basis=16)
        assert res.converged
        e0[rank] = res.eigenvalues[0]
        op.dm.destroy()

    for c in _run_ranks(P, body):
        c.destroy()
    r_block = D.arrFromHashedToBlock(reps, masks)
    single = lanczos_smallest(LocalOperator(h, [r_block], torch.float64), num_evals=1, eps=1e-9, max_basis=16)
    assert single.converged
    for e in e0:
        assert abs(e - single.eigenvalues[0]) <= 1e-7 * abs(single.eigenvalues[0])
    from oracle import bethe

    want = bethe.ground_state_energy_sigma(32)  # Bethe ansatz: nothing of this repository went into this number
    assert abs(single.eigenvalues[0] - want) <= 1e-8 * abs(want), (single.eigenvalues[0], want)
    for e in e0:
        assert abs(e - want) <= 1e-7 * abs(want)


@pytest.mark.parametrize("L", [24, 36])
def test_symmetric_chain_ground_state_equals_bethe_ansatz(torch, L):
    """E0 of heisenberg_chain_{24,36}_symm (trivial sector of the 4L-element group) from the HIP path -- enumeration,
    projected-basis pull kernel, device-resident Lanczos -- against the Bethe-ansatz energy of the L-site ring
    (oracle/bethe.py), 1e-8 relative.  With chain_32_symm (eight-rank test above) and chain_40_symm this covers every
    symmetric BASELINE shape; the number owes nothing to the term compiler, the K4 projection or either oracle."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import config
    from distributed_matvec_amd.diagonalize import LocalOperator, lanczos_smallest
    from oracle import bethe

    basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(L, symm=True), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    assert int(masks.numel()) == {24: 28968, 36: 63068876}[L]
    res = lanczos_smallest(LocalOperator(h, reps, torch.float64), num_evals=1, eps=1e-7, max_basis=16, max_restarts=200)
    assert res.converged, res.history[-2:]
    want = bethe.ground_state_energy_sigma(L)
    assert abs(res.eigenvalues[0] - want) <= 1e-8 * abs(want), (res.eigenvalues[0], want)
    # the same solve with the slot cache (what diagonalize() runs when HBM has room: the first matvec resolves the packet
    # streams, the other ~100 only gather): same pin, same number of matvecs
    matvecs = res.matvecs
    del res
    torch.cuda.empty_cache()
    op = LocalOperator(h, reps, torch.float64, slot_cache_bytes=64 << 30)
    assert op.cached_rows == int(masks.numel()) and op.plan.kernel == "tile-pull+indexed+cached"
    assert 0 < op.plan.slot_cache[1] <= 25 * 5 * int(masks.numel())  # exact stream lengths: <= 20 packets per row on average, 5 B each
    res = lanczos_smallest(op, num_evals=1, eps=1e-7, max_basis=16, max_restarts=200)
    assert res.converged and abs(res.matvecs - matvecs) <= 16
    assert abs(res.eigenvalues[0] - want) <= 1e-8 * abs(want), (res.eigenvalues[0], want)

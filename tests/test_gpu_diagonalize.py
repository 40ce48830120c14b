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


@pytest.mark.parametrize("name,exchange", [("heisenberg_chain_12", "replicated"), ("heisenberg_chain_24_symm", "replicated"),
                                           ("heisenberg_chain_12", "packets"), ("heisenberg_chain_24_symm", "auto"),
                                           ("heisenberg_chain_24_symm", "auto-ceiling")])
def test_diagonalize_distributed_one_rank(torch, tmp_path, name, exchange, monkeypatch, capfd):
    """Diagonalize.main with one process per GPU (diagonalize_distributed) on the one rank this box has: the C host's
    exchange with a single rank, RCCL reductions, and the block-distributed output file == what the one-process driver
    writes.  (With more ranks the same code runs under torchrun; the layout conversion and the per-rank hyperslab I/O are
    covered at world_size 2 and 3 by tests/test_distributed_gloo.py.)"""
    import os

    import torch.distributed as dist

    from distributed_matvec_amd import hdf5
    from distributed_matvec_amd.diagonalize import diagonalize, diagonalize_distributed

    try:
        hdf5.lib()
    except hdf5.Hdf5Unavailable:
        pytest.skip("libhdf5 not available")
    port = 29700 + os.getpid() % 200
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    if exchange == "auto-ceiling":  # the O(N) tables of the replicated-x exchange "do not fit": the packets take over by themselves
        monkeypatch.setenv("LS_AMD_EXCHANGE_HBM_CEILING", "100000")
    try:
        out = str(tmp_path / "distributed.h5")
        r = diagonalize_distributed(model_config(name), num_evals=1, eps=1e-10, output=out, exchange=exchange.split("-")[0],
                                    verbose=exchange.startswith("auto"))
    finally:
        dist.destroy_process_group()
    if exchange.startswith("auto"):
        assert f"exchange = {'packets' if exchange == 'auto-ceiling' else 'replicated'}" in capfd.readouterr().out
    ref_out = str(tmp_path / "one_process.h5")
    ref = diagonalize(model_config(name), num_evals=1, eps=1e-10, output=ref_out)
    assert r.converged and abs(r.eigenvalues[0] - ref.eigenvalues[0]) < 1e-9
    assert np.array_equal(hdf5.read_dataset(out, "/basis/representatives"), hdf5.read_dataset(ref_out, "/basis/representatives"))
    v, w = hdf5.read_dataset(out, "/hamiltonian/eigenvectors")[0], hdf5.read_dataset(ref_out, "/hamiltonian/eigenvectors")[0]
    assert v.shape == w.shape and abs(abs(np.dot(v, w)) - 1.0) < 1e-8
    assert abs(hdf5.read_dataset(out, "/hamiltonian/eigenvalues")[0] - r.eigenvalues[0]) < 1e-12


def test_diagonalize_distributed_extends_an_existing_output(torch, tmp_path, capfd):
    """makeBasisStates (Diagonalize.chpl:227-246) with one process per GPU: an output file that already holds
    /basis/representatives (and anything else) is extended, not truncated -- the stored states are read block-wise, checked
    against the configured basis and kept; hamiltonian/* of an earlier run is replaced; a file of another basis is refused"""
    import os

    import torch.distributed as dist

    from distributed_matvec_amd import api, hdf5
    from distributed_matvec_amd.diagonalize import diagonalize, diagonalize_distributed

    try:
        hdf5.lib()
    except hdf5.Hdf5Unavailable:
        pytest.skip("libhdf5 not available")
    out = str(tmp_path / "out.h5")
    ref = diagonalize(model_config("heisenberg_chain_12"), num_evals=1, eps=1e-10, output=out)  # the earlier run: basis + hamiltonian
    reps = hdf5.read_dataset(out, "/basis/representatives")
    hdf5.write_datasets(out, {"/extra/payload": np.arange(7, dtype=np.float64)}, append=True)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 200}", rank=0, world_size=1)
    try:
        r = diagonalize_distributed(model_config("heisenberg_chain_12"), num_evals=1, eps=1e-10, output=out, verbose=True)
        assert "verified, kept" in capfd.readouterr().out
        assert r.converged and abs(r.eigenvalues[0] - ref.eigenvalues[0]) < 1e-9
        assert np.array_equal(hdf5.read_dataset(out, "/basis/representatives"), reps)
        assert np.array_equal(hdf5.read_dataset(out, "/extra/payload"), np.arange(7, dtype=np.float64))
        assert hdf5.read_dataset(out, "/hamiltonian/eigenvectors").shape == (1, len(reps))
        assert abs(hdf5.read_dataset(out, "/hamiltonian/eigenvalues")[0] - r.eigenvalues[0]) < 1e-12
        # the same file handed to another sector / another model: refused, and left as it was
        for stale in ("heisenberg_kagome_12_symm", "heisenberg_chain_10"):
            with pytest.raises(api.LsAmdError, match="does not belong to the configured basis"):
                diagonalize_distributed(model_config(stale), num_evals=1, eps=1e-10, output=out)
        assert np.array_equal(hdf5.read_dataset(out, "/basis/representatives"), reps)
        # same count, one state different
        bad = reps.copy()
        bad[len(bad) // 2] ^= np.uint64(3)
        out2 = str(tmp_path / "bad.h5")
        hdf5.write_datasets(out2, {"/basis/representatives": bad})
        with pytest.raises(api.LsAmdError, match="stale output file"):
            diagonalize_distributed(model_config("heisenberg_chain_12"), num_evals=1, eps=1e-10, output=out2)
    finally:
        dist.destroy_process_group()


def test_square_4x4_published_energy(torch):
    """the 4 x 4 periodic square lattice (data/heisenberg_square_4x4.yaml: 107 symmetry-adapted states, a non-cyclic lattice
    group -- the general K4 path -- with vertical and wrap-around bonds) against the published
    E0 / N = -0.7017802 J (Schulz, Ziman, Poilblanc, PRB 54, 12946 (1996)) -- a two-dimensional third-party pin next to
    the Bethe-ansatz energies of the rings"""
    from distributed_matvec_amd.diagonalize import diagonalize

    r = diagonalize(model_config("heisenberg_square_4x4"), num_evals=1, eps=1e-10)
    assert r.converged and abs(r.eigenvalues[0] / (4 * 16) - (-0.7017802)) < 5e-8, r.eigenvalues
    # the same lattice without any symmetry: all 12 870 S^z = 0 states through the generic row kernel (k_direct); the global
    # ground state lives in the symmetric sector, so the energy is the same
    import copy

    cfg = copy.deepcopy(model_config("heisenberg_square_4x4"))
    cfg["basis"]["symmetries"] = []
    cfg["basis"]["spin_inversion"] = None
    full = diagonalize(cfg, num_evals=1, eps=1e-10)
    assert full.converged and abs(full.eigenvalues[0] / (4 * 16) - (-0.7017802)) < 5e-8, full.eigenvalues


def test_square_6x6_published_energy(torch):
    """The reference's own benchmark model (`make benchmark-*`, /root/reference/Makefile:86,109: data/heisenberg_square_6x6.yaml,
    36 sites, the 288-element space group of the periodic 6 x 6 square lattice x the global spin flip; 15.8 million
    symmetry-adapted states): enumeration on the GPU, the projected pull kernel with K4 in its lattice-group form (36
    translations as bit operations x 8 point-group networks), thick-restart Lanczos -- against the published
    E0 / N = -0.678872 J (Schulz, Ziman, Poilblanc, PRB 54, 12946 (1996))."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd.diagonalize import diagonalize

    cfg = model_config("heisenberg_square_6x6")
    basis = D.loadConfigFromDict(cfg)
    reps, _ = D.enumerateStates(basis, 1)
    n = int(reps[0].numel())
    assert 15_000_000 < n < 16_500_000, n  # C(36, 18) / 576 = 15.76 M plus the orbits with a non-trivial stabiliser
    r = diagonalize(cfg, num_evals=1, eps=1e-9, max_basis=40)
    assert r.converged and abs(r.eigenvalues[0] / (4 * 36) - (-0.678872)) < 2e-6, r.eigenvalues

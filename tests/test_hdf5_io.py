"""HDF5 layer (the reference's disk format either side of the path) + the mirrored integration tests."""
import io
import os

import numpy as np
import pytest

from helpers import SMALL_MODELS, golden_vectors, model_config


def _hdf5():
    from distributed_matvec_amd import hdf5

    try:
        hdf5.lib()
    except hdf5.Hdf5Unavailable:
        pytest.skip("libhdf5 not available")
    return hdf5


def test_round_trip(tmp_path):
    hdf5 = _hdf5()
    rs = np.random.RandomState(0)
    x = rs.rand(1, 1234)
    reps = (rs.randint(0, 2**62, size=77, dtype=np.int64)).astype(np.uint64)
    path = str(tmp_path / "a.h5")
    hdf5.write_datasets(path, {"/x": x, "/representatives": reps, "/hamiltonian/eigenvalues": np.array([-1.0, 2.0])})
    assert hdf5.dataset_shape(path, "/x") == (1, 1234)
    assert np.array_equal(hdf5.read_dataset(path, "/x"), x)
    got = hdf5.read_dataset(path, "/representatives")
    assert got.dtype == np.uint64 and np.array_equal(got, reps)
    assert np.array_equal(hdf5.read_dataset(path, "/hamiltonian/eigenvalues"), [-1.0, 2.0])
    with pytest.raises(KeyError):
        hdf5.read_dataset(path, "/nope")


def test_block_distributed_io(tmp_path):
    """MyHDF5.chpl:105-144, 214-253, 272-333: every locale reads / writes its own hyperslab.  Chapel's Block distribution of the
    last dimension (locale p owns floor(i P / n) == p), rank 1 and rank 2, uneven splits, more locales than elements, and the
    reference's halts on a rank or element-type mismatch."""
    hdf5 = _hdf5()
    for n, P in ((10, 3), (7, 7), (3, 5), (1000, 8), (0, 2)):
        owners = [i * P // n for i in range(n)]
        for p in range(P):
            lo, hi = hdf5.block_range(n, P, p)
            assert [i for i in range(n) if owners[i] == p] == list(range(lo, hi))
    rs = np.random.RandomState(3)
    path = str(tmp_path / "blocks.h5")
    n = 1003
    x = rs.rand(2, n)
    reps = rs.randint(0, 2**62, size=n, dtype=np.int64).astype(np.uint64)
    hdf5.write_datasets(path, {"/x": x, "/basis/representatives": reps})
    for P in (1, 3, 8):
        blocks = hdf5.read_dataset_as_blocks(path, "/x", P)
        assert [b.shape for b in blocks] == [(2, hdf5.block_range(n, P, p)[1] - hdf5.block_range(n, P, p)[0]) for p in range(P)]
        assert np.array_equal(np.concatenate(blocks, axis=1), x)
        rb = hdf5.read_dataset_as_blocks(path, "/basis/representatives", P, dtype=np.uint64)
        assert rb[0].dtype == np.uint64 and np.array_equal(np.concatenate(rb), reps)
        assert np.array_equal(hdf5.read_dataset_block(path, "/x", P, P - 1), blocks[-1])
        # every "locale" writes its own block; the file then holds the whole vector
        hdf5.write_dataset_as_blocks(path, f"/hamiltonian/eigenvectors_{P}", [2.0 * b for b in blocks])
        assert np.array_equal(hdf5.read_dataset(path, f"/hamiltonian/eigenvectors_{P}"), 2.0 * x)
    # an arbitrary chunk, and an existing dataset replaced by create_dataset
    assert np.array_equal(hdf5.read_dataset_chunk(path, "/x", (1, 17), (1, 40)), x[1:2, 17:57])
    hdf5.create_dataset(path, "/y", (1, n))
    hdf5.write_dataset_chunk(path, "/y", (0, 100), x[:1, 100:300])
    got = hdf5.read_dataset(path, "/y")
    assert np.array_equal(got[:, 100:300], x[:1, 100:300]) and hdf5.dataset_shape(path, "/y") == (1, n)
    with pytest.raises(ValueError, match="rank mismatch"):
        hdf5.read_dataset_chunk(path, "/x", (0,), (5,))
    with pytest.raises(TypeError, match="type mismatch"):
        hdf5.read_dataset_chunk(path, "/x", (0, 0), (1, 5), dtype=np.uint64)
    with pytest.raises(TypeError, match="type mismatch"):
        hdf5.write_dataset_chunk(path, "/y", (0, 0), np.zeros((1, 5), dtype=np.uint64))
    with pytest.raises(IndexError):
        hdf5.read_dataset_chunk(path, "/x", (0, n - 3), (2, 10))
    with pytest.raises(ValueError, match="Block distribution"):
        hdf5.write_dataset_as_blocks(path, "/z", [np.zeros((1, 5)), np.zeros((1, 3))])


def _write_golden_like(tmp_path, name):
    """a file with the layout of the reference's data/matvec/<name>.h5 (input_for_matvec.py:43-46), from the
    committed golden vectors (x by the reference's recipe, y by the dense oracle)."""
    import yaml

    hdf5 = _hdf5()
    v = golden_vectors()
    h5 = str(tmp_path / f"{name}.h5")
    hdf5.write_datasets(h5, {"/representatives": v[name + "/representatives"], "/x": v[name + "/x"][None, :], "/y": v[name + "/y"][None, :]})
    yml = str(tmp_path / f"{name}.yaml")
    with open(yml, "w", encoding="utf-8") as f:
        yaml.safe_dump(model_config(name), f, allow_unicode=True)
    return yml, h5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["heisenberg_chain_10", "heisenberg_chain_12", "heisenberg_kagome_12_symm", "heisenberg_square_4x4"])
@pytest.mark.parametrize("num_locales", [1, 3])
def test_mirrored_integration_tests(tmp_path, name, num_locales):
    """TestMatrixVectorProduct.chpl / TestStatesEnumeration.chpl on YAML + HDF5 files."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    from distributed_matvec_amd import check

    yml, h5 = _write_golden_like(tmp_path, name)
    out = io.StringIO()
    ok, _ = check.test_matrix_vector_product(yml, h5, num_locales, out=out)
    assert ok and out.getvalue().splitlines()[0] == "true"
    if num_locales == 1:
        ok2, _ = check.test_states_enumeration(yml, h5, out=io.StringIO())
        assert ok2


@pytest.mark.gpu
def test_diagonalize_writes_reference_style_hdf5(tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    hdf5 = _hdf5()
    from distributed_matvec_amd.diagonalize import diagonalize

    out = str(tmp_path / "exact_diagonalization_output.h5")
    r = diagonalize(model_config("heisenberg_chain_10"), num_evals=1, eps=1e-10, output=out)
    assert abs(hdf5.read_dataset(out, "/hamiltonian/eigenvalues")[0] - (-18.061785417968)) < 1e-8
    assert hdf5.read_dataset(out, "/basis/representatives").shape == (126,)
    assert hdf5.dataset_shape(out, "/hamiltonian/eigenvectors") == (1, 126)
    v = hdf5.read_dataset(out, "/hamiltonian/eigenvectors")[0]
    assert np.allclose(v, r.eigenvectors[0].cpu().numpy(), rtol=0, atol=1e-15) and abs(np.linalg.norm(v) - 1.0) < 1e-12
    # three hash partitions: the vector is converted to block order and written as three hyperslabs (Block distribution)
    out3 = str(tmp_path / "three_locales.h5")
    r3 = diagonalize(model_config("heisenberg_chain_10"), num_evals=1, eps=1e-10, output=out3, num_partitions=3)
    v3 = hdf5.read_dataset(out3, "/hamiltonian/eigenvectors")[0]
    assert abs(r3.eigenvalues[0] - r.eigenvalues[0]) < 1e-9 and abs(abs(np.dot(v3, v)) - 1.0) < 1e-8
    assert np.array_equal(hdf5.read_dataset(out3, "/basis/representatives"), hdf5.read_dataset(out, "/basis/representatives"))


def test_append_keeps_the_stored_basis_and_replaces_results(tmp_path):
    """the reference's output file is opened read-write and extended (Diagonalize.chpl:227-256): an existing
    basis/representatives survives, hamiltonian/* is replaced"""
    hdf5 = _hdf5()
    path = str(tmp_path / "out.h5")
    reps = np.arange(5, dtype=np.uint64)
    hdf5.write_datasets(path, {"/basis/representatives": reps, "/hamiltonian/eigenvalues": np.array([1.0])})
    assert hdf5.has_dataset(path, "/basis/representatives") and not hdf5.has_dataset(path, "/hamiltonian/residuals")
    assert not hdf5.has_dataset(str(tmp_path / "missing.h5"), "/basis/representatives")
    hdf5.write_datasets(path, append=True, datasets={"/basis/representatives": np.arange(7, dtype=np.uint64),
                                                     "/hamiltonian/eigenvalues": np.array([-2.0, 3.0]),
                                                     "/hamiltonian/residuals": np.array([1e-9])})
    assert np.array_equal(hdf5.read_dataset(path, "/basis/representatives"), reps)
    assert np.array_equal(hdf5.read_dataset(path, "/hamiltonian/eigenvalues"), [-2.0, 3.0])
    assert np.array_equal(hdf5.read_dataset(path, "/hamiltonian/residuals"), [1e-9])


@pytest.mark.gpu
def test_diagonalize_accepts_its_own_output_file(tmp_path):
    """makeBasisStates (Diagonalize.chpl:227-246): representatives found in the output file are taken over -- after they have
    been checked against what the configured basis enumerates to (seconds on the GPU; the reference skips hours of CPU time
    here and trusts the file)"""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    hdf5 = _hdf5()
    from distributed_matvec_amd.diagonalize import diagonalize

    out = str(tmp_path / "ed.h5")
    r1 = diagonalize(model_config("heisenberg_chain_10"), num_evals=1, eps=1e-10, output=out)
    r2 = diagonalize(model_config("heisenberg_chain_10"), num_evals=1, eps=1e-10, output=out)
    assert abs(r1.eigenvalues[0] - r2.eigenvalues[0]) < 1e-9
    assert hdf5.read_dataset(out, "/basis/representatives").shape == (126,)


@pytest.mark.gpu
def test_diagonalize_refuses_a_stale_output_file(tmp_path):
    """an output file whose /basis/representatives belongs to ANOTHER model or sector is an error, not a silent wrong answer"""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    from distributed_matvec_amd import api
    from distributed_matvec_amd.diagonalize import diagonalize

    out = str(tmp_path / "ed.h5")
    diagonalize(model_config("heisenberg_chain_12"), num_evals=1, eps=1e-8, output=out)  # the full space of 12 sites
    for other in ("heisenberg_chain_10", "heisenberg_chain_16", "heisenberg_kagome_12_symm"):
        with pytest.raises(api.LsAmdError, match="does not belong"):
            diagonalize(model_config(other), num_evals=1, eps=1e-8, output=out)
    # the same lattice in another Hamming-weight sector, through the YAML loader (whose Basis objects carry no Python-side
    # spec): ls_hs_is_representative alone would accept these states
    import copy

    import yaml

    out16 = str(tmp_path / "ed16.h5")
    diagonalize(model_config("heisenberg_chain_16"), num_evals=1, eps=1e-8, output=out16)  # hamming_weight 8
    cfg = copy.deepcopy(model_config("heisenberg_chain_16"))
    cfg["basis"]["hamming_weight"] = 7
    path = str(tmp_path / "w7.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f, allow_unicode=True)
    with pytest.raises(api.LsAmdError, match="does not belong"):
        diagonalize(path, num_evals=1, eps=1e-8, output=out16)
    # a truncated dataset of the right model: every stored state is a representative, but the basis is incomplete
    hdf5 = _hdf5()
    out2 = str(tmp_path / "short.h5")
    reps = hdf5.read_dataset(out, "/basis/representatives")
    hdf5.write_datasets(out2, {"/basis/representatives": reps[:-3]})
    with pytest.raises(api.LsAmdError, match="does not belong"):
        diagonalize(model_config("heisenberg_chain_12"), num_evals=1, eps=1e-8, output=out2)

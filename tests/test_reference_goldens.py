"""The reference's whole `make check` matrix over a directory of goldens.

/root/reference/Makefile:88-125 runs TestStatesEnumeration on 14 files and TestMatrixVectorProduct on 13 files of data/matvec/ --
artefacts its Makefile downloads (:128-146) and this container does not have.  The harness below walks such a directory for BOTH
the C oracle (CPU tests) and the HIP path (gpu tests): point $LS_REFERENCE_DATA at the unpacked download (the directory that
holds matvec/ and construction/; the YAML inputs are taken from next to them, else from tests/golden/models.json) and the two
`*_reference_goldens` tests stop skipping -- that is the day parity is pinned by reference artefacts.  Until then the same walk
runs on a SYNTHETIC directory in the reference's HDF5 layout (tests/golden/make_vectors.py::write_reference_layout: y from the
dense Kronecker oracle), which proves the harness end to end: file discovery, dataset layout (batch, N), numLocales in {1, 4},
tolerances of test/TestMatrixVectorProduct.chpl:15-20, exact representatives, and that a wrong golden FAILS.
"""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

DATA = os.environ.get("LS_REFERENCE_DATA")
NO_DATA = ("the reference's golden HDF5 files are downloaded artefacts (/root/reference/Makefile:128-146) and are absent: set "
           "LS_REFERENCE_DATA to the directory holding matvec/ and construction/")


def _hdf5():
    from distributed_matvec_amd import hdf5

    try:
        hdf5.lib()
    except hdf5.Hdf5Unavailable:
        pytest.skip("libhdf5 not available")
    return hdf5


class OracleEngine:
    """the CPU restatement behind the same walk (test infrastructure: lives in tests/, never in the package)"""

    name = "oracle"

    @staticmethod
    def _model(cfg):
        from oracle import c_oracle as CO
        from oracle import model as M

        kind, what = cfg
        if kind == "yaml":
            import yaml

            with open(what) as f:
                what = yaml.safe_load(f)
        return CO.COracle(M.model_from_config(what))

    def states(self, cfg):
        return self._model(cfg).enumerate()

    def matvec(self, cfg, x_block, numLocales):
        from oracle import c_oracle as CO

        o = self._model(cfg)
        reps = o.enumerate()
        if numLocales == 1:
            return o.local_matvec(reps, np.ascontiguousarray(x_block))
        # the reference's multi-locale product (DMV:1072-1093) restated: hashed blocks in, every locale expands its rows and the
        # contributions go to their owners, hashed blocks out
        keys = CO.locale_idx_of(reps, numLocales)
        reps_parts = CO.block_to_hashed(reps, keys, numLocales)
        x_parts = CO.block_to_hashed(np.ascontiguousarray(x_block), keys, numLocales)
        return CO.hashed_to_block(o.matvec_partitioned(reps_parts, x_parts), keys)


def _synthetic(tmp_path):
    _hdf5()
    import make_vectors

    d = str(tmp_path / "data")
    names = make_vectors.write_reference_layout(d)
    assert len(names) >= 10
    return d, names


def _assert_all_ok(results, engine):
    bad = [r for r in results if not r["ok"]]
    assert results and not bad, (engine, bad)


# ---- CPU: the oracle --------------------------------------------------------------------------------------------------------------

def test_oracle_walks_a_synthetic_golden_directory(tmp_path):
    from distributed_matvec_amd import check

    d, names = _synthetic(tmp_path)
    out = io.StringIO()
    res = check.walk_goldens(d, OracleEngine(), out=out)
    _assert_all_ok(res, "oracle")
    # every file: representatives twice (matvec/ and construction/), matvec with 1 and 4 locales
    assert len(res) == len(names) * 4, out.getvalue()
    assert {r["numLocales"] for r in res if r["kind"] == "matvec"} == {1, 4}
    # what `make check` names and this directory holds is found under the same file names
    assert set(names) <= set(check.MAKE_CHECK_STATES) | set(check.MAKE_CHECK_MATVEC)


def test_walk_reports_a_wrong_golden(tmp_path):
    """one element of one /y off by 1e-9 relative, one representative off by one bit: exactly those two checks fail"""
    from distributed_matvec_amd import check, hdf5

    d, _names = _synthetic(tmp_path)
    p = os.path.join(d, "matvec", "heisenberg_chain_10.h5")
    ds = {k: hdf5.read_dataset(p, k) for k in ("/representatives", "/x", "/y")}
    ds["/y"] = ds["/y"].copy()
    ds["/y"][0, 17] *= 1.0 + 1e-9
    hdf5.write_datasets(p, ds)
    p2 = os.path.join(d, "construction", "heisenberg_kagome_12.h5")
    reps = hdf5.read_dataset(p2, "/representatives").copy()
    reps[5] ^= 1
    hdf5.write_datasets(p2, {"/representatives": reps})
    res = check.walk_goldens(d, OracleEngine(), out=io.StringIO(), only={"heisenberg_chain_10", "heisenberg_kagome_12"})
    bad = {(r["kind"], r["name"], r["numLocales"]) for r in res if not r["ok"]}
    assert bad == {("matvec", "heisenberg_chain_10", 1), ("matvec", "heisenberg_chain_10", 4), ("states", "heisenberg_kagome_12", 1)}, bad


def test_golden_files_without_a_model_are_failures_not_skips(tmp_path):
    from distributed_matvec_amd import check, hdf5

    _hdf5()
    os.makedirs(tmp_path / "matvec")
    hdf5.write_datasets(str(tmp_path / "matvec" / "no_such_model.h5"), {"/representatives": np.arange(3, dtype=np.uint64)})
    res = check.walk_goldens(str(tmp_path), OracleEngine(), out=io.StringIO())
    assert len(res) == 1 and not res[0]["ok"] and "no model" in res[0]["detail"]


def test_make_check_matrix_is_the_reference_makefiles(have_reference):
    """the file lists of check.py are the Makefile's (read here, where /root/reference exists)"""
    if not have_reference:
        pytest.skip("/root/reference is absent (GPU box)")
    import re

    from distributed_matvec_amd import check

    mk = open("/root/reference/Makefile").read()
    states = re.findall(r"--kRepresentatives data/matvec/(\w+)\.h5", mk.split("check-states-enumeration:")[1].split(".PHONY")[0])
    matvec = re.findall(r"--kVectors data/matvec/(\w+)\.h5", mk.split("check-matrix-vector-product:")[1].split("TEST_DATA_URL")[0])
    assert states == check.MAKE_CHECK_STATES
    assert matvec == check.MAKE_CHECK_MATVEC
    for n in set(states) | set(matvec):  # ... and every one of them has its model in tests/golden/models.json (what the GPU box reads)
        assert check.find_config(n, "/nonexistent") is not None, n


@pytest.mark.skipif(not (DATA and os.path.isdir(DATA)), reason=NO_DATA)
def test_oracle_against_reference_goldens():
    from distributed_matvec_amd import check

    _hdf5()
    res = check.walk_goldens(DATA, OracleEngine(), only=set(check.MAKE_CHECK_STATES[:7] + check.MAKE_CHECK_STATES[8:13]) | {"issue_01"})
    _assert_all_ok(res, "oracle vs reference artefacts")


# ---- GPU: the HIP path --------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_hip_path_walks_a_synthetic_golden_directory(tmp_path):
    from distributed_matvec_amd import check

    d, names = _synthetic(tmp_path)
    out = io.StringIO()
    res = check.walk_goldens(d, check.HipEngine(), out=out)
    _assert_all_ok(res, "hip")
    assert len(res) == len(names) * 4, out.getvalue()


@pytest.mark.gpu
def test_check_all_command_line(tmp_path):
    """`python -m distributed_matvec_amd.check all --data DIR`: rc 0 on a clean directory, 1 after one /y is damaged, 2 without one"""
    from distributed_matvec_amd import hdf5

    d, _names = _synthetic(tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("LS_REFERENCE_DATA", None)
    run = lambda *a: subprocess.run([sys.executable, "-m", "distributed_matvec_amd.check", "all", *a], capture_output=True, text=True,  # noqa: E731
                                    timeout=600, env=env, cwd=ROOT)
    p = run("--data", d)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "checks passed" in p.stdout and "FAIL" not in p.stdout
    f = os.path.join(d, "matvec", "heisenberg_chain_8.h5")
    ds = {k: hdf5.read_dataset(f, k) for k in ("/representatives", "/x", "/y")}
    ds["/y"] = -ds["/y"]
    hdf5.write_datasets(f, ds)
    p = run("--data", d)
    assert p.returncode == 1 and "FAIL matvec heisenberg_chain_8" in p.stdout, p.stdout[-2000:]
    assert run().returncode == 2


@pytest.mark.gpu
@pytest.mark.skipif(not (DATA and os.path.isdir(DATA)), reason=NO_DATA)
def test_hip_path_against_reference_goldens():
    from distributed_matvec_amd import check

    _hdf5()
    res = check.walk_goldens(DATA, check.HipEngine())
    _assert_all_ok(res, "hip vs reference artefacts")

"""Generates tests/golden/vectors.npz: the reference goldens' INPUT vectors, regenerated from the recipe
of /root/reference/input_for_matvec.py:8,31,49-75 (np.random.seed(42), then sequentially
x = rand(N, 1) - 0.5 per model), and y = H x from the dense Kronecker/projector oracle
(oracle/model.py: independent of every term table and of the HIP path) for the models small enough
for a dense construction.  The reference's own /y (data/matvec/*.h5) are downloaded artefacts that
are absent offline (/root/reference/Makefile:128-146); if the recipe assumption (N per file, draw
order) is right these y agree with them to ~1e-12.

    python tests/golden/make_vectors.py     # needs only this repo (models.json)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import c_oracle as CO  # noqa: E402
from oracle import model as M  # noqa: E402

DENSE_MAX_SITES = 16
STORE_MAX_N = 13000


def main():
    d = json.load(open(os.path.join(HERE, "models.json"), encoding="utf-8"))
    order = d["fixtures"]["input_for_matvec_order"]
    rs = np.random.RandomState(d["fixtures"]["input_for_matvec_seed"])
    out = {}
    dims = {}
    for name in order:
        entry = d["models"][name]
        cfg_old = entry.get("old_config", entry["config"])
        m_old = M.model_from_config(cfg_old)
        reps_old = CO.COracle(m_old).enumerate()
        n = len(reps_old)
        dims[name] = n
        x = rs.rand(n, 1)[:, 0] - 0.5
        cfg = entry["config"]
        m = M.model_from_config(cfg)
        reps = CO.COracle(m).enumerate()
        assert len(reps) == n and np.array_equal(reps, reps_old), name
        if n <= STORE_MAX_N:
            out[name + "/x"] = x
            out[name + "/representatives"] = reps
            if m.number_sites <= DENSE_MAX_SITES:
                r2, H = M.dense_sector_matrix(cfg)
                assert np.array_equal(r2, reps)
                assert np.abs(H.imag).max() < 1e-12
                out[name + "/y"] = H.real @ x
        print(name, n, flush=True)
    out["dims_names"] = np.array(list(dims.keys()))
    out["dims_values"] = np.array(list(dims.values()), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **out)
    print("wrote vectors.npz")


def write_reference_layout(directory, names=None, with_construction=True):
    """A directory in the layout of the reference's downloaded goldens (/root/reference/Makefile:128-146 unpacks data/matvec,
    data/construction; /root/reference/input_for_matvec.py:41-46 writes /representatives, /x = x.T and /y = y.T, i.e. shape
    (batch, N)), filled from vectors.npz: `<dir>/matvec/<name>.h5` for every model that has a dense-oracle y there and
    `<dir>/construction/<name>.h5` (/representatives only).  SYNTHETIC: these y are this repository's, not the reference's --
    the directory exists so that the golden harness (distributed_matvec_amd.check.walk_goldens, tests/test_reference_goldens.py)
    can be exercised end to end without the artefacts.  Returns the model names written."""
    from distributed_matvec_amd import hdf5

    v = np.load(os.path.join(HERE, "vectors.npz"))
    have = sorted({k.split("/")[0] for k in v.keys() if k.endswith("/y")})
    names = [n for n in (names or have) if n in have]
    os.makedirs(os.path.join(directory, "matvec"), exist_ok=True)
    if with_construction:
        os.makedirs(os.path.join(directory, "construction"), exist_ok=True)
    for n in names:
        reps = np.ascontiguousarray(v[n + "/representatives"], dtype=np.uint64)
        x = np.ascontiguousarray(v[n + "/x"], dtype=np.float64)[None, :]
        y = np.ascontiguousarray(v[n + "/y"], dtype=np.float64)[None, :]
        hdf5.write_datasets(os.path.join(directory, "matvec", n + ".h5"), {"/representatives": reps, "/x": x, "/y": y})
        if with_construction:
            hdf5.write_datasets(os.path.join(directory, "construction", n + ".h5"), {"/representatives": reps})
    return names


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--layout":
        print("wrote", write_reference_layout(sys.argv[2]))
    else:
        main()

"""The reference's two integration tests, runnable against this library on the same YAML/HDF5 inputs:

    python -m distributed_matvec_amd.check matvec  --kHamiltonian data/heisenberg_chain_10.yaml \\
            --kVectors data/matvec/heisenberg_chain_10.h5 [--numLocales 4]
    python -m distributed_matvec_amd.check states  --kHamiltonian data/heisenberg_kagome_12.yaml \\
            --kRepresentatives data/heisenberg_kagome_12.h5

    python -m distributed_matvec_amd.check all --data data     # every golden under data/{matvec,construction} x numLocales 1, 4

mirrors /root/reference/test/TestMatrixVectorProduct.chpl:25-60 and
/root/reference/test/TestStatesEnumeration.chpl:12-45 (same flag names as the Chapel `config const`s,
same tolerance formula, same printed output: True/False then elapsed seconds).
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def approx_equal(a, b, atol=1e-14, rtol=1e-12):
    """TestMatrixVectorProduct.chpl:15-20"""
    return np.abs(a - b) <= np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b)))


def test_matrix_vector_product(kHamiltonian: str, kVectors: str, numLocales: int = 1, kAbsTol=1e-14, kRelTol=1e-12,
                               out=sys.stdout):
    import torch

    from . import api, hdf5

    _, matrix = api.loadConfigFromYaml(kHamiltonian, hamiltonian=True)
    basisStates, masks = api.enumerateStates(matrix.basis, numLocales)
    x_block = hdf5.read_dataset(kVectors, "/x")[0, :]
    x = api.arrFromBlockToHashed(torch.from_numpy(np.ascontiguousarray(x_block)).cuda(), masks, numLocales)
    z = [torch.zeros_like(v) for v in x]  # similar(x), Vector.chpl:303-313
    torch.cuda.synchronize()
    t = time.perf_counter()
    api.matrixVectorProduct(matrix, x, z, basisStates)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t
    yBlock = hdf5.read_dataset(kVectors, "/y")[0, :]
    zBlock = api.arrFromHashedToBlock(z, masks).cpu().numpy()
    close = approx_equal(zBlock, yBlock, kAbsTol, kRelTol)
    ok = bool(close.all())
    print(str(ok).lower(), file=out)
    if not ok:
        for i in np.flatnonzero(~close)[:10]:
            print(f"at {i}: {zBlock[i]} (computed) != {yBlock[i]} (expected)", file=out)
    print(elapsed, file=out)
    return ok, elapsed


def test_states_enumeration(kHamiltonian: str, kRepresentatives: str, out=sys.stdout):
    from . import api, hdf5

    reference = hdf5.read_dataset(kRepresentatives, "/representatives")
    basis = api.loadConfigFromYaml(kHamiltonian)
    t = time.perf_counter()
    basis.build()
    elapsed = time.perf_counter() - t
    predicted = basis.representatives()
    print(predicted.size, file=out)
    same = predicted.shape == reference.shape and bool((predicted == reference).all())
    if not same:
        for i in np.flatnonzero(predicted[: reference.size] != reference[: predicted.size])[:10]:
            print(f"at index {i}: {reference[i]} != {predicted[i]}", file=out)
    print(elapsed, file=out)
    return same, elapsed


# ---- the reference's whole check matrix over a directory of goldens ------------------------------------------------------------
# `make check` (/root/reference/Makefile:88-125) = TestStatesEnumeration on 14 files + TestMatrixVectorProduct on 13 files of
# data/matvec/ (downloaded artefacts, Makefile:128-146: data/construction, data/matvec, data/large-scale); CI runs the matvec
# test with numLocales 1 and 4.  `walk_goldens` runs that matrix -- every `<DIR>/matvec/*.h5` x numLocales, every
# `<DIR>/construction/*.h5` (and `<DIR>/large-scale/{matvec,construction}/*.h5` with large=True) -- through an ENGINE:
#   HipEngine (here)       the product: HIP enumeration, hashed layout, matrixVectorProduct
#   tests/...OracleEngine  the CPU oracle (test infrastructure; never imported from this package)
# so that the day the artefacts exist, `LS_REFERENCE_DATA=<DIR> pytest tests/test_reference_goldens.py` pins both.

MAKE_CHECK_STATES = ["heisenberg_chain_4", "heisenberg_chain_6", "heisenberg_chain_8", "heisenberg_chain_10", "heisenberg_chain_12",
                     "heisenberg_chain_16", "heisenberg_chain_20", "heisenberg_chain_24", "heisenberg_chain_24_symm",
                     "heisenberg_kagome_12", "heisenberg_kagome_12_symm", "heisenberg_kagome_16", "heisenberg_square_4x4",
                     "heisenberg_square_5x5"]  # Makefile:88-103
MAKE_CHECK_MATVEC = ["issue_01"] + MAKE_CHECK_STATES[:12]  # Makefile:111-125
MAKE_CHECK_LOCALES = (1, 4)


def _models_json():
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models.json")
    try:
        with open(path, encoding="utf-8") as f:
            return json.load(f)["models"]
    except (OSError, ValueError, KeyError):
        return {}


def find_config(name: str, data_dir: str, yaml_dir: str | None = None):
    """the model of a golden file `<name>.h5`: `<yaml_dir>/<name>.yaml`, else `<data_dir>/<name>.yaml` (the reference keeps its
    inputs next to data/matvec), else the parsed copy of the reference's input in tests/golden/models.json.  Returns
    ("yaml", path) | ("dict", config) | None."""
    import os

    for d in (yaml_dir, data_dir):
        if d and os.path.isfile(os.path.join(d, name + ".yaml")):
            return "yaml", os.path.join(d, name + ".yaml")
    m = _models_json().get(name)
    return ("dict", m["config"]) if m else None


class HipEngine:
    """the product path behind the walk: loadConfigFromYaml / enumerateStates / matrixVectorProduct on the current device"""

    name = "hip"

    def _load(self, cfg, hamiltonian):
        from . import api

        kind, what = cfg
        if kind == "yaml":
            return api.loadConfigFromYaml(what, hamiltonian=hamiltonian)
        return api.loadConfigFromDict(what, hamiltonian=hamiltonian)

    def states(self, cfg):
        basis = self._load(cfg, False)
        basis.build()
        return np.array(basis.representatives(), dtype=np.uint64, copy=True)  # (a view into the basis' own memory otherwise)

    def matvec(self, cfg, x_block, numLocales):
        import torch

        from . import api

        _, matrix = self._load(cfg, True)
        basisStates, masks = api.enumerateStates(matrix.basis, numLocales)
        x = api.arrFromBlockToHashed(torch.from_numpy(np.ascontiguousarray(x_block)).cuda(), masks, numLocales)
        z = [torch.zeros_like(v) for v in x]
        api.matrixVectorProduct(matrix, x, z, basisStates)
        return api.arrFromHashedToBlock(z, masks).cpu().numpy()


def golden_files(data_dir: str, large: bool = False):
    """[(kind, name, path)]: kind 'matvec' for files with /x and /y, 'states' for every file with /representatives"""
    import glob
    import os

    from . import hdf5

    out = []
    subs = ["matvec", "construction"] + (["large-scale/matvec", "large-scale/construction"] if large else [])
    for sub in subs:
        for path in sorted(glob.glob(os.path.join(data_dir, sub, "*.h5"))):
            name = os.path.splitext(os.path.basename(path))[0]
            if hdf5.has_dataset(path, "/representatives"):
                out.append(("states", name, path))
            if hdf5.has_dataset(path, "/x") and hdf5.has_dataset(path, "/y"):
                out.append(("matvec", name, path))
    return out


def walk_goldens(data_dir: str, engine, locales=MAKE_CHECK_LOCALES, yaml_dir: str | None = None, large: bool = False,
                 only=None, kAbsTol=1e-14, kRelTol=1e-12, out=sys.stdout):
    """every golden of `data_dir` through `engine`; returns a list of result dicts {kind, name, numLocales, ok, detail, seconds}.
    Tolerances and comparisons are the reference tests' (TestMatrixVectorProduct.chpl:15-20, TestStatesEnumeration.chpl:34)."""
    from . import hdf5

    results = []
    for kind, name, path in golden_files(data_dir, large):
        if only and name not in only:
            continue
        cfg = find_config(name, data_dir, yaml_dir)
        if cfg is None:
            results.append({"kind": kind, "name": name, "numLocales": None, "ok": False, "detail": "no model (yaml) for this file", "seconds": 0.0})
            continue
        if kind == "states":
            reference = hdf5.read_dataset(path, "/representatives").reshape(-1).view(np.uint64)
            t = time.perf_counter()
            predicted = np.asarray(engine.states(cfg)).reshape(-1).view(np.uint64)
            dt = time.perf_counter() - t
            ok = predicted.shape == reference.shape and bool((predicted == reference).all())
            detail = f"{predicted.size} representatives" if ok else (
                f"{predicted.size} representatives, reference {reference.size}; first difference at index "
                f"{int(np.flatnonzero(predicted[:reference.size] != reference[:predicted.size])[:1].tolist()[0]) if min(predicted.size, reference.size) and (predicted[:reference.size] != reference[:predicted.size]).any() else min(predicted.size, reference.size)}")
            results.append({"kind": kind, "name": name, "numLocales": 1, "ok": ok, "detail": detail, "seconds": dt})
        else:
            x = hdf5.read_dataset(path, "/x")
            y = hdf5.read_dataset(path, "/y")
            for nl in locales:
                worst, ok, dt = 0.0, True, 0.0
                for k in range(x.shape[0]):  # rows of /x are vectors (input_for_matvec.py:45-46 stores x.T); the reference reads row 0
                    t = time.perf_counter()
                    z = np.asarray(engine.matvec(cfg, x[k, :], nl))
                    dt += time.perf_counter() - t
                    close = approx_equal(z, y[k, :], kAbsTol, kRelTol)
                    ok = ok and bool(close.all())
                    scale = max(float(np.abs(y[k, :]).max()), 1e-300) if y.shape[1] else 1.0
                    worst = max(worst, float(np.abs(z - y[k, :]).max()) / scale if y.shape[1] else 0.0)
                    if k == 0 and not nl == 1 and x.shape[0] > 1:
                        break  # further vectors of a batch only with one locale (the reference reads row 0 alone)
                results.append({"kind": kind, "name": name, "numLocales": nl, "ok": ok,
                                "detail": f"max |z - y| / max |y| = {worst:.3e}, {int(y.shape[1])} rows", "seconds": dt})
        r = results[-1]
    for r in results:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['kind']:<6} {r['name']:<28} nl={r['numLocales']} {r['detail']} ({r['seconds']:.3f} s)", file=out)
    return results


def check_all(data_dir: str, yaml_dir: str | None = None, large: bool = False, locales=MAKE_CHECK_LOCALES, out=sys.stdout):
    results = walk_goldens(data_dir, HipEngine(), locales=locales, yaml_dir=yaml_dir, large=large, out=out)
    have = {(r["kind"], r["name"]) for r in results}
    missing = [("states", n) for n in MAKE_CHECK_STATES if ("states", n) not in have] + \
              [("matvec", n) for n in MAKE_CHECK_MATVEC if ("matvec", n) not in have]
    for kind, n in missing:
        print(f"note: `make check` also runs {kind} on {n}: no such golden under {data_dir}", file=out)
    bad = [r for r in results if not r["ok"]]
    print(f"{len(results) - len(bad)} of {len(results)} checks passed ({len(missing)} of the reference's `make check` files absent)", file=out)
    return bool(results) and not bad


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("matvec")
    m.add_argument("--kHamiltonian", default="data/heisenberg_chain_10.yaml")
    m.add_argument("--kVectors", default="data/matvec/heisenberg_chain_10.h5")
    m.add_argument("--numLocales", "-nl", type=int, default=1)
    m.add_argument("--kAbsTol", type=float, default=1e-14)
    m.add_argument("--kRelTol", type=float, default=1e-12)
    s = sub.add_parser("states")
    s.add_argument("--kHamiltonian", default="data/heisenberg_kagome_12.yaml")
    s.add_argument("--kRepresentatives", default="data/heisenberg_kagome_12.h5")
    al = sub.add_parser("all", help="the reference's whole `make check` matrix over a directory of goldens")
    al.add_argument("--data", default=None, help="directory holding matvec/ and construction/ (default: $LS_REFERENCE_DATA)")
    al.add_argument("--yaml-dir", default=None, help="where <name>.yaml lives (default: --data, then tests/golden/models.json)")
    al.add_argument("--large-scale", action="store_true", help="also large-scale/{matvec,construction}")
    al.add_argument("--numLocales", "-nl", type=int, nargs="+", default=list(MAKE_CHECK_LOCALES))
    a = ap.parse_args(argv)
    if a.cmd == "all":
        import os

        data = a.data or os.environ.get("LS_REFERENCE_DATA")
        if not data or not os.path.isdir(data):
            print("check all: no golden directory (--data DIR or $LS_REFERENCE_DATA; the reference downloads it: Makefile:128-146)", file=sys.stderr)
            return 2
        return 0 if check_all(data, a.yaml_dir, a.large_scale, tuple(a.numLocales)) else 1
    if a.cmd == "matvec":
        ok, _ = test_matrix_vector_product(a.kHamiltonian, a.kVectors, a.numLocales, a.kAbsTol, a.kRelTol)
    else:
        ok, _ = test_states_enumeration(a.kHamiltonian, a.kRepresentatives)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

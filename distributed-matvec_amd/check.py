"""The reference's two integration tests, runnable against this library on the same YAML/HDF5 inputs:

    python -m distributed_matvec_amd.check matvec  --kHamiltonian data/heisenberg_chain_10.yaml \\
            --kVectors data/matvec/heisenberg_chain_10.h5 [--numLocales 4]
    python -m distributed_matvec_amd.check states  --kHamiltonian data/heisenberg_kagome_12.yaml \\
            --kRepresentatives data/heisenberg_kagome_12.h5

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
    a = ap.parse_args(argv)
    if a.cmd == "matvec":
        ok, _ = test_matrix_vector_product(a.kHamiltonian, a.kVectors, a.numLocales, a.kAbsTol, a.kRelTol)
    else:
        ok, _ = test_states_enumeration(a.kHamiltonian, a.kRepresentatives)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

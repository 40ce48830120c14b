"""A foreign ls_hs_operator -- the reference's struct prefix and nothing of ours -- through the plug-in entry
ls_chpl_matrix_vector_product (/root/reference/src/DistributedMatrixVector.chpl:1095-1110) and through a plan."""
import ctypes as C

import numpy as np
import pytest

from helpers import model_config, oracle_for, oracle_reps
from test_host_tables import _foreign_copy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["heisenberg_chain_10", "heisenberg_chain_16", "heisenberg_kagome_12_symm", "heisenberg_chain_24_symm"])
def test_foreign_operator_matvec(name):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib

    lib = _lib.load()
    fb, fo, (basis, h) = _foreign_copy(name)
    spec = basis.spec
    ng = len(spec.permutations)
    perms = (C.c_int * max(1, ng * spec.number_sites))(*[v for p in spec.permutations for v in p])
    sectors = (C.c_int * max(1, ng))(*spec.sectors)
    bp = C.cast(C.pointer(fb), C.POINTER(_lib.LsHsBasis))
    op = C.cast(C.pointer(fo), C.POINTER(_lib.LsHsOperator))
    assert lib.ls_amd_adopt_basis(bp, ng, perms, sectors) == 0, lib.ls_amd_last_error()
    assert lib.ls_amd_adopt_operator(op) == 0, lib.ls_amd_last_error()
    try:
        reps = oracle_reps(name)
        # the foreign library built the basis: its representatives sit in the prefix (host memory, borrowed)
        fb.representatives = _lib.ChplExternalArray(reps.ctypes.data, reps.size, None)
        n = len(reps)
        x = np.random.RandomState(5).rand(n) - 0.5
        y = np.full(n, 7.0)
        lib.ls_chpl_matrix_vector_product(op, 1, x.ctypes.data_as(_lib.c_f64p), y.ctypes.data_as(_lib.c_f64p))
        _lib.raise_pending_halt()
        want = oracle_for(name).local_matvec(reps, x)
        assert np.abs(y - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        # and through the device-level API with hash partitions
        wrapped = D.Operator(op, owning=False)
        P = 3
        from oracle import c_oracle as CO

        keys = CO.locale_idx_of(reps, P)
        parts = CO.block_to_hashed(reps, keys, P)
        xs = CO.block_to_hashed(x, keys, P)
        rt = [torch.from_numpy(p.view(np.int64).copy()).cuda() for p in parts]
        xt = [torch.from_numpy(v.copy()).cuda() for v in xs]
        yt = [torch.zeros_like(v) for v in xt]
        pl = D.MatvecPlan(wrapped, rt, torch.float64)
        pl.matvec(xt, yt)
        got = CO.hashed_to_block([v.cpu().numpy() for v in yt], keys)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        pl.destroy()
        assert bytes(fb.other_stuff) == b"\xab" * 64 and bytes(fo.other_stuff) == b"\xcd" * 64
    finally:
        fb.representatives = _lib.ChplExternalArray(None, 0, None)
        lib.ls_amd_release(C.cast(op, C.c_void_p))
        lib.ls_amd_release(C.cast(bp, C.c_void_p))

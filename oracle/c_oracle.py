"""ctypes binding of oracle/ls_oracle.c (TEST INFRASTRUCTURE ONLY; see oracle/model.py header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import model as M

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


def build(force: bool = False):
    so = os.path.join(_HERE, "liblsoracle.so")
    src = os.path.join(_HERE, "ls_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # the .so travels to the GPU box with the repo snapshot (built -march=x86-64-v3 for that reason)
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblsoracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        try:
            _LIB = C.CDLL(so)
        except OSError:
            so = build(force=True)
            _LIB = C.CDLL(so)
        L = _LIB
        L.lso_model_create.restype = C.c_void_p
        L.lso_hash64_01.restype = C.c_uint64
        L.lso_hash64_01.argtypes = [C.c_uint64]
        L.lso_fixed_hamming_state_to_index.restype = C.c_int64
        L.lso_fixed_hamming_state_to_index.argtypes = [C.c_uint64]
        L.lso_fixed_hamming_index_to_state.restype = C.c_uint64
        L.lso_fixed_hamming_index_to_state.argtypes = [C.c_int64, C.c_int]
        L.lso_enumerate.restype = C.c_int64
        L.lso_enumerate_parallel.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class COracle:
    """Owns an lso_model built from an oracle.model.Model."""

    def __init__(self, model: M.Model):
        self.model = model
        L = lib()
        g = model.group
        perms = np.ascontiguousarray(g.perms, dtype=np.int32)
        chars = np.ascontiguousarray(g.chars, dtype=np.complex128)

        def pack(t: M.Terms):
            order = np.argsort(t.x, kind="stable")
            return (
                np.ascontiguousarray(t.v[order], dtype=np.complex128),
                np.ascontiguousarray(t.m[order], dtype=np.uint64),
                np.ascontiguousarray(t.r[order], dtype=np.uint64),
                np.ascontiguousarray(t.x[order], dtype=np.uint64),
                np.ascontiguousarray(t.s[order], dtype=np.uint64),
            )

        d = pack(model.diag)
        o = pack(model.offdiag)
        self._keep = (perms, chars, d, o)
        self.h = C.c_void_p(
            L.lso_model_create(
                C.c_int(model.number_sites), C.c_int(model.hamming_weight), C.c_int(model.spin_inversion),
                C.c_int(perms.shape[0]), _p(perms, _i32p), _p(chars, _f64p),
                C.c_int(len(d[0])), _p(d[0], _f64p), _p(d[1], _u64p), _p(d[2], _u64p), _p(d[3], _u64p), _p(d[4], _u64p),
                C.c_int(len(o[0])), _p(o[0], _f64p), _p(o[1], _u64p), _p(o[2], _u64p), _p(o[3], _u64p), _p(o[4], _u64p),
            )
        )

    def __del__(self):
        try:
            lib().lso_model_destroy(self.h)
        except Exception:
            pass

    # -- restated externs ----------------------------------------------------------------
    def apply_diag(self, alphas, xs=None):
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        ys = np.empty(len(alphas), dtype=np.float64)
        xp = _p(np.ascontiguousarray(xs, dtype=np.float64), _f64p) if xs is not None else None
        lib().lso_apply_diag_x1(self.h, C.c_int64(len(alphas)), _p(alphas, _u64p), _p(ys, _f64p), xp)
        return ys

    def apply_off_diag(self, alphas, xs=None):
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        n = len(alphas)
        T = max(lib().lso_max_number_off_diag(self.h), 1)
        betas = np.empty(n * T, dtype=np.uint64)
        cs = np.empty(n * T, dtype=np.complex128)
        offs = np.empty(n + 1, dtype=np.int64)
        if xs is not None and np.iscomplexobj(xs):
            xs = np.ascontiguousarray(xs, dtype=np.complex128)
            lib().lso_apply_off_diag_x1_c128(self.h, C.c_int64(n), _p(alphas, _u64p), _p(betas, _u64p),
                                             _p(cs, _f64p), _p(offs, _i64p), _p(xs, _f64p))
        else:
            xp = _p(np.ascontiguousarray(xs, dtype=np.float64), _f64p) if xs is not None else None
            lib().lso_apply_off_diag_x1(self.h, C.c_int64(n), _p(alphas, _u64p), _p(betas, _u64p),
                                        _p(cs, _f64p), _p(offs, _i64p), xp)
        tot = offs[n]
        return betas[:tot].copy(), cs[:tot].copy(), offs

    def state_info(self, alphas):
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        n = len(alphas)
        betas = np.empty(n, dtype=np.uint64)
        chars = np.empty(n, dtype=np.complex128)
        norms = np.empty(n, dtype=np.float64)
        lib().lso_state_info(self.h, C.c_int64(n), _p(alphas, _u64p), _p(betas, _u64p), _p(chars, _f64p), _p(norms, _f64p))
        return betas, chars, norms

    def is_representative(self, alphas):
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        n = len(alphas)
        flags = np.empty(n, dtype=np.uint8)
        norms = np.empty(n, dtype=np.float64)
        lib().lso_is_representative(self.h, C.c_int64(n), _p(alphas, _u64p), _p(flags, _u8p), _p(norms, _f64p))
        return flags, norms

    def enumerate(self, num_threads: int = 0):
        if num_threads <= 0:
            num_threads = lib().lso_num_threads()
        n = lib().lso_enumerate_parallel(self.h, None, C.c_int64(0), C.c_int(num_threads))
        out = np.empty(n, dtype=np.uint64)
        n2 = lib().lso_enumerate_parallel(self.h, _p(out, _u64p), C.c_int64(n), C.c_int(num_threads))
        assert n2 == n
        return out

    # -- matvecs --------------------------------------------------------------------------
    @staticmethod
    def prepare_index(reps):
        """build the look-up table of `reps` (a C-contiguous uint64 array that is then passed to local_matvec AS IS) ahead of a
        timed call"""
        assert reps.dtype == np.uint64 and reps.flags.c_contiguous
        lib().lso_prepare_index(_p(reps, _u64p), C.c_int64(len(reps)))

    def local_matvec(self, reps, x, y=None, num_threads: int = 0):
        reps = np.ascontiguousarray(reps, dtype=np.uint64)
        n = len(reps)
        if np.iscomplexobj(x):
            x = np.ascontiguousarray(x, dtype=np.complex128)
            y = np.zeros(n, dtype=np.complex128) if y is None else y
            rc = lib().lso_local_matvec_c128(self.h, C.c_int64(n), _p(reps, _u64p), _p(x, _f64p), _p(y, _f64p), C.c_int(num_threads))
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            y = np.zeros(n, dtype=np.float64) if y is None else y
            rc = lib().lso_local_matvec_f64(self.h, C.c_int64(n), _p(reps, _u64p), _p(x, _f64p), _p(y, _f64p), C.c_int(num_threads))
        if rc != 0:
            raise RuntimeError("oracle matvec: invalid index (operator does not respect basis symmetries)")
        return y

    def matvec_partitioned(self, reps_parts, x_parts, y_parts=None):
        P = len(reps_parts)
        cplx = np.iscomplexobj(x_parts[0])
        dt = np.complex128 if cplx else np.float64
        reps_parts = [np.ascontiguousarray(r, dtype=np.uint64) for r in reps_parts]
        x_parts = [np.ascontiguousarray(v, dtype=dt) for v in x_parts]
        if y_parts is None:
            y_parts = [np.zeros(len(r), dtype=dt) for r in reps_parts]
        counts = np.array([len(r) for r in reps_parts], dtype=np.int64)
        rp = (C.c_void_p * P)(*[r.ctypes.data for r in reps_parts])
        xp = (C.c_void_p * P)(*[v.ctypes.data for v in x_parts])
        yp = (C.c_void_p * P)(*[v.ctypes.data for v in y_parts])
        fn = lib().lso_matvec_partitioned_c128 if cplx else lib().lso_matvec_partitioned_f64
        rc = fn(self.h, C.c_int(P), _p(counts, _i64p), rp, xp, yp)
        if rc != 0:
            raise RuntimeError("oracle partitioned matvec: invalid index")
        return y_parts


def fixed_hamming_ranks(states):
    """ls_hs_fixed_hamming_state_to_index for an array of states"""
    states = np.ascontiguousarray(states, dtype=np.uint64)
    out = np.empty(len(states), dtype=np.int64)
    lib().lso_fixed_hamming_state_to_index_batch(C.c_int64(len(states)), _p(states, _u64p), _p(out, _i64p))
    return out


def state_index(reps, spins):
    reps = np.ascontiguousarray(reps, dtype=np.uint64)
    spins = np.ascontiguousarray(spins, dtype=np.uint64)
    out = np.empty(len(spins), dtype=np.int64)
    lib().lso_state_index(_p(reps, _u64p), C.c_int64(len(reps)), C.c_int64(len(spins)), _p(spins, _u64p), _p(out, _i64p))
    return out


def locale_idx_of(states, num_locales):
    states = np.ascontiguousarray(states, dtype=np.uint64)
    keys = np.empty(len(states), dtype=np.uint8)
    lib().lso_locale_idx_of(C.c_int64(len(states)), _p(states, _u64p), C.c_int(num_locales), _p(keys, _u8p))
    return keys


def block_to_hashed(arr, masks, P):
    arr = np.ascontiguousarray(arr)
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    counts = np.bincount(masks, minlength=P).astype(np.int64)
    parts = [np.empty(int(c), dtype=arr.dtype) for c in counts]
    dp = (C.c_void_p * P)(*[p.ctypes.data for p in parts])
    lib().lso_block_to_hashed(C.c_int64(len(arr)), _p(masks, _u8p), C.c_int(P), C.c_int(arr.dtype.itemsize),
                              C.c_void_p(arr.ctypes.data), dp, None)
    return parts


def hashed_to_block(parts, masks):
    P = len(parts)
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    parts = [np.ascontiguousarray(p) for p in parts]
    out = np.empty(len(masks), dtype=parts[0].dtype)
    sp = (C.c_void_p * P)(*[p.ctypes.data for p in parts])
    lib().lso_hashed_to_block(C.c_int64(len(masks)), _p(masks, _u8p), C.c_int(P), C.c_int(out.dtype.itemsize), sp,
                              C.c_void_p(out.ctypes.data))
    return out

"""Host-side mirror of the reference's operator interface for the matvec path.

Names, argument meaning and error behaviour follow the Chapel modules they stand in for:
    Basis, Operator, loadConfigFromYaml     /root/reference/src/ForeignTypes.chpl:8-288
    enumerateStates                          /root/reference/src/StatesEnumeration.chpl:516-585
    arrFromBlockToHashed / arrFromHashedToBlock   /root/reference/src/BlockToHashed.chpl:87,
                                             /root/reference/src/HashedToBlock.chpl:67
    matrixVectorProduct / localMatrixVector  /root/reference/src/DistributedMatrixVector.chpl:1055-1093
so that the parity tests read like test/TestMatrixVectorProduct.chpl.  Everything that computes
calls the C ABI of libls_amd.so (include/*.h); torch only provides device memory and streams.
A "BlockVector" here is simply a list with one device tensor per locale (hash partition).
uint64 arrays are held as torch.int64 tensors (bit-identical).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import config as _config
from ._lib import LsAmdError

__all__ = [
    "Basis", "Operator", "LsAmdError", "loadConfigFromYaml", "loadConfigFromDict", "enumerateStates",
    "arrFromBlockToHashed", "arrFromHashedToBlock", "matrixVectorProduct", "localMatrixVector",
    "localeIdxOf", "hash64_01", "MatvecPlan", "ReplicatedPlan", "build_library", "fillRandom",
    "Communicator", "DistMatvec", "ReplMatvec",
]


def build_library(verbose: bool = False):
    """hipcc --offload-arch=gfx950 build of libls_amd.so in-tree (cross-compiles without a GPU)."""
    import os
    import subprocess

    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    subprocess.check_call(["make", "-C", csrc] + ([] if verbose else ["-s"]))
    return _lib.LIB_PATH


def _torch():
    import torch

    return torch


def _stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def hash64_01(x: int) -> int:
    """/root/reference/src/StatesEnumeration.chpl:122-127"""
    return int(_lib.load().ls_amd_hash64_01(C.c_uint64(x)))


def localeIdxOf(basisState: int, numLocales: int) -> int:
    """/root/reference/src/StatesEnumeration.chpl:133-136"""
    return int(_lib.load().ls_amd_locale_idx_of(C.c_uint64(basisState), numLocales))


class Basis:
    """RAII wrapper of ls_hs_basis (ForeignTypes.chpl:8-117)."""

    def __init__(self, payload, owning=True, spec=None):
        self.payload = payload
        self.owning = owning
        self.spec = spec
        self._host_reps = None  # keeps a numpy array alive for uncheckedSetRepresentatives

    @staticmethod
    def fromSpec(spec: _config.BasisSpec) -> "Basis":
        L = _lib.load()
        ng = len(spec.permutations)
        perms = (C.c_int * max(1, ng * spec.number_sites))(*[v for p in spec.permutations for v in p])
        sectors = (C.c_int * max(1, ng))(*spec.sectors)
        p = L.ls_hs_create_spin_basis(spec.number_sites, spec.hamming_weight, spec.spin_inversion, ng, perms, sectors)
        if not p:
            raise LsAmdError(L.ls_amd_last_error().decode())
        return Basis(p, True, spec)

    def __del__(self):
        try:
            if self.owning and self.payload:
                _lib.load().ls_hs_destroy_basis(self.payload)
                self.payload = None
        except Exception:
            pass

    # accessors, same names as the Chapel record ------------------------------------------------
    def numberSites(self): return int(self.payload.contents.number_sites)
    def numberBits(self): return int(_lib.load().ls_hs_basis_number_bits(self.payload))
    def numberWords(self): return int(_lib.load().ls_hs_basis_number_words(self.payload))
    def spinInversion(self): return int(self.payload.contents.spin_inversion)
    def isStateIndexIdentity(self): return bool(self.payload.contents.state_index_is_identity)
    def requiresProjection(self): return bool(self.payload.contents.requires_projection)
    def isHammingWeightFixed(self): return bool(_lib.load().ls_hs_basis_has_fixed_hamming_weight(self.payload))
    def hasSpinInversionSymmetry(self): return bool(_lib.load().ls_hs_basis_has_spin_inversion_symmetry(self.payload))
    def hasPermutationSymmetries(self): return bool(_lib.load().ls_hs_basis_has_permutation_symmetries(self.payload))
    def minStateEstimate(self): return int(_lib.load().ls_hs_min_state_estimate(self.payload))
    def maxStateEstimate(self): return int(_lib.load().ls_hs_max_state_estimate(self.payload))
    def groupOrder(self): return int(_lib.load().ls_amd_basis_group_order(self.payload))

    def build(self):
        """ls_hs_basis_build (ForeignTypes.chpl:72): dispatches to the registered enumerate_states
        kernel, i.e. ls_chpl_enumerate_representatives -> the HIP enumeration."""
        _lib.require_device()
        _lib.load().ls_hs_basis_build(self.payload)
        _lib.raise_pending_halt()

    def uncheckedSetRepresentatives(self, representatives: np.ndarray):
        """ForeignTypes.chpl:74-77 (borrows host memory)."""
        arr = np.ascontiguousarray(representatives, dtype=np.uint64)
        self._host_reps = arr
        ext = _lib.ChplExternalArray(arr.ctypes.data, arr.size, None)
        _lib.load().ls_hs_unchecked_set_representatives(self.payload, C.byref(ext))

    def representatives(self) -> np.ndarray:
        """ForeignTypes.chpl:111-116; halts with "basis is not built"."""
        rs = self.payload.contents.representatives
        if not rs.elts:
            raise LsAmdError("halt: basis is not built")
        buf = (C.c_uint64 * rs.num_elts).from_address(rs.elts)
        return np.frombuffer(buf, dtype=np.uint64)


class Operator:
    """RAII wrapper of ls_hs_operator (ForeignTypes.chpl:154-259)."""

    def __init__(self, payload, owning=True):
        self.payload = payload
        self.owning = owning
        self.basis = Basis(payload.contents.basis, owning=False)
        self._plans = {}

    @staticmethod
    def fromSpec(basis: Basis, spec: _config.OperatorSpec) -> "Operator":
        L = _lib.load()
        n = len(spec.terms)
        v = np.zeros(2 * max(n, 1), dtype=np.float64)
        m = np.zeros(max(n, 1), dtype=np.uint64)
        r = np.zeros(max(n, 1), dtype=np.uint64)
        x = np.zeros(max(n, 1), dtype=np.uint64)
        s = np.zeros(max(n, 1), dtype=np.uint64)
        for i, (vv, mm, rr, xx, ss) in enumerate(spec.terms):
            v[2 * i], v[2 * i + 1] = vv.real, vv.imag
            m[i], r[i], x[i], s[i] = mm, rr, xx, ss
        p = L.ls_hs_create_operator_from_terms(
            basis.payload, n, v.ctypes.data_as(_lib.c_f64p), m.ctypes.data_as(_lib.c_u64p),
            r.ctypes.data_as(_lib.c_u64p), x.ctypes.data_as(_lib.c_u64p), s.ctypes.data_as(_lib.c_u64p))
        if not p:
            raise LsAmdError(L.ls_amd_last_error().decode())
        op = Operator(p, True)
        op.basis.spec = basis.spec
        return op

    def clear_plans(self):
        """destroy the cached matvec plans (and release the HBM they pin)"""
        for pl in list(self._plans.values()):
            pl.destroy()
        self._plans.clear()

    def __del__(self):
        try:
            self.clear_plans()
            if self.owning and self.payload:
                _lib.load().ls_hs_destroy_operator(self.payload)
                self.payload = None
        except Exception:
            pass

    def numberDiagTerms(self):
        p = self.payload.contents.diag_terms
        return int(p.contents.number_terms) if p else 0

    def numberOffDiagTerms(self):
        return int(_lib.load().ls_hs_operator_max_number_off_diag(self.payload))

    @property
    def isHermitian(self): return bool(_lib.load().ls_hs_operator_is_hermitian(self.payload))

    @property
    def isReal(self): return bool(_lib.load().ls_hs_operator_is_real(self.payload))

    # host-pointer entry points of the kernel table ----------------------------------------------
    def __matmul__(self, x: np.ndarray) -> np.ndarray:
        """kernels->matrix_vector_product (ls_chpl_matrix_vector_product, DMV:1095-1110) on host
        f64 arrays; y starts zeroed."""
        _lib.require_device()
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        _lib.load().ls_chpl_matrix_vector_product(self.payload, 1, x.ctypes.data_as(_lib.c_f64p), y.ctypes.data_as(_lib.c_f64p))
        _lib.raise_pending_halt()
        return y

    def applyDiag(self, alphas: np.ndarray) -> np.ndarray:
        """ls_chpl_operator_apply_diag (BatchedOperator.chpl:217-234)."""
        _lib.require_device()
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        out = _lib.ChplExternalArray()
        _lib.load().ls_chpl_operator_apply_diag(self.payload, alphas.size, alphas.ctypes.data_as(_lib.c_u64p), C.byref(out), 1)
        _lib.raise_pending_halt()
        return _take_external(out, np.float64, 1)

    def applyOffDiag(self, alphas: np.ndarray):
        """ls_chpl_operator_apply_off_diag (BatchedOperator.chpl:236-275) -> (betas, coeffs, offsets)."""
        _lib.require_device()
        alphas = np.ascontiguousarray(alphas, dtype=np.uint64)
        b, c, o = _lib.ChplExternalArray(), _lib.ChplExternalArray(), _lib.ChplExternalArray()
        _lib.load().ls_chpl_operator_apply_off_diag(self.payload, alphas.size, alphas.ctypes.data_as(_lib.c_u64p),
                                                    C.byref(b), C.byref(c), C.byref(o), 1)
        _lib.raise_pending_halt()
        offsets = _take_external(o, np.int64, 1)
        betas = _take_external(b, np.uint64, 1)
        coeffs = _take_external(c, np.complex128, 1)
        return betas, coeffs, offsets


_libc_free = None


def _take_external(arr: "_lib.ChplExternalArray", dtype, _unused):
    """copies a callee-allocated chpl_external_array into numpy and frees it through `freer`."""
    n = int(arr.num_elts)
    if not arr.elts or n == 0:
        return np.zeros(0, dtype=dtype)
    itemsize = np.dtype(dtype).itemsize
    buf = (C.c_char * (n * itemsize)).from_address(arr.elts)
    out = np.frombuffer(buf, dtype=dtype).copy()
    if arr.freer:
        C.CFUNCTYPE(None, C.c_void_p)(arr.freer)(arr.elts)
    return out


def loadConfigFromDict(cfg: dict, hamiltonian: bool = False, observables: bool = False):
    bspec = _config.parse_basis(cfg)
    basis = Basis.fromSpec(bspec)
    h = None
    if hamiltonian:
        if cfg.get("hamiltonian") is None:
            raise LsAmdError("halt: config does not contain a Hamiltonian")  # ForeignTypes.chpl:273-274
        h = Operator.fromSpec(basis, _config.parse_operator(cfg["hamiltonian"]))
    obs = [Operator.fromSpec(basis, _config.parse_operator(o)) for o in (cfg.get("observables") or [])] if observables else []
    if not hamiltonian and not observables:
        return basis
    if hamiltonian and not observables:
        return basis, h
    if not hamiltonian and observables:
        return basis, obs
    return basis, h, obs


def loadConfigFromYaml(filename: str, hamiltonian: bool = False, observables: bool = False):
    """ForeignTypes.chpl:261-288, step by step: ls_hs_load_yaml_config (the C library's own loader, csrc/yaml.c) -> clone
    basis / hamiltonian / observables -> ls_hs_destroy_yaml_config."""
    L = _lib.load()
    conf = L.ls_hs_load_yaml_config(filename.encode())
    if not conf:
        raise LsAmdError(f"halt: failed to load Config from '{filename}' ({L.ls_amd_last_error().decode()})")
    try:
        c = conf.contents
        basis = Basis(L.ls_hs_clone_basis(c.basis), owning=True)
        h = None
        if hamiltonian:
            if not c.hamiltonian:
                raise LsAmdError(f"halt: '{filename}' does not contain a Hamiltonian")  # ForeignTypes.chpl:273-274
            h = Operator(L.ls_hs_clone_operator(c.hamiltonian), owning=True)
        obs = [Operator(L.ls_hs_clone_operator(c.observables[i]), owning=True) for i in range(c.number_observables)] if observables else []
    finally:
        L.ls_hs_destroy_yaml_config(conf)
    if not hamiltonian and not observables:
        return basis
    if hamiltonian and not observables:
        return basis, h
    if not hamiltonian and observables:
        return basis, obs
    return basis, h, obs


# ------------------------------------------------------------------------------------------------
# device-side pieces
# ------------------------------------------------------------------------------------------------

def _adopt(ptr: int, count: int, torch_dtype):
    """copies a library-owned device array into a torch tensor and frees the original."""
    torch = _torch()
    L = _lib.load()
    t = torch.empty(count, dtype=torch_dtype, device="cuda")
    try:
        if count > 0:
            _lib.check(L.ls_amd_memcpy_d2d(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), count * t.element_size(), _stream_ptr()))
            _lib.check(L.ls_amd_synchronize(_stream_ptr()))
    finally:
        if ptr:
            L.ls_amd_free(C.c_void_p(ptr))
    return t


def enumerateStates(basis: Basis, numLocales: int = 1):
    """enumerateStates(basis, masks) (StatesEnumeration.chpl:580-585): returns
    (basisStates, masks) -- basisStates[p] = ascending representatives owned by locale p
    (hash64_01 % numLocales), masks = owner of every state in global ascending order."""
    _lib.require_device()
    torch = _torch()
    L = _lib.load()
    d_states, d_masks, count = C.c_void_p(), C.c_void_p(), C.c_int64()
    _lib.check(L.ls_amd_enumerate_states(basis.payload, numLocales, C.byref(d_states), C.byref(d_masks), C.byref(count), _stream_ptr()))
    states = _adopt(d_states.value, count.value, torch.int64)
    masks = _adopt(d_masks.value, count.value, torch.uint8)
    parts = [states] if numLocales == 1 else arrFromBlockToHashed(states, masks, numLocales)
    return parts, masks


def _mask_counts(masks, numLocales):
    counts = (C.c_int64 * numLocales)()
    _lib.check(_lib.load().ls_amd_mask_counts(masks.numel(), C.c_void_p(masks.data_ptr()), numLocales, counts, _stream_ptr()))
    return [int(c) for c in counts]


def arrFromBlockToHashed(arr, masks, numLocales: int):
    """BlockToHashed.chpl:87-208: stable partition of a block-order device tensor (8- or 16-byte
    elements) into one tensor per locale."""
    torch = _torch()
    _lib.require_device()
    assert arr.is_cuda and masks.is_cuda and arr.is_contiguous()
    n = arr.numel()
    assert masks.numel() == n
    counts = _mask_counts(masks, numLocales)
    parts = [torch.empty(c, dtype=arr.dtype, device=arr.device) for c in counts]
    dest = (C.c_void_p * numLocales)(*[p.data_ptr() for p in parts])
    _lib.check(_lib.load().ls_amd_block_to_hashed(n, C.c_void_p(masks.data_ptr()), numLocales, arr.element_size(),
                                                  C.c_void_p(arr.data_ptr()), dest, _stream_ptr()))
    return parts


def arrFromHashedToBlock(parts, masks):
    """HashedToBlock.chpl:67-153: inverse of arrFromBlockToHashed."""
    torch = _torch()
    _lib.require_device()
    numLocales = len(parts)
    n = masks.numel()
    out = torch.empty(n, dtype=parts[0].dtype, device=parts[0].device)
    src = (C.c_void_p * numLocales)(*[p.data_ptr() for p in parts])
    _lib.check(_lib.load().ls_amd_hashed_to_block(n, C.c_void_p(masks.data_ptr()), numLocales, parts[0].element_size(),
                                                  src, C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


class MatvecPlan:
    """ls_amd_plan: binds an Operator to a partition layout (include/ls_amd.h)."""

    MODES = {"auto": 0, "push": 1, "pull": 2}

    def __init__(self, matrix: Operator, representatives, dtype, my_partition: int = -1,
                 num_partitions: int | None = None, num_rounds: int = 0, mode: str = "auto"):
        torch = _torch()
        _lib.require_device()
        L = _lib.load()
        self.matrix = matrix
        reps = list(representatives) if my_partition < 0 else [representatives]
        self.reps = reps  # borrowed by the plan: keep alive
        self.P = num_partitions if num_partitions is not None else len(reps)
        self.me = my_partition
        self.cplx = dtype in (torch.complex128, "c128")
        n = len(reps)
        ptrs = (C.c_void_p * n)(*[r.data_ptr() for r in reps])
        counts = (C.c_int64 * n)(*[r.numel() for r in reps])
        h = C.c_void_p()
        _lib.check(L.ls_amd_plan_create(C.byref(h), matrix.payload, 1 if self.cplx else 0, self.P, my_partition,
                                        ptrs, counts, num_rounds, self.MODES[mode], _stream_ptr()))
        self.h = h

    def destroy(self):
        if getattr(self, "h", None):
            _lib.load().ls_amd_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    @property
    def kernel(self): return _lib.load().ls_amd_plan_kernel_name(self.h).decode()
    @property
    def num_rounds(self): return int(_lib.load().ls_amd_plan_num_rounds(self.h))
    @property
    def nnz(self): return int(_lib.load().ls_amd_plan_nnz(self.h))
    @property
    def packet_bytes(self): return int(_lib.load().ls_amd_plan_packet_bytes(self.h))
    @property
    def row_bytes(self): return int(_lib.load().ls_amd_plan_row_bytes(self.h))
    @property
    def key_bytes(self):
        """8: packets carry the state; 4: pre-indexed packets (u32 index at the destination; include/ls_amd.h)"""
        return int(_lib.load().ls_amd_plan_key_bytes(self.h))
    @property
    def packet_index_bytes(self): return int(_lib.load().ls_amd_plan_packet_index_bytes(self.h))

    def segment_bytes(self, count: int) -> int:
        """bytes of a segment of `count` packets: [key x count, padded to 8 bytes][value x count]"""
        return int(_lib.load().ls_amd_plan_segment_bytes(self.h, int(count)))

    def segment_value_offset(self, count: int) -> int:
        return int(_lib.load().ls_amd_plan_segment_value_offset(self.h, int(count)))

    def cache_slots(self, max_bytes: int = 0) -> int:
        """keep the resolved packet streams (slot of every partner, 5 B per non-zero on the Heisenberg models) across matvecs:
        the first matvec resolves them, later ones only gather -- for plans that are applied many times (eigensolvers).  Opt-in,
        not matrix-free; returns the rows cached (0: nothing, e.g. an unprojected basis or no room)."""
        rows = int(_lib.load().ls_amd_plan_cache_slots(self.h, int(max_bytes)))
        if rows < 0:
            _lib.check(-1)
        return rows

    @property
    def slot_cache(self):
        """(rows, bytes) held by the slot cache"""
        r, b = C.c_int64(0), C.c_int64(0)
        _lib.load().ls_amd_plan_slot_cache_rows(self.h, C.byref(r), C.byref(b))
        return int(r.value), int(b.value)

    def send_counts(self, rnd: int):
        c = (C.c_int64 * self.P)()
        _lib.check(_lib.load().ls_amd_plan_send_counts(self.h, rnd, c))
        return [int(v) for v in c]

    def matvec(self, x, y, check: bool = True):
        xs = (C.c_void_p * len(x))(*[t.data_ptr() for t in x])
        ys = (C.c_void_p * len(y))(*[t.data_ptr() for t in y])
        _lib.check(_lib.load().ls_amd_matvec(self.h, xs, ys, _stream_ptr()))
        if check:
            self.check()

    def check(self):
        _lib.check(_lib.load().ls_amd_plan_check(self.h, _stream_ptr()))

    def inject_fault(self) -> bool:
        """test hook (ls_amd_test_corrupt_plan): the row kernel skips one row from now on; False when the plan has no tile map"""
        return bool(_lib.load().ls_amd_test_corrupt_plan(self.h))

    def enable_timing(self, max_samples: int):
        _lib.check(_lib.load().ls_amd_plan_enable_timing(self.h, max_samples))

    def kernel_times_ms(self, capacity: int = 4096):
        buf = (C.c_float * capacity)()
        n = C.c_int()
        _lib.check(_lib.load().ls_amd_plan_kernel_times(self.h, buf, capacity, C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def enable_stage_timing(self, max_events: int = 65536):
        """the reference's kDisplayTimings (DMV:1028-1052): per-stage device time"""
        _lib.check(_lib.load().ls_amd_plan_enable_stage_timing(self.h, max_events))

    def stage_times(self):
        ms = (C.c_double * 7)()
        calls = (C.c_int64 * 7)()
        mv = C.c_int64()
        _lib.check(_lib.load().ls_amd_plan_stage_times(self.h, ms, calls, C.byref(mv)))
        names = ("localDiagonal", "hashRefresh", "rowKernel", "producers", "exchangeWait", "consumers", "resultsToOwners")
        return {n: (float(ms[i]), int(calls[i])) for i, n in enumerate(names)}, int(mv.value)

    def timing_report(self) -> str:
        buf = C.create_string_buffer(4096)
        _lib.check(_lib.load().ls_amd_plan_timing_report(self.h, buf, 4096))
        return buf.value.decode("utf-8")

    def diag(self, x, y):
        _lib.check(_lib.load().ls_amd_diag(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _stream_ptr()))

    def generate(self, rnd, x, y, send):
        _lib.check(_lib.load().ls_amd_generate(self.h, rnd, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                               C.c_void_p(send.data_ptr() if send is not None else 0), _stream_ptr()))

    def scatter(self, n, betas_ptr: int, vals_ptr: int, y):
        _lib.check(_lib.load().ls_amd_scatter(self.h, n, C.c_void_p(betas_ptr), C.c_void_p(vals_ptr),
                                              C.c_void_p(y.data_ptr()), _stream_ptr()))

    def scatter_round(self, counts, offsets, recv_ptr: int, y):
        """all segments of one round's receive buffer in one launch (ls_amd_scatter_round)"""
        n = len(counts)
        _lib.check(_lib.load().ls_amd_scatter_round(self.h, n, (C.c_int64 * n)(*counts), (C.c_int64 * n)(*offsets), C.c_void_p(recv_ptr),
                                                    C.c_void_p(y.data_ptr()), _stream_ptr()))


class _BorrowedPlan(MatvecPlan):
    """view of a plan owned by another object (ls_amd_dist): same accessors, no destroy."""

    def __init__(self, handle, owner, P, me):  # noqa: super().__init__ deliberately not called
        self.h, self.owner, self.P, self.me = handle, owner, P, me

    def destroy(self):
        self.h = None


class Communicator:
    """ls_amd_comm: RCCL communicator owned by the C host (include/ls_amd.h).  rank == locale index."""

    def __init__(self, size: int, rank: int, unique_id: bytes):
        _lib.require_device()
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(_lib.load().ls_amd_comm_create(C.byref(h), size, rank, buf))
        self.h, self.size, self.rank = h, size, rank

    @staticmethod
    def local_group(size: int):
        """`size` loop-back communicators in this process (one per host thread): the multi-rank logic of the C host on a
        one-GPU box, where RCCL refuses more than one rank per device (test infrastructure)"""
        _lib.require_device()
        arr = (C.c_void_p * size)()
        _lib.check(_lib.load().ls_amd_comm_create_local(arr, size))
        out = []
        for r in range(size):
            c = Communicator.__new__(Communicator)
            c.h, c.size, c.rank = C.c_void_p(arr[r]), size, r
            out.append(c)
        return out

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().ls_amd_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def from_torch(group=None) -> "Communicator":
        """bootstrap through an existing torch.distributed group (any backend): rank 0 creates the id,
        the group's object broadcast carries it -- the role MPI_Bcast plays for a C caller."""
        import torch.distributed as dist

        rank, size = dist.get_rank(group), dist.get_world_size(group)
        box = [Communicator.unique_id() if rank == 0 else None]
        if size > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return Communicator(size, rank, box[0])

    def wait(self, timeout_s: float = 0.0):
        """ls_amd_comm_wait: until the current stream and the exchange stream have drained; raises after timeout_s (<= 0: the
        watchdog's deadline, LS_AMD_COMM_WATCHDOG_S) instead of hanging in a synchronisation"""
        _lib.check(_lib.load().ls_amd_comm_wait(self.h, _stream_ptr(), float(timeout_s)))

    def rccl_count(self) -> int:
        """the communicator size as RCCL reports it (ncclCommCount); 0 for a loop-back group"""
        return int(_lib.load().ls_amd_comm_rccl_count(self.h))

    def set_default(self):
        """the communicator of primmeGlobalSumReal / primmeBroadcastReal / ls_chpl_primme_matvec"""
        _lib.load().ls_amd_set_default_comm(self.h)

    def allreduce_sum(self, t):
        torch = _torch()
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        _lib.check(_lib.load().ls_amd_comm_allreduce_sum_f64(self.h, C.c_void_p(t.data_ptr()), t.numel(), _stream_ptr()))
        return t

    def allreduce_max(self, t):
        torch = _torch()
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
        _lib.check(_lib.load().ls_amd_comm_allreduce_max_i64(self.h, C.c_void_p(t.data_ptr()), t.numel(), _stream_ptr()))
        return t

    def broadcast(self, t, root: int = 0):
        assert t.is_cuda and t.is_contiguous()
        _lib.check(_lib.load().ls_amd_comm_broadcast(self.h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), root,
                                                     _stream_ptr()))
        return t

    def destroy(self):
        if getattr(self, "h", None):
            L = _lib.load()
            if L.ls_amd_default_comm() == self.h.value:
                L.ls_amd_set_default_comm(None)
            L.ls_amd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DistMatvec:
    """ls_amd_dist: matrixVectorProduct for this rank's block, exchange inside the C host (RCCL)."""

    def __init__(self, comm: Communicator, matrix: Operator, representatives, dtype, num_rounds: int = 0):
        torch = _torch()
        _lib.require_device()
        self.comm, self.matrix, self.reps = comm, matrix, representatives  # borrowed by the C object: keep alive
        self.cplx = dtype in (torch.complex128, "c128")
        h = C.c_void_p()
        _lib.check(_lib.load().ls_amd_dist_create(C.byref(h), comm.h, matrix.payload, 1 if self.cplx else 0,
                                                  C.c_void_p(representatives.data_ptr()), representatives.numel(),
                                                  num_rounds, _stream_ptr()))
        self.h = h
        self.plan = _BorrowedPlan(C.c_void_p(_lib.load().ls_amd_dist_plan(h)), self, comm.size, comm.rank)

    @property
    def num_rounds(self): return int(_lib.load().ls_amd_dist_num_rounds(self.h))
    @property
    def exchange_bytes(self): return int(_lib.load().ls_amd_dist_exchange_bytes(self.h))

    def matvec(self, x, y, check: bool = True):
        _lib.check(_lib.load().ls_amd_dist_matvec(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _stream_ptr()))
        if check:
            self.plan.check()

    def inject_fault(self) -> bool:
        """test hook (ls_amd_test_corrupt_dist): misplace one received segment; False when this rank receives none"""
        return bool(_lib.load().ls_amd_test_corrupt_dist(self.h))

    def destroy(self):
        if getattr(self, "h", None):
            self.plan.destroy()
            _lib.load().ls_amd_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ReplMatvec:
    """ls_amd_repl: matrixVectorProduct for this rank's block with the replicated-x exchange inside the C host (RCCL)."""

    def __init__(self, comm: Communicator, matrix: Operator, reps_global, masks, dtype):
        torch = _torch()
        _lib.require_device()
        self.comm, self.matrix, self.reps_global, self.masks = comm, matrix, reps_global, masks  # borrowed: keep alive
        self.cplx = dtype in (torch.complex128, "c128")
        assert masks.dtype == torch.uint8 and masks.is_cuda and masks.numel() == reps_global.numel()
        h = C.c_void_p()
        _lib.check(_lib.load().ls_amd_repl_create(C.byref(h), comm.h, matrix.payload, 1 if self.cplx else 0,
                                                  C.c_void_p(reps_global.data_ptr()), C.c_void_p(masks.data_ptr()),
                                                  reps_global.numel(), _stream_ptr()))
        self.h = h
        self.plan = _BorrowedPlan(C.c_void_p(_lib.load().ls_amd_repl_plan(h)), self, comm.size, comm.rank)

    @property
    def exchange_bytes(self): return int(_lib.load().ls_amd_repl_exchange_bytes(self.h))
    @property
    def x_in_bytes(self): return int(_lib.load().ls_amd_repl_x_in_bytes(self.h))

    def matvec(self, x, y, check: bool = True):
        _lib.check(_lib.load().ls_amd_repl_matvec(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _stream_ptr()))
        if check:
            self.plan.check()

    def inject_fault(self) -> bool:
        """test hook (ls_amd_test_corrupt_repl): this rank's own rows come back one element late"""
        return bool(_lib.load().ls_amd_test_corrupt_repl(self.h))

    def destroy(self):
        if getattr(self, "h", None):
            self.plan.destroy()
            _lib.load().ls_amd_repl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def fillRandom(states, seed: int, dtype):
    """deterministic vector keyed by the basis states (same logical vector for every partitioning)."""
    torch = _torch()
    _lib.require_device()
    out = torch.empty(states.numel(), dtype=dtype, device=states.device)
    _lib.check(_lib.load().ls_amd_fill_random(states.numel(), C.c_void_p(states.data_ptr()), C.c_uint64(seed),
                                              1 if dtype == torch.complex128 else 0, C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


class ReplicatedPlan:
    """ls_amd replicated-x plan: this process owns partition `my_partition`'s rows and is handed the
    whole x in global ascending order (include/ls_amd.h)."""

    def __init__(self, matrix: "Operator", reps_local, reps_global, dtype, num_partitions: int, my_partition: int):
        torch = _torch()
        _lib.require_device()
        self.matrix, self.reps_local, self.reps_global = matrix, reps_local, reps_global
        self.cplx = dtype in (torch.complex128, "c128")
        h = C.c_void_p()
        _lib.check(_lib.load().ls_amd_plan_create_replicated(
            C.byref(h), matrix.payload, 1 if self.cplx else 0, num_partitions, my_partition,
            C.c_void_p(reps_local.data_ptr()), reps_local.numel(), C.c_void_p(reps_global.data_ptr()), reps_global.numel(),
            _stream_ptr()))
        self.h = h

    destroy = MatvecPlan.destroy
    __del__ = MatvecPlan.__del__
    kernel = MatvecPlan.kernel
    check = MatvecPlan.check
    enable_timing = MatvecPlan.enable_timing
    kernel_times_ms = MatvecPlan.kernel_times_ms
    enable_stage_timing = MatvecPlan.enable_stage_timing
    stage_times = MatvecPlan.stage_times
    timing_report = MatvecPlan.timing_report

    def matvec(self, x_global, y_local, check: bool = True):
        _lib.check(_lib.load().ls_amd_matvec_replicated(self.h, C.c_void_p(x_global.data_ptr()), C.c_void_p(y_local.data_ptr()), _stream_ptr()))
        if check:
            self.check()


def _plan_for(matrix: Operator, representatives, dtype, mode="auto"):
    key = (tuple(int(r.data_ptr()) for r in representatives), tuple(int(r.numel()) for r in representatives), str(dtype), mode)
    pl = matrix._plans.pop(key, None)
    if pl is None:
        # a plan pins its representative tensors and device tables (hash table, norms, tile map, partner-rank cache:
        # several GB for the large chains): keep only the two most recently used ones per operator
        while len(matrix._plans) >= 2:
            oldest = next(iter(matrix._plans))
            matrix._plans.pop(oldest).destroy()
        pl = MatvecPlan(matrix, representatives, dtype, mode=mode)
    matrix._plans[key] = pl  # (re)inserted last: dicts keep insertion order = LRU order
    return pl


def matrixVectorProduct(matrix: Operator, x, y, representatives, mode: str = "auto", check: bool = True):
    """matrixVectorProduct(matrix, x, y, representatives) (DMV:1072-1093) with all locales as logical
    partitions on the current device.  x, y, representatives: lists with one device tensor per
    locale (f64 or c128; representatives int64-viewed uint64).  y is overwritten by the diagonal
    pass when the operator has diagonal terms, then accumulated into."""
    torch = _torch()
    if isinstance(x, torch.Tensor):
        x, y, representatives = [x], [y], [representatives]
    if len(x) != len(representatives) or len(y) != len(representatives):
        raise LsAmdError("x, y and representatives must have one block per locale")
    for a, b, r in zip(x, y, representatives):
        if a.dtype != b.dtype or a.numel() != r.numel() or b.numel() != r.numel():
            raise LsAmdError("block shapes / dtypes do not match")
    pl = _plan_for(matrix, representatives, x[0].dtype, mode)
    pl.matvec(x, y, check=check)
    return pl


def localMatrixVector(matrix: Operator, x, y, representatives, mode: str = "auto"):
    """localMatrixVector (DMV:1055-1070): the single-locale case."""
    return matrixVectorProduct(matrix, [x], [y], [representatives], mode=mode)

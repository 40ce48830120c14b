"""ctypes binding of libls_amd.so (C host side + HIP kernels).

The library is the product: there is NO Python/NumPy/torch fallback for any compute entry point.
If the shared object is missing, or no HIP device is visible when a device function is called,
this module raises -- loudly -- instead of computing anything on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LS_AMD_LIB: another build of the same library next to the product (profiling / A/B builds: libls_amd_ablate.so, libls_amd_skew.so)
LIB_PATH = os.path.join(_HERE, os.environ.get("LS_AMD_LIB", "libls_amd.so"))

c_u64p = C.POINTER(C.c_uint64)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
c_intp = C.POINTER(C.c_int)


class ChplExternalArray(C.Structure):
    """chpl_external_array (/root/reference/src/FFI.chpl:36-55)."""

    _fields_ = [("elts", C.c_void_p), ("num_elts", C.c_uint64), ("freer", C.c_void_p)]


class LsHsBasis(C.Structure):
    """Prefix of ls_hs_basis (/root/reference/src/FFI.chpl:94-105)."""

    _fields_ = [
        ("number_sites", C.c_int),
        ("number_particles", C.c_int),
        ("number_up", C.c_int),
        ("particle_type", C.c_int),
        ("spin_inversion", C.c_int),
        ("state_index_is_identity", C.c_bool),
        ("requires_projection", C.c_bool),
        ("kernels", C.c_void_p),
        ("representatives", ChplExternalArray),
    ]


class LsHsNonbranchingTerms(C.Structure):
    _fields_ = [
        ("number_terms", C.c_int),
        ("number_bits", C.c_int),
        ("v", C.c_void_p),
        ("m", C.c_void_p),
        ("l", C.c_void_p),
        ("r", C.c_void_p),
        ("x", C.c_void_p),
        ("s", C.c_void_p),
    ]


class LsHsOperator(C.Structure):
    """Prefix of ls_hs_operator (/root/reference/src/FFI.chpl:114-119)."""

    _fields_ = [
        ("basis", C.POINTER(LsHsBasis)),
        ("off_diag_terms", C.POINTER(LsHsNonbranchingTerms)),
        ("diag_terms", C.POINTER(LsHsNonbranchingTerms)),
    ]


class YamlConfig(C.Structure):  # ls_hs_yaml_config (include/ls_hs.h; /root/reference/src/FFI.chpl:121-126)
    _fields_ = [("basis", C.POINTER(LsHsBasis)), ("hamiltonian", C.POINTER(LsHsOperator)), ("number_observables", C.c_int),
                ("observables", C.POINTER(C.POINTER(LsHsOperator)))]


class BoundaryStats(C.Structure):  # struct ls_amd_boundary_stats (include/ls_amd.h)
    _fields_ = [("calls", C.c_int64), ("columns", C.c_int64), ("bytes_h2d", C.c_int64), ("bytes_d2h", C.c_int64),
                ("device_x", C.c_int64), ("device_y", C.c_int64)]


class LsAmdError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LsAmdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    try:
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already
        # mapped when libls_amd.so resolves libamdhip64.so.7, or the process ends up with two runtimes
        # and this library sees no device.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    bp, op = C.POINTER(LsHsBasis), C.POINTER(LsHsOperator)
    vp = C.c_void_p
    sigs = {
        "ls_amd_last_error": (C.c_char_p, []),
        "ls_amd_set_error_handler": (None, [vp]),
        "ls_amd_device_count": (C.c_int, []),
        "ls_amd_set_device": (C.c_int, [C.c_int]),
        "ls_amd_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
        "ls_amd_free": (C.c_int, [vp]),
        "ls_amd_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
        "ls_amd_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
        "ls_amd_memcpy_d2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "ls_amd_memset": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
        "ls_amd_synchronize": (C.c_int, [vp]),
        "ls_amd_stream_copy": (C.c_int, [vp, vp, C.c_int64, vp]),
        "ls_amd_stream_read": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp]),
        "ls_amd_orth_max_rows": (C.c_int, []),
        "ls_amd_orth_pass": (C.c_int, [C.c_int, C.c_int64, vp, C.c_int64, vp, vp, vp, vp]),
        "ls_amd_basis_rotate": (C.c_int, [C.c_int, C.c_int, C.c_int64, vp, C.c_int64, vp, vp]),
        "ls_amd_pointer_kind": (C.c_int, [vp]),
        "ls_amd_host_register": (C.c_int, [vp, C.c_size_t]),
        "ls_amd_host_unregister": (C.c_int, [vp]),
        "ls_amd_boundary_stats_get": (None, [C.POINTER(BoundaryStats), C.c_int]),
        "ls_amd_hash64_01": (C.c_uint64, [C.c_uint64]),
        "ls_amd_locale_idx_of": (C.c_int, [C.c_uint64, C.c_int]),
        "ls_amd_plan_create": (C.c_int, [C.POINTER(vp), op, C.c_int, C.c_int, C.c_int, C.POINTER(vp), c_i64p, C.c_int, C.c_int, vp]),
        "ls_amd_plan_create_replicated": (C.c_int, [C.POINTER(vp), op, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int64, vp]),
        "ls_amd_matvec_replicated": (C.c_int, [vp, vp, vp, vp]),
        "ls_amd_plan_destroy": (None, [vp]),
        "ls_amd_plan_num_rounds": (C.c_int, [vp]),
        "ls_amd_plan_kernel_name": (C.c_char_p, [vp]),
        "ls_amd_plan_send_counts": (C.c_int, [vp, C.c_int, c_i64p]),
        "ls_amd_plan_packet_bytes": (C.c_int, [vp]),
        "ls_amd_plan_row_bytes": (C.c_int, [vp]),
        "ls_amd_plan_nnz": (C.c_int64, [vp]),
        "ls_amd_plan_cache_slots": (C.c_int64, [vp, C.c_int64]),
        "ls_amd_plan_slot_cache_rows": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "ls_amd_matvec": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), vp]),
        "ls_amd_plan_check": (C.c_int, [vp, vp]),
        "ls_amd_plan_enable_timing": (C.c_int, [vp, C.c_int]),
        "ls_amd_plan_kernel_times": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int, c_intp]),
        "ls_amd_plan_enable_stage_timing": (C.c_int, [vp, C.c_int]),
        "ls_amd_plan_stage_times": (C.c_int, [vp, c_f64p, c_i64p, c_i64p]),
        "ls_amd_plan_timing_report": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
        "ls_amd_fill_random": (C.c_int, [C.c_int64, vp, C.c_uint64, C.c_int, vp, vp]),
        "ls_amd_diag": (C.c_int, [vp, vp, vp, vp]),
        "ls_amd_generate": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
        "ls_amd_scatter": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp]),
        "ls_amd_plan_key_bytes": (C.c_int, [vp]),
        "ls_amd_plan_segment_bytes": (C.c_int64, [vp, C.c_int64]),
        "ls_amd_plan_segment_value_offset": (C.c_int64, [vp, C.c_int64]),
        "ls_amd_plan_packet_index_bytes": (C.c_int64, [vp]),
        "ls_amd_scatter_round": (C.c_int, [vp, C.c_int, c_i64p, c_i64p, vp, vp, vp]),
        "ls_amd_adopt_basis": (C.c_int, [bp, C.c_int, c_intp, c_intp]),
        "ls_amd_adopt_operator": (C.c_int, [op]),
        "ls_amd_release": (None, [vp]),
        "ls_amd_comm_available": (C.c_int, []),
        "ls_amd_comm_unique_id": (C.c_int, [vp]),
        "ls_amd_comm_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp]),
        "ls_amd_comm_destroy": (None, [vp]),
        "ls_amd_comm_create_local": (C.c_int, [C.POINTER(vp), C.c_int]),
        "ls_amd_comm_size": (C.c_int, [vp]),
        "ls_amd_comm_rank": (C.c_int, [vp]),
        "ls_amd_comm_rccl_count": (C.c_int, [vp]),
        "ls_amd_test_corrupt_dist": (C.c_int, [vp]),
        "ls_amd_test_set_stream_windows_per_block": (None, [C.c_int]),
        "ls_amd_test_fail_stream_buffers": (None, [C.c_int]),
        "ls_amd_test_fail_dist_streams": (None, [C.c_int]),
        "ls_amd_test_corrupt_repl": (C.c_int, [vp]),
        "ls_amd_test_corrupt_plan": (C.c_int, [vp]),
        "ls_amd_internal_plan_split_active": (C.c_int64, [vp]),
        "ls_amd_comm_wait": (C.c_int, [vp, vp, C.c_double]),
        "ls_amd_test_skew_exchange": (None, [C.c_int, C.c_int64, C.c_int]),
        "ls_amd_comm_test_stall": (C.c_int, [vp, C.c_double]),
        "ls_amd_internal_set_no_packet_index": (None, [C.c_int]),
        "ls_amd_comm_allreduce_sum_f64": (C.c_int, [vp, vp, C.c_int64, vp]),
        "ls_amd_comm_allreduce_max_i64": (C.c_int, [vp, vp, C.c_int64, vp]),
        "ls_amd_comm_broadcast": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
        "ls_amd_set_default_comm": (None, [vp]),
        "ls_amd_default_comm": (vp, []),
        "ls_amd_dist_create": (C.c_int, [C.POINTER(vp), vp, op, C.c_int, vp, C.c_int64, C.c_int, vp]),
        "ls_amd_dist_destroy": (None, [vp]),
        "ls_amd_dist_matvec": (C.c_int, [vp, vp, vp, vp]),
        "ls_amd_dist_plan": (vp, [vp]),
        "ls_amd_dist_exchange_bytes": (C.c_int64, [vp]),
        "ls_amd_dist_num_rounds": (C.c_int, [vp]),
        "ls_amd_repl_create": (C.c_int, [C.POINTER(vp), vp, op, C.c_int, vp, vp, C.c_int64, vp]),
        "ls_amd_repl_destroy": (None, [vp]),
        "ls_amd_repl_matvec": (C.c_int, [vp, vp, vp, vp]),
        "ls_amd_repl_plan": (vp, [vp]),
        "ls_amd_repl_exchange_bytes": (C.c_int64, [vp]),
        "ls_amd_repl_x_in_bytes": (C.c_int64, [vp]),
        "ls_amd_enumerate_states": (C.c_int, [bp, C.c_int, C.POINTER(vp), C.POINTER(vp), c_i64p, vp]),
        "ls_amd_gather": (C.c_int, [C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp]),
        "ls_amd_mask_counts": (C.c_int, [C.c_int64, vp, C.c_int, c_i64p, vp]),
        "ls_amd_block_to_hashed": (C.c_int, [C.c_int64, vp, C.c_int, C.c_int, vp, C.POINTER(vp), vp]),
        "ls_amd_hashed_to_block": (C.c_int, [C.c_int64, vp, C.c_int, C.c_int, C.POINTER(vp), vp, vp]),
        "ls_amd_basis_group_order": (C.c_int, [bp]),
        "ls_amd_basis_apply_group_element": (C.c_uint64, [bp, C.c_int, C.c_uint64]),
        "ls_amd_basis_group_character": (C.c_int, [bp, C.c_int, c_f64p, c_f64p]),
        "ls_amd_test_tilemap": (C.c_int64, [C.c_int64, C.c_int, C.c_int64, C.POINTER(C.POINTER(C.c_uint64))]),
        "ls_amd_test_window_find": (C.c_int, [C.POINTER(C.c_uint64), C.c_int, C.c_uint64]),
        "ls_amd_test_nw_find": (C.c_int, [C.POINTER(C.c_uint64), C.c_int, C.c_uint64]),
        "ls_amd_test_chain_near_table": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int16)]),
        "ls_amd_test_rep_trivial_dihedral": (C.c_uint64, [C.c_uint64, C.c_int, C.c_int, C.c_int]),
        "ls_amd_bench_k4": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp]),
        "ls_amd_test_free": (None, [vp]),
        "ls_amd_test_translation_cosets": (C.c_int, [bp, C.POINTER(C.c_int)]),
        "ls_amd_test_rep_by_cosets": (C.c_uint64, [bp, C.c_uint64]),
        "ls_amd_test_d4_mask": (C.c_int, [bp]),
        "ls_amd_test_gtab_bits": (C.c_int, [C.c_int, C.c_int64]),
        "ls_amd_test_gtab_build": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint64))]),
        "ls_amd_test_gtab_find": (C.c_int64, [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]),
        "ls_amd_test_primme_comminfo_offset": (C.c_int, []),
        "ls_amd_test_primme_sumtype_offset": (C.c_int, []),
        "ls_amd_test_primme_nlocal_offset": (C.c_int, []),
        "ls_amd_test_primme_matrix_offset": (C.c_int, []),
        "ls_hs_init": (None, []),
        "ls_hs_exit": (None, []),
        "ls_hs_create_spin_basis": (bp, [C.c_int, C.c_int, C.c_int, C.c_int, c_intp, c_intp]),
        "ls_hs_clone_basis": (bp, [bp]),
        "ls_hs_destroy_basis": (None, [bp]),
        "ls_hs_create_operator_from_terms": (op, [bp, C.c_int, c_f64p, c_u64p, c_u64p, c_u64p, c_u64p]),
        "ls_hs_clone_operator": (op, [op]),
        "ls_hs_load_yaml_config": (C.POINTER(YamlConfig), [C.c_char_p]),
        "ls_amd_load_yaml_config_from_string": (C.POINTER(YamlConfig), [C.c_char_p]),
        "ls_hs_destroy_yaml_config": (None, [C.POINTER(YamlConfig)]),
        "ls_hs_destroy_operator": (None, [op]),
        "ls_hs_min_state_estimate": (C.c_uint64, [bp]),
        "ls_hs_max_state_estimate": (C.c_uint64, [bp]),
        "ls_hs_basis_number_bits": (C.c_int, [bp]),
        "ls_hs_basis_number_words": (C.c_int, [bp]),
        "ls_hs_basis_has_fixed_hamming_weight": (C.c_bool, [bp]),
        "ls_hs_basis_has_spin_inversion_symmetry": (C.c_bool, [bp]),
        "ls_hs_basis_has_permutation_symmetries": (C.c_bool, [bp]),
        "ls_hs_basis_requires_projection": (C.c_bool, [bp]),
        "ls_hs_fixed_hamming_state_to_index": (C.c_ssize_t, [C.c_uint64]),
        "ls_hs_fixed_hamming_index_to_state": (C.c_uint64, [C.c_ssize_t, C.c_int]),
        "ls_hs_basis_build": (None, [bp]),
        "ls_hs_unchecked_set_representatives": (None, [bp, C.POINTER(ChplExternalArray)]),
        "ls_hs_operator_max_number_off_diag": (C.c_int, [op]),
        "ls_hs_operator_is_hermitian": (C.c_bool, [op]),
        "ls_hs_operator_is_real": (C.c_bool, [op]),
        "ls_hs_state_index": (None, [bp, C.c_ssize_t, c_u64p, C.c_ssize_t, C.POINTER(C.c_ssize_t), C.c_ssize_t]),
        "ls_hs_is_representative": (None, [bp, C.c_ssize_t, c_u64p, C.c_ssize_t, C.POINTER(C.c_uint8), c_f64p]),
        "ls_hs_state_info": (None, [bp, C.c_ssize_t, c_u64p, C.c_ssize_t, c_u64p, C.c_ssize_t, c_f64p, c_f64p]),
        "ls_internal_operator_apply_diag_x1": (None, [op, C.c_ssize_t, c_u64p, c_f64p, c_f64p]),
        "ls_internal_operator_apply_off_diag_x1": (None, [op, C.c_ssize_t, c_u64p, c_u64p, c_f64p, C.POINTER(C.c_ssize_t), c_f64p]),
        "ls_hs_internal_set_chpl_kernels": (None, [vp]),
        "ls_hs_internal_get_chpl_kernels": (vp, []),
        "ls_chpl_init": (None, []),
        "ls_chpl_finalize": (None, []),
        "ls_chpl_init_kernels": (None, []),
        "ls_chpl_matrix_vector_product": (None, [op, C.c_int, c_f64p, c_f64p]),
        "ls_chpl_operator_apply_diag": (None, [op, C.c_int64, c_u64p, C.POINTER(ChplExternalArray), C.c_int64]),
        "ls_chpl_operator_apply_off_diag": (None, [op, C.c_int64, c_u64p, C.POINTER(ChplExternalArray), C.POINTER(ChplExternalArray), C.POINTER(ChplExternalArray), C.c_int64]),
        "ls_chpl_enumerate_representatives": (None, [bp, C.c_uint64, C.c_uint64, C.POINTER(ChplExternalArray)]),
        "ls_chpl_primme_matvec": (None, [vp, c_i64p, vp, c_i64p, c_intp, vp, c_intp]),
        "primmeGlobalSumReal": (None, [vp, vp, c_intp, vp, c_intp]),
        "primmeBroadcastReal": (None, [vp, c_intp, vp, c_intp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError == a declared symbol is not exported: fail loudly
        fn.restype = res
        fn.argtypes = args
    _install_python_error_handler(L)
    _lib = L
    return L


_HANDLER_TYPE = C.CFUNCTYPE(None, C.c_char_p)
_pending_halt = []


@_HANDLER_TYPE
def _on_halt(msg):
    # the reference aborts the process (`halt`); from Python we surface it as an exception raised by
    # the wrapper right after the C call returns
    _pending_halt.append(msg.decode("utf-8", "replace"))


def _install_python_error_handler(L):
    L.ls_amd_set_error_handler(C.cast(_on_halt, C.c_void_p))


def raise_pending_halt():
    if _pending_halt:
        msg = _pending_halt.pop()
        _pending_halt.clear()
        raise LsAmdError("halt: " + msg)


def check(rc):
    if rc != 0:
        raise LsAmdError(load().ls_amd_last_error().decode("utf-8", "replace"))
    raise_pending_halt()


def require_device():
    if load().ls_amd_device_count() < 1:
        raise LsAmdError("no HIP device visible: the matvec path is HIP-only and has no CPU fallback")

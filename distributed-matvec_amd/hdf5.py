"""Minimal HDF5 reader/writer for the reference's on-disk format either side of the matvec path
(/root/reference/src/MyHDF5.chpl:71-144,272-333; produced by /root/reference/input_for_matvec.py:43-46):
    /x, /y               float64, rank 2, shape [batch, N]   (vectors in global ascending order)
    /representatives     uint64,  rank 1
and the groups `basis`, `hamiltonian` the eigensolver driver writes (Diagonalize.chpl:241,252-255).

h5py is not available in this image; the HDF5 C library is (libhdf5.so, located through the usual
loader path or LS_AMD_HDF5_LIB / /opt/conda/lib), and is driven directly through ctypes.  I/O only --
nothing here is on the compute path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_lib = None
hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0x0000, 0x0001, 0x0002
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT = 0, 1


class Hdf5Unavailable(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    candidates = [os.environ.get("LS_AMD_HDF5_LIB"), "libhdf5.so", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so"]
    for cand in candidates:
        if not cand:
            continue
        try:
            L = C.CDLL(cand)
            break
        except OSError:
            continue
    else:
        raise Hdf5Unavailable("libhdf5.so not found (set LS_AMD_HDF5_LIB)")
    for name, res, args in [
        ("H5open", C.c_int, []),
        ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
        ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        ("H5Fclose", C.c_int, [hid_t]),
        ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]),
        ("H5Dclose", C.c_int, [hid_t]),
        ("H5Dget_space", hid_t, [hid_t]),
        ("H5Dget_type", hid_t, [hid_t]),
        ("H5Tget_class", C.c_int, [hid_t]),
        ("H5Tget_size", C.c_size_t, [hid_t]),
        ("H5Tclose", C.c_int, [hid_t]),
        ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
        ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Sclose", C.c_int, [hid_t]),
        ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        ("H5Gclose", C.c_int, [hid_t]),
        ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]),
        ("H5Ldelete", C.c_int, [hid_t, C.c_char_p, hid_t]),
    ]:
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.H5open()
    _lib = L
    return L


def _native(dtype):
    L = lib()
    name = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g",
            np.dtype(np.int64): "H5T_NATIVE_INT64_g"}[np.dtype(dtype)]
    return hid_t.in_dll(L, name).value


def dataset_shape(path: str, name: str):
    """datasetShape (MyHDF5.chpl:36-69)."""
    L = lib()
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot open {path}")
    try:
        d = L.H5Dopen2(f, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        s = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(s)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(s, dims, None)
        L.H5Sclose(s)
        L.H5Dclose(d)
        return tuple(int(v) for v in dims[:nd])
    finally:
        L.H5Fclose(f)


def read_dataset(path: str, name: str) -> np.ndarray:
    """readDataset (MyHDF5.chpl:71-103): whole dataset as float64 or uint64."""
    L = lib()
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot open {path}")
    try:
        d = L.H5Dopen2(f, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        s = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(s)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(s, dims, None)
        L.H5Sclose(s)
        t = L.H5Dget_type(d)
        cls = L.H5Tget_class(t)
        L.H5Tclose(t)
        dtype = np.float64 if cls == H5T_FLOAT else np.uint64
        out = np.empty(tuple(int(v) for v in dims[:nd]), dtype=dtype)
        rc = L.H5Dread(d, _native(dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
        L.H5Dclose(d)
        if rc < 0:
            raise OSError(f"H5Dread failed for {name}")
        return out
    finally:
        L.H5Fclose(f)


def has_dataset(path: str, name: str) -> bool:
    """doesObjectExist (MyHDF5.chpl) for a file that may not exist"""
    import os

    if not os.path.exists(path):
        return False
    L = lib()
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        return False
    try:
        prefix = ""
        for g in [p for p in name.split("/") if p]:
            prefix += "/" + g
            if L.H5Lexists(f, prefix.encode(), H5P_DEFAULT) <= 0:
                return False
        return True
    finally:
        L.H5Fclose(f)


def write_datasets(path: str, datasets: dict, append: bool = False):
    """writes every {"/group/name": array}; intermediate groups are created (makeGroup + writeDataset,
    MyHDF5.chpl:288-333).  append=False creates / truncates `path`; append=True opens an existing file read-write
    and adds the datasets to it (the reference keeps basis/representatives in the output file and adds the
    hamiltonian group later, Diagonalize.chpl:227-256): an existing /basis/* dataset is left as it is (it is what the run
    was computed from; diagonalize() validates it against the configured basis before using it), every other existing
    dataset -- hamiltonian/* of an earlier run -- is deleted and replaced."""
    import os

    L = lib()
    if append and os.path.exists(path):
        f = L.H5Fopen(path.encode(), H5F_ACC_RDWR, H5P_DEFAULT)
    else:
        f = L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot create {path}")
    try:
        for name, arr in datasets.items():
            if append:
                comps = [p for p in name.split("/") if p]
                exists = all(L.H5Lexists(f, ("/" + "/".join(comps[:k + 1])).encode(), H5P_DEFAULT) > 0 for k in range(len(comps)))
                if exists and comps[0] == "basis":
                    continue  # the stored basis is what this run was computed from
                if exists and L.H5Ldelete(f, ("/" + "/".join(comps)).encode(), H5P_DEFAULT) < 0:
                    raise OSError(f"cannot replace {name}")
            arr = np.ascontiguousarray(arr)
            if arr.dtype not in (np.float64, np.uint64, np.int64):
                raise TypeError(f"{name}: only float64 / uint64 / int64 datasets are supported")
            parts = [p for p in name.split("/") if p]
            prefix = ""
            for g in parts[:-1]:
                prefix += "/" + g
                if L.H5Lexists(f, prefix.encode(), H5P_DEFAULT) <= 0:
                    L.H5Gclose(L.H5Gcreate2(f, prefix.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT))
            dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
            s = L.H5Screate_simple(arr.ndim, dims, None)
            d = L.H5Dcreate2(f, ("/" + "/".join(parts)).encode(), _native(arr.dtype), s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            rc = L.H5Dwrite(d, _native(arr.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p))
            L.H5Dclose(d)
            L.H5Sclose(s)
            if rc < 0:
                raise OSError(f"H5Dwrite failed for {name}")
    finally:
        L.H5Fclose(f)

"""Minimal HDF5 reader/writer for the reference's on-disk format either side of the matvec path
(/root/reference/src/MyHDF5.chpl:71-144,214-333; produced by /root/reference/input_for_matvec.py:43-46):
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
H5S_SELECT_SET = 0


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
        ("H5Sselect_hyperslab", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Tequal", C.c_int, [hid_t, hid_t]),
        ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]),
        ("H5Ldelete", C.c_int, [hid_t, C.c_char_p, hid_t]),
        ("H5Dget_offset", C.c_uint64, [hid_t]),
        ("H5Pcreate", hid_t, [hid_t]),
        ("H5Pclose", C.c_int, [hid_t]),
        ("H5Pset_alloc_time", C.c_int, [hid_t, C.c_int]),
        ("H5Pset_fill_time", C.c_int, [hid_t, C.c_int]),
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


# ---------------------------------------------------------------------------------------------
# Block-distributed I/O (MyHDF5.chpl:105-144, 214-253, 272-333): every locale reads / writes its own hyperslab of a
# dataset, so a vector that does not fit one host buffer (chain_40_symm eigenvectors: 6.9 GB each) never has to.  The blocks
# are Chapel's Block distribution of the LAST dimension over the locales (rank 1: the only one; rank 2: [batch, N] split along
# N, `reshape(Locales, {0 ..# 1, 0 ..# numLocales})`): locale p owns the indices i with floor(i * P / n) == p.
# ---------------------------------------------------------------------------------------------
def block_range(n: int, num_blocks: int, block: int):
    """[lo, hi) of the indices of 0..n-1 that Chapel's Block distribution over `num_blocks` locales gives locale `block`"""
    if not (0 <= block < num_blocks):
        raise ValueError(f"block {block} of {num_blocks}")
    lo = -((-block * n) // num_blocks)
    hi = -((-(block + 1) * n) // num_blocks)
    return lo, hi


def _open_checked(L, f, path, name, rank, dtype):
    d = L.H5Dopen2(f, name.encode(), H5P_DEFAULT)
    if d < 0:
        raise KeyError(name)
    s = L.H5Dget_space(d)
    nd = L.H5Sget_simple_extent_ndims(s)
    if nd != rank:  # MyHDF5.chpl:119-121 / 228-230
        L.H5Sclose(s)
        L.H5Dclose(d)
        raise ValueError(f"halt: rank mismatch in file: '{path}' dataset: '{name}'  {rank} != {nd}")
    t = L.H5Dget_type(d)
    same = L.H5Tequal(_native(dtype), t) > 0
    L.H5Tclose(t)
    if not same:  # MyHDF5.chpl:125-127 / 233-235
        L.H5Sclose(s)
        L.H5Dclose(d)
        raise TypeError(f"halt: type mismatch in file: '{path}' dataset: '{name}'  (expected {np.dtype(dtype).name})")
    return d, s


def _select(L, s, offset, shape):
    nd = len(shape)
    c_off = (hsize_t * nd)(*[int(v) for v in offset])
    c_shape = (hsize_t * nd)(*[int(v) for v in shape])
    dims = (hsize_t * nd)()
    L.H5Sget_simple_extent_dims(s, dims, None)
    if any(int(c_off[k]) + int(c_shape[k]) > int(dims[k]) for k in range(nd)):
        raise IndexError(f"hyperslab {tuple(offset)} + {tuple(shape)} exceeds the dataset {tuple(int(v) for v in dims)}")
    if L.H5Sselect_hyperslab(s, H5S_SELECT_SET, c_off, None, c_shape, None) < 0:
        raise OSError("H5Sselect_hyperslab failed")
    return L.H5Screate_simple(nd, c_shape, None)


def read_dataset_chunk(path: str, name: str, offset, shape, dtype=np.float64) -> np.ndarray:
    """readDatasetChunk (MyHDF5.chpl:105-159): the hyperslab `offset` + `shape` of a dataset; the rank and the element type
    must be the file's (the reference halts on either mismatch)."""
    L = lib()
    shape = tuple(int(v) for v in shape)
    if len(offset) != len(shape):
        raise ValueError("offset and shape differ in rank")
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot open {path}")
    try:
        d, s = _open_checked(L, f, path, name, len(shape), dtype)
        try:
            out = np.empty(shape, dtype=dtype)
            if out.size == 0:
                return out
            m = _select(L, s, offset, shape)
            rc = L.H5Dread(d, _native(dtype), m, s, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
            L.H5Sclose(m)
            if rc < 0:
                raise OSError(f"H5Dread failed for {name}")
            return out
        finally:
            L.H5Sclose(s)
            L.H5Dclose(d)
    finally:
        L.H5Fclose(f)


def create_dataset(path: str, name: str, shape, dtype=np.float64, allocate: bool = False):
    """the first half of writeDatasetAsBlocks (MyHDF5.chpl:303-322): an empty dataset of the full shape (an existing one of
    that name is replaced; the file and the intermediate groups are created when missing).

    allocate=True: the (contiguous) storage is allocated in the file right away and its byte address is returned
    (None when the library does not give one), so that the hyperslab writers of all ranks can go to the file side by side
    with write_hyperslab_raw -- no second process ever touches the HDF5 metadata."""
    L = lib()
    address = None
    f = L.H5Fopen(path.encode(), H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(path) else \
        L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot open {path}")
    try:
        parts = [p for p in name.split("/") if p]
        prefix = ""
        for g in parts[:-1]:
            prefix += "/" + g
            if L.H5Lexists(f, prefix.encode(), H5P_DEFAULT) <= 0:
                L.H5Gclose(L.H5Gcreate2(f, prefix.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT))
        full = "/" + "/".join(parts)
        if L.H5Lexists(f, full.encode(), H5P_DEFAULT) > 0 and L.H5Ldelete(f, full.encode(), H5P_DEFAULT) < 0:
            raise OSError(f"cannot replace {name}")
        shape = tuple(int(v) for v in shape)
        dims = (hsize_t * max(len(shape), 1))(*shape)
        s = L.H5Screate_simple(len(shape), dims, None)
        dcpl = H5P_DEFAULT
        if allocate and all(shape):
            dcpl = L.H5Pcreate(hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value)
            # H5D_ALLOC_TIME_EARLY = 1: space at creation; H5D_FILL_TIME_NEVER = 1: every byte is written by its owner anyway
            if dcpl < 0 or L.H5Pset_alloc_time(dcpl, 1) < 0 or L.H5Pset_fill_time(dcpl, 1) < 0:
                raise OSError("cannot set up the dataset creation properties")
        d = L.H5Dcreate2(f, full.encode(), _native(dtype), s, H5P_DEFAULT, dcpl, H5P_DEFAULT)
        L.H5Sclose(s)
        if dcpl != H5P_DEFAULT:
            L.H5Pclose(dcpl)
        if d < 0:
            raise OSError(f"cannot create {name}")
        if allocate and all(shape):
            a = int(L.H5Dget_offset(d))
            address = None if a == 0xFFFFFFFFFFFFFFFF else a  # HADDR_UNDEF: not contiguous / not allocated
        L.H5Dclose(d)
    finally:
        L.H5Fclose(f)  # flushes the metadata and extends the file to the end of the allocation
    if address is not None and os.path.getsize(path) < address + int(np.prod(shape)) * np.dtype(dtype).itemsize:
        address = None  # the file does not cover the allocation (a library that defers it after all): stay with H5Dwrite
    return address


def write_hyperslab_raw(path: str, address: int, shape, offset, arr):
    """the second half of writeDatasetAsBlocks for a dataset made by create_dataset(allocate=True): `arr` goes straight to the
    bytes of its hyperslab (row-major contiguous storage at `address`; one pwrite per contiguous run), so the writers of all
    ranks run at the same time on disjoint byte ranges -- what the reference's `coforall loc in Locales` (MyHDF5.chpl:328-332)
    does through one HDF5 handle per locale.  Native byte order, like H5T_NATIVE_* on this platform."""
    arr = np.ascontiguousarray(arr)
    shape, offset = tuple(int(v) for v in shape), tuple(int(v) for v in offset)
    if len(offset) != arr.ndim or len(shape) != arr.ndim:
        raise ValueError("offset, dataset and array differ in rank")
    if any(o < 0 or o + e > n for o, e, n in zip(offset, arr.shape, shape)):
        raise IndexError(f"hyperslab {offset} + {arr.shape} exceeds the dataset {shape}")
    if arr.size == 0:
        return
    item = arr.dtype.itemsize
    strides = [item] * arr.ndim
    for k in range(arr.ndim - 2, -1, -1):
        strides[k] = strides[k + 1] * shape[k + 1]
    # the trailing dimensions the hyperslab covers completely are contiguous in the file together with the first partial one
    run_dim = arr.ndim - 1
    while run_dim > 0 and arr.shape[run_dim] == shape[run_dim]:
        run_dim -= 1
    run_bytes = int(np.prod(arr.shape[run_dim:])) * item
    flat = arr.reshape(-1).view(np.uint8)
    fd = os.open(path, os.O_WRONLY)
    try:
        for idx, lead in enumerate(np.ndindex(*arr.shape[:run_dim])):
            pos = address + sum((offset[k] + lead[k]) * strides[k] for k in range(run_dim)) + offset[run_dim] * strides[run_dim]
            buf = memoryview(flat[idx * run_bytes:(idx + 1) * run_bytes])
            done = 0
            while done < run_bytes:
                done += os.pwrite(fd, buf[done:], pos + done)
    finally:
        os.close(fd)


def write_dataset_chunk(path: str, name: str, offset, arr):
    """writeDatasetChunk (MyHDF5.chpl:214-253): `arr` into the hyperslab at `offset` of an existing dataset of the same rank
    and element type."""
    L = lib()
    arr = np.ascontiguousarray(arr)
    if len(offset) != arr.ndim:
        raise ValueError("offset and array differ in rank")
    f = L.H5Fopen(path.encode(), H5F_ACC_RDWR, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"cannot open {path}")
    try:
        d, s = _open_checked(L, f, path, name, arr.ndim, arr.dtype)
        try:
            if arr.size == 0:
                return
            m = _select(L, s, offset, arr.shape)
            rc = L.H5Dwrite(d, _native(arr.dtype), m, s, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p))
            L.H5Sclose(m)
            if rc < 0:
                raise OSError(f"halt: HDF5 error: could not write array to dataset {name}")
        finally:
            L.H5Sclose(s)
            L.H5Dclose(d)
    finally:
        L.H5Fclose(f)


def read_dataset_block(path: str, name: str, num_blocks: int, block: int, dtype=np.float64) -> np.ndarray:
    """what one locale does inside readDatasetAsBlocks (MyHDF5.chpl:272-287): its block of the last dimension"""
    shape = dataset_shape(path, name)
    lo, hi = block_range(shape[-1], num_blocks, block)
    offset = (0,) * (len(shape) - 1) + (lo,)
    return read_dataset_chunk(path, name, offset, shape[:-1] + (hi - lo,), dtype)


def read_dataset_as_blocks(path: str, name: str, num_blocks: int, dtype=np.float64):
    """readDatasetAsBlocks: the blocks of all locales, in locale order (one process standing in for every locale)"""
    return [read_dataset_block(path, name, num_blocks, b, dtype) for b in range(num_blocks)]


def write_dataset_as_blocks(path: str, name: str, blocks):
    """writeDatasetAsBlocks (MyHDF5.chpl:303-333): create the dataset at its full shape, then every locale's block goes to its
    hyperslab of the last dimension.  `blocks[p]` must have the extent block_range gives locale p."""
    blocks = [np.ascontiguousarray(b) for b in blocks]
    n = sum(b.shape[-1] for b in blocks)
    lead = blocks[0].shape[:-1]
    create_dataset(path, name, lead + (n,), blocks[0].dtype)
    for p, b in enumerate(blocks):
        lo, hi = block_range(n, len(blocks), p)
        if b.shape[:-1] != lead or b.shape[-1] != hi - lo:
            raise ValueError(f"block {p} has shape {b.shape}; the Block distribution of {n} over {len(blocks)} gives it {hi - lo}")
        write_dataset_chunk(path, name, (0,) * len(lead) + (lo,), b)

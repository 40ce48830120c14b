"""world_size-2/3 exchange logic of distributed-matvec_amd/distributed.py on CPU (gloo).

The HIP engine cannot run here, so an oracle-backed engine with the same interface is injected:
what is under test is the host logic of the N>1 path -- round agreement, the counts exchange, exact
all_to_all_single splits, segment layout, scatter of every received segment -- against the
single-locale oracle result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in for HipEngine (TEST ONLY): same interface, oracle arithmetic."""

    def __init__(self, matrix, representatives, dtype, num_partitions, my_partition, num_rounds):
        from oracle import c_oracle as CO

        self.o = matrix  # a COracle
        self.m = matrix.model
        self.reps = np.ascontiguousarray(representatives.numpy().view(np.uint64))
        self.P, self.me, self.num_rounds = num_partitions, my_partition, num_rounds
        self.cplx = dtype == torch.complex128
        self.packet_bytes = 24 if self.cplx else 16
        self.CO = CO
        n = len(self.reps)
        self.bounds = [(n * r // num_rounds, n * (r + 1) // num_rounds) for r in range(num_rounds)]
        self.norms = self.o.state_info(self.reps)[2] if self.m.has_permutations else None

    def _expand(self, rnd, x):
        lo, hi = self.bounds[rnd]
        rows = self.reps[lo:hi]
        betas, cs, offs = self.o.apply_off_diag(rows)
        row_of = np.repeat(np.arange(lo, hi), np.diff(offs))
        vals = cs * (x[row_of] if x is not None else 1.0)
        if self.m.has_permutations:
            reps, chars, norms = self.o.state_info(betas)
            vals = vals * chars * norms / self.norms[row_of]
            keep = norms > 0
            betas, vals = reps[keep], vals[keep]
        elif self.m.spin_inversion != 0:
            flipped = betas ^ np.uint64(self.m.mask)
            sw = flipped < betas
            betas = np.where(sw, flipped, betas)
            vals = np.where(sw, vals * self.m.spin_inversion, vals)
        keys = self.CO.locale_idx_of(betas, self.P)
        return betas, vals, keys

    def send_counts(self, rnd):
        _, _, keys = self._expand(rnd, None)
        c = np.bincount(keys, minlength=self.P)
        c[self.me] = 0
        return [int(v) for v in c]

    def alloc_bytes(self, n):
        return torch.zeros(max(int(n), 8), dtype=torch.uint8)

    def diag(self, x, y):
        d = self.o.apply_diag(self.reps)
        if self.m.diag.v.size:
            y.copy_(torch.from_numpy(d * x.numpy()))

    def _add(self, y, betas, vals):
        idx = self.CO.state_index(self.reps, betas)
        assert (idx >= 0).all()
        yn = y.numpy()
        np.add.at(yn, idx, vals if self.cplx else vals.real)

    def generate(self, rnd, x, y, send):
        betas, vals, keys = self._expand(rnd, x.numpy())
        self._add(y, betas[keys == self.me], vals[keys == self.me])
        buf = send.numpy()
        off = 0
        for d in range(self.P):
            if d == self.me:
                continue
            sel = keys == d
            n = int(sel.sum())
            buf[off:off + 8 * n] = betas[sel].view(np.uint8)
            v = vals[sel] if self.cplx else np.ascontiguousarray(vals[sel].real)
            w = 16 if self.cplx else 8
            buf[off + 8 * n:off + (8 + w) * n] = v.view(np.uint8)
            off += (8 + w) * n

    def scatter(self, recv, byte_offset, n, y):
        buf = recv.numpy()
        betas = buf[byte_offset:byte_offset + 8 * n].view(np.uint64)
        w = 16 if self.cplx else 8
        vals = buf[byte_offset + 8 * n:byte_offset + (8 + w) * n].view(np.complex128 if self.cplx else np.float64)
        self._add(y, betas.copy(), vals.copy())

    def check(self):
        pass


class OracleEngineIndexed(OracleEngine):
    """... with the PRE-INDEXED packet layout of the HIP engine (include/ls_amd.h "Packet layout"): a segment of c packets is
    [u32 index at the destination x c, padded to 8 bytes][value x c], and all segments of a round are consumed by ONE call"""

    all_reps = None  # the whole basis (ascending): the producer ranks a state inside its owner's block, as the rank directory does
    key_bytes = 4

    def with_state_keys(self):
        """what HipEngine.with_state_keys does: the same engine with state-carrying packets"""
        return OracleEngine(self.o, torch.from_numpy(self.reps.view(np.int64).copy()), torch.complex128 if self.cplx else torch.float64,
                            self.P, self.me, self.num_rounds)

    def segment_bytes(self, c):
        return ((4 * c + 7) & ~7) + (16 if self.cplx else 8) * c

    def generate(self, rnd, x, y, send):
        betas, vals, keys = self._expand(rnd, x.numpy())
        self._add(y, betas[keys == self.me], vals[keys == self.me])
        buf = send.numpy()
        owners = self.CO.locale_idx_of(self.all_reps, self.P)
        off = 0
        for d in range(self.P):
            if d == self.me:
                continue
            sel = keys == d
            n = int(sel.sum())
            idx = self.CO.state_index(np.ascontiguousarray(self.all_reps[owners == d]), betas[sel]).astype(np.uint32)
            kb = (4 * n + 7) & ~7
            buf[off:off + 4 * n] = idx.view(np.uint8)
            v = vals[sel] if self.cplx else np.ascontiguousarray(vals[sel].real)
            w = 16 if self.cplx else 8
            buf[off + kb:off + kb + w * n] = v.view(np.uint8)
            off += kb + w * n

    def scatter(self, recv, byte_offset, n, y):
        raise AssertionError("the operator must consume a round through scatter_round when the engine offers it")

    def scatter_round(self, recv, counts, offsets, y):
        buf = recv.numpy()
        w = 16 if self.cplx else 8
        yn = y.numpy()
        for n, off in zip(counts, offsets):
            if n == 0:
                continue
            kb = (4 * n + 7) & ~7
            idx = buf[off:off + 4 * n].view(np.uint32).astype(np.int64)
            vals = buf[off + kb:off + kb + w * n].view(np.complex128 if self.cplx else np.float64)
            np.add.at(yn, idx, vals)


def _worker(rank, world, port, name, cplx, num_rounds, out_dir, indexed=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import model_config
    from oracle import c_oracle as CO
    from oracle import model as M

    from distributed_matvec_amd.distributed import DistributedOperator

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        o = CO.COracle(M.model_from_config(model_config(name)))
        reps = o.enumerate()
        rs = np.random.RandomState(5)
        x = rs.rand(len(reps)) - 0.5
        if cplx:
            x = x + 1j * (rs.rand(len(reps)) - 0.5)
        keys = CO.locale_idx_of(reps, world)
        mine = keys == rank
        my_reps = torch.from_numpy(reps[mine].view(np.int64).copy())
        my_x = torch.from_numpy(x[mine].copy())
        my_y = torch.full_like(my_x, 7.0)  # overwritten by the diagonal pass
        OracleEngineIndexed.all_reps = reps
        # indexed == "mixed": the ranks DISAGREE -- rank 0's plan comes back with pre-indexed packets, the others' with state-carrying
        # ones (uneven free HBM, a failed self-check).  The operator must settle on ONE layout: all state-carrying.
        factory = OracleEngine if not indexed or (indexed == "mixed" and rank != 0) else OracleEngineIndexed
        op = DistributedOperator(o, my_reps, my_x.dtype, engine_factory=factory, num_rounds=num_rounds)
        assert op.num_rounds == (num_rounds if num_rounds else 1)
        if indexed == "mixed":
            assert op.key_bytes == 8 and type(op.engine) is OracleEngine
        elif indexed:
            assert op.key_bytes == 4
        if indexed is True:  # 12 instead of 16 bytes per f64 packet on the wire (keys padded to 8 bytes per segment)
            packets = sum(sum(c) for c in op.send_counts)
            assert op.exchange_bytes_per_matvec <= packets * ((20 if cplx else 12)) + 4 * world * op.num_rounds
        # send/recv count matrices are transposes of each other across ranks
        gathered = [None] * world
        dist.all_gather_object(gathered, (op.send_counts, op.recv_counts))
        for r in range(op.num_rounds):
            for a in range(world):
                for b in range(world):
                    assert gathered[a][0][r][b] == gathered[b][1][r][a]
        op.matvec(my_x, my_y, check=True)
        # reductions used by the eigensolver callbacks
        nrm2 = op.dot(my_x, my_x)
        assert abs(complex(nrm2) - np.vdot(x, x)) < 1e-9
        np.save(os.path.join(out_dir, f"y{rank}.npy"), my_y.numpy())
        # the self-verification of `bench.py --gpus N` (distributed-matvec_amd/verify.py) across REAL processes: every rank's block
        # against its rows of the one-partition result, element-wise + all-reduced invariants; then a fault on ONE rank, which
        # every rank must learn about
        from distributed_matvec_amd import verify

        def allsum(v):
            t = torch.tensor([float(v)], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())

        def allmax(v):
            t = torch.tensor([float(v)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        y_ref = torch.from_numpy(o.local_matvec(reps, x)[mine].copy())
        ymax = float(np.abs(o.local_matvec(reps, x)).max())
        par = verify.parity_object(my_y, my_x, y_ref, ymax, allsum=allsum, allmax=allmax, reference_kernel="oracle")
        assert par["ok"] and par["max_rel_err"] <= 1e-12 and par["rows_off"] == 0, par
        bad = my_y.clone()
        if rank == 0:
            bad[:-1] = my_y[1:]  # rank 0's rows arrive one element late
        par = verify.parity_object(bad, my_x, y_ref, ymax, allsum=allsum, allmax=allmax, reference_kernel="oracle")
        assert not par["ok"] and par["rows_off"] > 0, (rank, par)
        # a block of the wrong SIZE on one rank: every rank gets the verdict (and none is left waiting in a reduction)
        short = my_y[:-1] if rank == world - 1 else my_y
        par = verify.parity_object(short, my_x, y_ref, ymax, allsum=allsum, allmax=allmax, reference_kernel="oracle")
        assert not par["ok"] and "block sizes differ" in par["error"], (rank, par)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,cplx,rounds,indexed", [
    ("heisenberg_chain_12", 2, False, 1, False),
    ("heisenberg_chain_10", 2, False, 3, False),
    ("heisenberg_kagome_12_symm", 2, True, 2, False),
    ("heisenberg_chain_16", 3, False, 2, False),
    ("heisenberg_chain_16", 3, False, 2, True),
    ("heisenberg_kagome_12", 2, True, 3, True),
    ("heisenberg_chain_16", 3, False, 2, "mixed"),
    ("heisenberg_kagome_12", 2, True, 1, "mixed"),
])
def test_all_to_all_exchange(tmp_path, name, world, cplx, rounds, indexed):
    sys.path.insert(0, ROOT)
    from helpers import oracle_for, oracle_reps
    from oracle import c_oracle as CO

    port = 29600 + (os.getpid() % 200) + world * 7 + rounds
    mp.spawn(_worker, args=(world, port + (13 if indexed is True else 29 if indexed else 0), name, cplx, rounds, str(tmp_path), indexed), nprocs=world, join=True)
    reps = oracle_reps(name)
    rs = np.random.RandomState(5)
    x = rs.rand(len(reps)) - 0.5
    if cplx:
        x = x + 1j * (rs.rand(len(reps)) - 0.5)
    want = oracle_for(name).local_matvec(reps, x)
    keys = CO.locale_idx_of(reps, world)
    parts = [np.load(os.path.join(str(tmp_path), f"y{r}.npy")) for r in range(world)]
    got = CO.hashed_to_block(parts, keys)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


class OracleReplicatedEngine:
    """CPU stand-in for HipReplicatedEngine (TEST ONLY): oracle arithmetic on this rank's rows."""

    def __init__(self, matrix, reps_local, reps_global, dtype, num_partitions, my_partition):
        self.o = matrix
        self.reps_global = np.ascontiguousarray(reps_global.numpy().view(np.uint64))
        self.reps_local = np.ascontiguousarray(reps_local.numpy().view(np.uint64))
        from oracle import c_oracle as CO

        self.rows = CO.state_index(self.reps_global, self.reps_local)

    def matvec(self, x_global, y_local):
        full = self.o.local_matvec(self.reps_global, x_global.numpy().copy())
        y_local.copy_(torch.from_numpy(full[self.rows]))

    def check(self):
        pass


def _worker_replicated(rank, world, port, name, cplx, out_dir, xgather="p2p"):
    os.environ["LS_AMD_XGATHER"] = xgather
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import model_config
    from oracle import c_oracle as CO
    from oracle import model as M

    from distributed_matvec_amd.distributed import ReplicatedOperator

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        o = CO.COracle(M.model_from_config(model_config(name)))
        reps = o.enumerate()
        rs = np.random.RandomState(6)
        x = rs.rand(len(reps)) - 0.5
        if cplx:
            x = x + 1j * (rs.rand(len(reps)) - 0.5)
        keys = CO.locale_idx_of(reps, world)
        mine = keys == rank
        op = ReplicatedOperator(o, torch.from_numpy(reps[mine].view(np.int64).copy()), torch.from_numpy(reps.view(np.int64).copy()),
                                torch.from_numpy(keys.copy()), torch.complex128 if cplx else torch.float64,
                                engine_factory=OracleReplicatedEngine)
        my_x = torch.from_numpy(x[mine].copy())
        xg = op.gather_x(my_x)
        assert np.array_equal(xg.numpy(), x)  # hashed blocks -> global ascending order, bit-exact
        my_y = torch.zeros_like(my_x)
        op.matvec(my_x, my_y, check=True)
        np.save(os.path.join(out_dir, f"y{rank}.npy"), my_y.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,cplx,xgather", [("heisenberg_chain_12", 2, False, "p2p"),
                                                     ("heisenberg_kagome_12_symm", 3, True, "p2p"),
                                                     ("heisenberg_chain_10", 3, False, "allgather")])
def test_replicated_x_exchange(tmp_path, name, world, cplx, xgather):
    sys.path.insert(0, ROOT)
    from helpers import oracle_for, oracle_reps
    from oracle import c_oracle as CO

    port = 29900 + (os.getpid() % 90) + world + (11 if xgather == "allgather" else 0)
    mp.spawn(_worker_replicated, args=(world, port, name, cplx, str(tmp_path), xgather), nprocs=world, join=True)
    reps = oracle_reps(name)
    rs = np.random.RandomState(6)
    x = rs.rand(len(reps)) - 0.5
    if cplx:
        x = x + 1j * (rs.rand(len(reps)) - 0.5)
    want = oracle_for(name).local_matvec(reps, x)
    keys = CO.locale_idx_of(reps, world)
    got = CO.hashed_to_block([np.load(os.path.join(str(tmp_path), f"y{r}.npy")) for r in range(world)], keys)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def _worker_layouts(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_reps
    from oracle import c_oracle as CO

    from distributed_matvec_amd import hdf5
    from distributed_matvec_amd.distributed import block_to_hashed, hashed_to_block, read_hashed_vector, write_hashed_vectors

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        reps = oracle_reps(name)
        n = len(reps)
        keys = CO.locale_idx_of(reps, world)
        masks = torch.from_numpy(keys.astype(np.uint8))
        rs = np.random.RandomState(9)
        xs = [rs.rand(n) - 0.5 for _ in range(2)]
        mine = keys == rank
        parts = [torch.from_numpy(x[mine].copy()) for x in xs]
        lo, hi = hdf5.block_range(n, world, rank)
        # hashed -> block -> hashed (HashedToBlock.chpl / BlockToHashed.chpl across processes), 8-byte payloads of both kinds
        blk = hashed_to_block(parts[0], masks)
        assert blk.shape == (hi - lo,) and np.array_equal(blk.numpy(), xs[0][lo:hi])
        assert np.array_equal(block_to_hashed(blk, masks).numpy(), xs[0][mine])
        rblk = hashed_to_block(torch.from_numpy(reps[mine].view(np.int64).copy()), masks)
        assert np.array_equal(rblk.numpy().view(np.uint64), reps[lo:hi])
        with pytest.raises(ValueError):
            hashed_to_block(parts[0][:-1], masks)
        # every rank writes its own hyperslab of [k, N]; then reads its block back into the hash partition
        try:
            hdf5.lib()
        except hdf5.Hdf5Unavailable:
            return
        path = os.path.join(out_dir, "vectors.h5")
        # all ranks write at once, straight to the bytes of their hyperslabs (write_block_dataset / hdf5.write_hyperslab_raw) ...
        assert write_hashed_vectors(path, "/hamiltonian/eigenvectors", parts, masks) is True
        dist.barrier()
        if rank == 0:
            assert np.array_equal(hdf5.read_dataset(path, "/hamiltonian/eigenvectors"), np.stack(xs))
        for row in range(2):
            assert np.array_equal(read_hashed_vector(path, "/hamiltonian/eigenvectors", row, masks).numpy(), xs[row][mine])
        # ... a rank-1 dataset next to it in the same file (basis/representatives of diagonalize_distributed) leaves it intact ...
        from distributed_matvec_amd.distributed import write_block_dataset

        assert write_block_dataset(path, "/basis/representatives", (n,), np.uint64, [reps[lo:hi]]) is True
        assert np.array_equal(hdf5.read_dataset(path, "/basis/representatives"), reps)
        assert np.array_equal(hdf5.read_dataset_block(path, "/basis/representatives", world, rank, np.uint64), reps[lo:hi])
        assert np.array_equal(hdf5.read_dataset(path, "/hamiltonian/eigenvectors"), np.stack(xs))
        with pytest.raises(ValueError):  # a block of the wrong length is refused before anything is written
            write_block_dataset(path, "/basis/wrong", (n,), np.uint64, [np.zeros(hi - lo + 1, dtype=np.uint64)])
        dist.barrier()
        # rank 0 cannot create the dataset (no such directory): EVERY rank raises, nobody is left waiting in the broadcast
        with pytest.raises(OSError):
            write_block_dataset(os.path.join(out_dir, "no-such-directory", "v.h5"), "/basis/representatives", (n,), np.uint64, [reps[lo:hi]])
        dist.barrier()
        # ... and the one-H5Dwrite-at-a-time fallback (a library that gives no storage address) writes the same file
        os.environ["LS_AMD_HDF5_RAW"] = "0"
        try:
            path2 = os.path.join(out_dir, "vectors_serial.h5")
            assert write_hashed_vectors(path2, "/hamiltonian/eigenvectors", parts, masks) is False
            assert np.array_equal(hdf5.read_dataset(path2, "/hamiltonian/eigenvectors"), np.stack(xs))
        finally:
            del os.environ["LS_AMD_HDF5_RAW"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("heisenberg_chain_12", 2), ("heisenberg_kagome_12_symm", 3), ("heisenberg_chain_4", 3)])
def test_layout_converters_and_block_io_across_processes(tmp_path, name, world):
    """hashed <-> block conversion as one all-to-all-v with counts derived from the masks alone, and per-rank hyperslab I/O
    of hash-partitioned vectors (Diagonalize.chpl:248-256, MyHDF5.chpl:272-333) -- world_size 2 and 3, including a basis with
    fewer states than ranks can split evenly"""
    port = 29850 + (os.getpid() % 100) + world
    mp.spawn(_worker_layouts, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)

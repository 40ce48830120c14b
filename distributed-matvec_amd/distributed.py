"""One process per GPU: hash-partitioned y <- H x.

Two exchange strategies (both keep x, y and the representatives hash-partitioned at the interface):
  * DistributedOperator  -- all-to-all-v of (sigma_j, c_j x_i) packets, the reference's formulation;
  * ReplicatedOperator   -- Hermitian operators: all-gather x (N w bytes instead of nnz (8 + w) bytes,
                            ~ 30x fewer on the chains) and pull locally with no atomics.


Replaces the reference's locale-to-locale machinery
(/root/reference/src/DistributedMatrixVector.chpl:313-853: _LocalBuffer/_RemoteBuffer mailboxes,
Producer/Consumer tasks, one-sided PUTs + fast-on flag hand-shakes, 3 barriers per matvec) by
bulk-synchronous rounds on each rank:

    generate(round)      HIP: rows -> terms -> projection -> hash64_01 % P -> per-destination segments
    all_to_all_single    RCCL over xGMI (backend "nccl"); exact split sizes are known from the plan's
                         count pass, so no per-round size exchange and no host sync
    scatter(segment)     HIP: local index lookup + 64-bit atomic add

Packets owned by the generating rank never leave the GPU (scattered inside `generate`).  The engine
is pluggable so that the exchange logic can be exercised on CPU with the gloo backend (tests inject
an oracle-backed engine); the product engine is HipEngine and nothing else ships.
"""
from __future__ import annotations

import os


class _Transport:
    """torch.distributed calls used by the operators.  With the "nccl" backend (RCCL) device tensors are
    handed over as they are; with any other backend (gloo in the 2-process tests) device tensors are
    staged through host memory -- a transport detail only, the compute stays on the GPU."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.direct = dist.get_backend(group) == "nccl"

    class _Done:
        def __init__(self, fn=None):
            self.fn = fn

        def wait(self):
            if self.fn:
                self.fn()

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, async_op=False):
        dist = self.dist
        if self.direct or not out.is_cuda:
            work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)
            return work if async_op else None
        inp_h = inp.cpu()
        out_h = self.torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(out_h, inp_h, out_splits, in_splits, group=self.group)
        out.copy_(out_h)
        return self._Done() if async_op else None

    def exchange_blocks(self, send, outs, rank, P):
        """every rank's `send` lands in outs[rank] of every peer (outs[rank] itself is written locally).
        Grouped point-to-point by default; LS_AMD_XGATHER=allgather (or a failing P2P batch) switches to
        all_gather on blocks padded to the largest one."""
        dist = self.dist
        outs[rank].copy_(send)
        if P == 1:
            return
        if getattr(self, "use_allgather", None) is None:
            self.use_allgather = os.environ.get("LS_AMD_XGATHER", "p2p") == "allgather"
        if not self.use_allgather:
            try:
                return self._exchange_p2p(send, outs, rank, P)
            except RuntimeError as e:  # e.g. a backend without grouped send/recv
                import warnings

                warnings.warn(f"grouped send/recv failed ({e!r}); falling back to all_gather")
                self.use_allgather = True
        torch = self.torch
        mc = max(int(o.numel()) for o in outs)
        stage = not self.direct and send.is_cuda
        dev = "cpu" if stage else send.device
        padded = torch.zeros(mc, dtype=send.dtype, device=dev)
        padded[: send.numel()].copy_(send)
        bufs = [torch.empty(mc, dtype=send.dtype, device=dev) for _ in range(P)]
        dist.all_gather(bufs, padded, group=self.group)
        for p in range(P):
            if p != rank:
                outs[p].copy_(bufs[p][: outs[p].numel()])

    def _exchange_p2p(self, send, outs, rank, P):
        dist = self.dist
        stage = not self.direct and send.is_cuda
        src_t = send.cpu() if stage else send
        bufs = {}
        ops = []
        for step in range(1, P):
            dst, src = (rank + step) % P, (rank - step) % P
            bufs[src] = self.torch.empty(outs[src].shape, dtype=outs[src].dtype) if stage else outs[src]
            ops.append(dist.P2POp(dist.isend, src_t, dst, self.group))
            ops.append(dist.P2POp(dist.irecv, bufs[src], src, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if stage:
            for src, b in bufs.items():
                outs[src].copy_(b)


class HipEngine:
    """ls_amd plan that owns exactly one partition (this rank's)."""

    def __init__(self, matrix, representatives, dtype, num_partitions: int, my_partition: int, num_rounds: int,
                 state_keys: bool = False):
        import torch

        from . import _lib
        from .api import MatvecPlan

        self.torch = torch
        self._args = (matrix, representatives, dtype, num_partitions, my_partition, num_rounds)
        if state_keys:  # the ranks agreed on state-carrying packets (DistributedOperator.__init__): no all-destinations directory
            _lib.load().ls_amd_internal_set_no_packet_index(1)
        try:
            self.plan = MatvecPlan(matrix, representatives, dtype, my_partition=my_partition,
                                   num_partitions=num_partitions, num_rounds=num_rounds)
        finally:
            if state_keys:
                _lib.load().ls_amd_internal_set_no_packet_index(0)
        self.device = representatives.device
        self.num_rounds = self.plan.num_rounds
        self.packet_bytes = self.plan.packet_bytes

    @property
    def key_bytes(self):
        """4: pre-indexed packets (u32 index at the destination), 8: the packets carry the state"""
        return self.plan.key_bytes

    def with_state_keys(self):
        """the same engine with state-carrying packets (the layout every rank can always produce); this one is destroyed"""
        self.plan.destroy()
        return HipEngine(*self._args, state_keys=True)

    def segment_bytes(self, count):
        """bytes of a segment of `count` packets (keys padded to 8 bytes, then values; include/ls_amd.h)"""
        return self.plan.segment_bytes(count)

    def scatter_round(self, recv, counts, offsets, y):
        self.plan.scatter_round(counts, offsets, recv.data_ptr(), y)

    def send_counts(self, rnd):
        return self.plan.send_counts(rnd)

    def alloc_bytes(self, n):
        return self.torch.empty(max(int(n), 8), dtype=self.torch.uint8, device=self.device)

    def diag(self, x, y):
        self.plan.diag(x, y)

    def generate(self, rnd, x, y, send):
        self.plan.generate(rnd, x, y, send)

    def scatter(self, recv, byte_offset, n, y):
        base = recv.data_ptr() + byte_offset
        self.plan.scatter(n, base, base + self.plan.segment_value_offset(n), y)

    def check(self):
        self.plan.check()


def _rows_per_round():
    v = os.environ.get("LS_AMD_ROWS_PER_ROUND")
    return int(v) if v else 1 << 24


class DistributedOperator:
    """matrixVectorProduct with one locale per process (rank == locale index == hash64_01 % P)."""

    def __init__(self, matrix, representatives, dtype, group=None, engine_factory=None, num_rounds=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.P = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.meta_device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        n_local = int(representatives.numel()) if hasattr(representatives, "numel") else len(representatives)
        # every rank must run the same number of rounds (the collectives are matched)
        counts = torch.tensor([n_local], dtype=torch.int64, device=self.meta_device)
        all_counts = [torch.zeros_like(counts) for _ in range(self.P)]
        dist.all_gather(all_counts, counts, group=group)
        self.counts = [int(c.item()) for c in all_counts]
        if num_rounds is None:
            rpr = _rows_per_round()
            num_rounds = max(1, max((c + rpr - 1) // rpr for c in self.counts))
        self.num_rounds = num_rounds
        factory = engine_factory or HipEngine
        self.engine = factory(matrix, representatives, dtype, self.P, self.rank, num_rounds)
        assert self.engine.num_rounds == num_rounds
        # ONE packet layout for all ranks (csrc/dist.c does the same for the C host's driver): a plan decides alone whether
        # it writes pre-indexed 12-byte packets (the all-destinations directory must fit a quarter of ITS free HBM and pass its
        # self-check) or state-carrying 16-byte ones -- ranks that disagreed would read each other's u64 states as u32 indices.
        # All-reduce the key width with MAX; a rank that is better off than the agreement rebuilds its plan with state keys.
        kb = torch.tensor([int(getattr(self.engine, "key_bytes", 8))], dtype=torch.int64, device=self.meta_device)
        dist.all_reduce(kb, op=dist.ReduceOp.MAX, group=group)
        self.key_bytes = int(kb.item())
        if int(getattr(self.engine, "key_bytes", 8)) != self.key_bytes:
            self.engine = self.engine.with_state_keys()
            assert self.engine.num_rounds == num_rounds and int(getattr(self.engine, "key_bytes", 8)) == self.key_bytes
        pb = self.engine.packet_bytes
        # counts matrix exchange, once: S[d, r] = packets this rank sends to d in round r
        S = torch.tensor([self.engine.send_counts(r) for r in range(num_rounds)], dtype=torch.int64).t().contiguous()
        R = torch.zeros_like(S)
        self.transport = _Transport(group)
        S_dev, R_dev = S.to(self.meta_device), R.to(self.meta_device)
        dist.all_to_all_single(R_dev, S_dev, group=group)
        self.send_counts = S.t().tolist()             # [round][dest]
        self.recv_counts = R_dev.cpu().t().tolist()   # [round][src]
        self.packet_bytes = pb
        # bytes of a segment of c packets: the engine's layout (pre-indexed packets pad their u32 keys to 8 bytes), else c * pb
        seg = getattr(self.engine, "segment_bytes", None) or (lambda c: c * pb)
        self.send_bytes = [[seg(c) for c in row] for row in self.send_counts]
        self.recv_bytes = [[seg(c) for c in row] for row in self.recv_counts]
        max_send = max(sum(b) for b in self.send_bytes)
        max_recv = max(sum(b) for b in self.recv_bytes)
        # double-buffered so that generate(r + 1) can be queued while exchange(r) is in flight
        self.send_bufs = [self.engine.alloc_bytes(max_send) for _ in range(2)]
        self.recv_bufs = [self.engine.alloc_bytes(max_recv) for _ in range(2)]
        self.exchange_bytes_per_matvec = sum(sum(b) for b in self.send_bytes)

    def matvec(self, x, y, check: bool = False):
        """y <- H x for this rank's blocks of the hashed vectors.

        Software pipeline, depth 2: exchange(r + 1) is issued (async) right after generate(r + 1) and
        before scatter(r), so the all-to-all of one round overlaps the scatter of the previous and the
        generate of the next one; send/recv buffers alternate between two slots."""
        dist, pb = self.dist, self.packet_bytes
        eng = self.engine
        R = self.num_rounds
        eng.diag(x, y)  # localDiagonal first: y is assigned (DMV:1062-1063)

        def exchange(r):
            send, recv = self.send_bufs[r & 1], self.recv_bufs[r & 1]
            in_splits, out_splits = self.send_bytes[r], self.recv_bytes[r]
            n_in, n_out = sum(in_splits), sum(out_splits)
            return self.transport.all_to_all_single(recv[:max(n_out, 0)], send[:max(n_in, 0)], out_splits, in_splits,
                                                    async_op=True)

        eng.generate(0, x, y, self.send_bufs[0])
        work = exchange(0)
        for r in range(R):
            nxt = None
            if r + 1 < R:
                eng.generate(r + 1, x, y, self.send_bufs[(r + 1) & 1])
                nxt = exchange(r + 1)
            work.wait()
            recv = self.recv_bufs[r & 1]
            offs, off = [], 0
            for s in range(self.P):
                offs.append(off)
                off += self.recv_bytes[r][s]
            if hasattr(eng, "scatter_round"):  # all segments of the round in one consumer launch
                eng.scatter_round(recv, self.recv_counts[r], offs, y)
            else:
                for s in range(self.P):
                    if self.recv_counts[r][s]:
                        eng.scatter(recv, offs[s], self.recv_counts[r][s], y)
            work = nxt
        if check:
            eng.check()

    # PRIMME-style reductions across locales (/root/reference/src/PRIMME.chpl:267-373) ------------
    def global_sum(self, t):
        """globalSumReal: in-place sum over all locales."""
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast(self, t, src: int = 0):
        """broadcastReal."""
        self.dist.broadcast(t, src=src, group=self.group)
        return t

    def dot(self, a, b):
        """<a, b> over the whole hashed vector (conjugating a)."""
        torch = self.torch
        local = torch.vdot(a, b) if a.is_complex() else torch.dot(a, b)
        local = local.reshape(1).clone()
        if local.is_complex():
            parts = torch.view_as_real(local).clone()
            self.global_sum(parts)
            return torch.view_as_complex(parts)[0]
        return self.global_sum(local)[0]


class RcclDistributedOperator:
    """matrixVectorProduct with one locale per process, the exchange done by the C host
    (ls_amd_dist_matvec: generate -> grouped ncclSend/ncclRecv -> scatter, double-buffered over two HIP
    streams; include/ls_amd.h).  Python only bootstraps the communicator's unique id through the existing
    torch.distributed group and forwards pointers; nothing of the data path goes through torch."""

    class _Engine:
        def __init__(self, plan):
            self.plan = plan

        def check(self):
            self.plan.check()

    def __init__(self, matrix, representatives, dtype, group=None, comm=None, num_rounds: int = 0):
        from .api import Communicator, DistMatvec

        self.comm = comm if comm is not None else Communicator.from_torch(group)
        self.rank, self.P = self.comm.rank, self.comm.size
        self.dm = DistMatvec(self.comm, matrix, representatives, dtype, num_rounds)
        self.engine = self._Engine(self.dm.plan)
        self.num_rounds = self.dm.num_rounds
        self.exchange_bytes_per_matvec = self.dm.exchange_bytes

    def matvec(self, x, y, check: bool = False):
        self.dm.matvec(x, y, check=check)

    def inject_fault(self) -> bool:
        return self.dm.inject_fault()

    def global_sum(self, t):
        """globalSumReal (PRIMME.chpl:267-322) on a device tensor, in place"""
        import torch

        if t.is_complex():
            self.comm.allreduce_sum(torch.view_as_real(t))
            return t
        return self.comm.allreduce_sum(t)

    def broadcast(self, t, src: int = 0):
        return self.comm.broadcast(t, src)

    def dot(self, a, b):
        import torch

        local = (torch.vdot(a, b) if a.is_complex() else torch.dot(a, b)).reshape(1).clone()
        return self.global_sum(local)[0]


class RcclReplicatedOperator:
    """matrixVectorProduct with one locale per process and the replicated-x exchange done by the C host
    (ls_amd_repl_matvec, include/ls_amd.h): blocks of x to every peer with one grouped ncclSend/ncclRecv, one permutation
    pass into global order, the pull kernels on a contiguous block of global rows, results back to their owners with one
    all-to-all-v.  Same interface as RcclDistributedOperator; Hermitian operators only."""

    def __init__(self, matrix, reps_global, masks, dtype, group=None, comm=None):
        from .api import Communicator, ReplMatvec

        self.comm = comm if comm is not None else Communicator.from_torch(group)
        self.rank, self.P = self.comm.rank, self.comm.size
        self.rm = ReplMatvec(self.comm, matrix, reps_global, masks, dtype)
        self.engine = RcclDistributedOperator._Engine(self.rm.plan)
        self.exchange_bytes_per_matvec = self.rm.exchange_bytes
        self.x_bytes_in = self.rm.x_in_bytes  # what this rank receives of x per matvec (sub-range exchange: < N - N/P elements)

    def matvec(self, x, y, check: bool = False):
        self.rm.matvec(x, y, check=check)

    def inject_fault(self) -> bool:
        return self.rm.inject_fault()

    global_sum = RcclDistributedOperator.global_sum
    broadcast = RcclDistributedOperator.broadcast
    dot = RcclDistributedOperator.dot


def exchange_memory_estimate(n_global: int, n_local: int, P: int, elem_bytes: int, projected: bool, terms_per_row: int,
                             rows_per_round: int | None = None, krylov_vectors: int = 0):
    """Per-rank HBM (bytes) of the two exchange strategies, next to what the solver itself keeps:
      replicated  O(N): the global representatives (8 B / state), `masks` (1), the gathered x (w), and -- projected bases -- the
                  static {rep -> slot} table (16-32 B) + the row -> slot permutation (4) + norms, or -- unprojected bases -- x in
                  global order (w) + the permutation (4) + the staged kernel's records (8 / P); y staging 3 w / P.  Fast (x is
                  ~30x smaller than the packets) but it only serves a basis whose tables fit EVERY rank.
      packets     O(N / P): this rank's representatives, index structure and vectors, two send and two receive buffers of one
                  round (rows_per_round x terms x packet bytes each) -- the reference's formulation (DMV:663-853) and the strategy
                  that scales in capacity; optionally the all-destinations directory of the pre-indexed packets (P / 4 B per state,
                  dropped by itself when it does not fit)."""
    w = elem_bytes
    rpr = rows_per_round or _rows_per_round()
    if projected:
        table = 16 * (1 << max(1, (2 * n_global - 1).bit_length()))  # 2 N entries of 8 B, rounded up to a power of two
        repl = n_global * (8 + 1 + 4 + w) + table + 8 * n_local + 3 * w * (n_global // P + 1)
    else:
        repl = n_global * (8 + 1 + 4 + 2 * w) + (8 + 3 * w) * (n_global // P + 1)
    rows = min(n_local, rpr)
    key, tables = 8, 0
    if not projected and P >= 2 and not os.environ.get("LS_AMD_ROWS_PER_ROUND"):
        # sorted streams (unprojected fixed-weight bases, exchange operators; csrc/dist.c): at most three rounds unless a buffer
        # would pass ~24 GB, 12-byte pre-indexed packets, the own partition's packets in the send buffer too, the (tile, destination,
        # stream) table of the producer (4 B x P x 2 terms per 256 rows) and the all-destinations directory (P / 4 B per state)
        key = 4
        rows = max(rows, -(-n_local // 3))
        rows = min(rows, max(1, (24 << 30) // (max(1, terms_per_row // 2) * (4 + w) + 16)), max(1, (1 << 32) // max(1, terms_per_row)))
        tables = (n_local // 256 + 1) * P * 2 * terms_per_row * 4 + n_global * P // 4
    # (about half of the flip-mask groups act on a given state of the Heisenberg models; the buffers are sized from the plan's exact
    # counts and shrink with LS_AMD_ROWS_PER_ROUND -- they do not grow with N)
    packets = n_local * (8 + 8 + 4) + 4 * rows * max(1, terms_per_row // 2) * (key + w) + (8 * n_local if projected else 0) + tables
    vectors = (2 + krylov_vectors) * w * n_local
    return {"replicated": int(repl + vectors), "packets": int(packets + vectors), "vectors": int(vectors)}


def choose_exchange(hermitian: bool, estimate: dict, free_bytes: int, ceiling: int | None = None):
    """'replicated' when the operator is Hermitian and its O(N) tables fit the HBM this rank may use, else 'packets' (the O(N / P)
    strategy).  LS_AMD_EXCHANGE_HBM_CEILING (bytes) stands in for the free memory (tests; capacity planning)."""
    env = os.environ.get("LS_AMD_EXCHANGE_HBM_CEILING")
    limit = int(env) if env else (ceiling if ceiling is not None else int(0.9 * free_bytes))
    if hermitian and estimate["replicated"] <= limit:
        return "replicated"
    return "packets"


class HipReplicatedEngine:
    """ls_amd replicated-x plan for this rank (include/ls_amd.h)."""

    def __init__(self, matrix, reps_local, reps_global, dtype, num_partitions, my_partition):
        from .api import ReplicatedPlan

        self.plan = ReplicatedPlan(matrix, reps_local, reps_global, dtype, num_partitions, my_partition)

    def matvec(self, x_global, y_local):
        self.plan.matvec(x_global, y_local, check=False)

    def check(self):
        self.plan.check()


class ReplicatedOperator:
    """matrixVectorProduct with one locale per process, exchanging x (and y) instead of packets.

    x, y and the representatives stay hash-partitioned at the interface.  Per matvec:
      1. every rank sends its block of x to every peer directly (one grouped send/recv over all xGMI
         links, not a ring); one gather pass through a permutation built once from `masks` puts the
         blocks into global ascending order (arrFromHashedToBlock, HashedToBlock.chpl:67-153);
      2. the pull kernel computes a CONTIGUOUS range of global rows [N r / P, N (r + 1) / P) -- contiguous
         rows keep the gather locality of the single-GPU kernel (a hashed eighth of the rows still touches
         two thirds of x's cache lines per bond: measured 9.2 ms instead of 15.6 / 8 ms at P = 8);
      3. the N / P results are partitioned by owner and returned with one all_to_all_single
         (arrFromBlockToHashed restricted to the range; the pieces arrive in source order = ascending).
    Exchange volume N w (P - 1) / P + N w (P - 1) / P^2 bytes per rank instead of nnz (8 + w) (P - 1) / P.
    Needs a Hermitian operator and N w bytes of HBM per rank for the replicated x."""

    def __init__(self, matrix, reps_local, reps_global, masks, dtype, group=None, engine_factory=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.P = dist.get_world_size(group)
        self.dtype = dtype
        dev = reps_local.device
        backend = dist.get_backend(group)
        meta_device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        n = int(masks.numel())
        m64 = masks.to(torch.int64)
        counts = torch.bincount(m64, minlength=self.P).tolist()
        assert counts[self.rank] == int(reps_local.numel()), "masks do not describe this rank's block"
        self.counts = counts
        self.max_count = max(counts)
        # perm[i] = position of global state i inside the [P, max_count] gathered buffer
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        for p in range(self.P):
            sel = (m64 == p).nonzero(as_tuple=True)[0]
            perm[sel] = p * self.max_count + torch.arange(sel.numel(), device=dev, dtype=torch.int64)
        self.perm = perm.to(torch.int32) if self.P * self.max_count < 2**31 else perm
        del perm
        self.gathered = torch.empty(self.P * self.max_count, dtype=dtype, device=dev)
        self.x_global = torch.empty(n, dtype=dtype, device=dev)
        # this rank computes the contiguous global rows [n0, n1)
        self.n0, self.n1 = n * self.rank // self.P, n * (self.rank + 1) // self.P
        mslice = m64[self.n0:self.n1]
        order = torch.argsort(mslice, stable=True)
        self.y_order = order.to(torch.int32) if (self.n1 - self.n0) < 2**31 else order
        self.y_send_counts = torch.bincount(mslice, minlength=self.P).tolist()
        S = torch.tensor(self.y_send_counts, dtype=torch.int64, device=meta_device)
        R = torch.zeros_like(S)
        dist.all_to_all_single(R, S, group=group)
        self.transport = _Transport(group)
        self.y_recv_counts = R.cpu().tolist()
        assert sum(self.y_recv_counts) == counts[self.rank]
        del m64, mslice, order
        self.y_block = torch.zeros(self.n1 - self.n0, dtype=dtype, device=dev)
        self.y_send = torch.empty(self.n1 - self.n0, dtype=dtype, device=dev)
        self.y_recv = torch.empty(counts[self.rank], dtype=dtype, device=dev)
        factory = engine_factory or HipReplicatedEngine
        self.engine = factory(matrix, reps_global[self.n0:self.n1], reps_global, dtype, self.P, self.rank)
        self.accumulate = getattr(matrix, "numberDiagTerms", lambda: 1)() == 0  # y += H x when H has no diagonal
        es = self.x_global.element_size()
        self.exchange_bytes_per_matvec = (n - counts[self.rank]) * es + (self.n1 - self.n0 - self.y_send_counts[self.rank]) * es

    def gather_x(self, x_local):
        """all ranks' blocks -> self.x_global (global ascending order)."""
        torch, dist = self.torch, self.dist
        P, mc = self.P, self.max_count
        outs = [self.gathered[p * mc:p * mc + self.counts[p]] for p in range(P)]
        # every block goes straight to every peer: one grouped send/recv (RCCL uses all xGMI links at
        # once; an all_gather would be a ring)
        self.transport.exchange_blocks(x_local, outs, self.rank, P)
        self._gather(self.gathered, self.perm, self.x_global)
        return self.x_global

    def _gather(self, src, perm, out):
        """out[i] = src[perm[i]] on the device: the library's permutation kernel (ls_amd_gather); torch.index_select when the
        tensors live on the CPU (gloo tests with injected engines)"""
        if not src.is_cuda:
            self.torch.index_select(src, 0, perm, out=out)
            return
        import ctypes as C

        from . import _lib
        from .api import _stream_ptr

        _lib.check(_lib.load().ls_amd_gather(out.numel(), C.c_void_p(perm.data_ptr()), 1 if perm.dtype == self.torch.int64 else 0,
                                             out.element_size(), C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), _stream_ptr()))

    def matvec(self, x, y, check: bool = False):
        torch, dist = self.torch, self.dist
        xg = self.gather_x(x)
        if self.accumulate:
            self.y_block.zero_()
        self.engine.matvec(xg, self.y_block)
        self._gather(self.y_block, self.y_order, self.y_send)
        if self.P > 1:
            self.transport.all_to_all_single(self.y_recv, self.y_send, self.y_recv_counts, self.y_send_counts)
        else:
            self.y_recv.copy_(self.y_send)
        if self.accumulate:
            y.add_(self.y_recv)
        else:
            y.copy_(self.y_recv)
        if check:
            self.engine.check()

    global_sum = DistributedOperator.global_sum
    broadcast = DistributedOperator.broadcast
    dot = DistributedOperator.dot


# ---------------------------------------------------------------------------------------------
# Layout converters ACROSS processes and per-rank block I/O: the reference keeps vectors hash-partitioned while it
# computes and block-distributed on disk (HashedToBlock.chpl:67-153, BlockToHashed.chpl:87-208, MyHDF5.chpl:272-333); with
# one process per GPU the conversion is one all-to-all-v whose counts every rank derives from `masks` (the owner of every
# state in global ascending order) without talking to anyone, and every rank reads / writes its own hyperslab.
# ---------------------------------------------------------------------------------------------
def _layout_counts(masks, rank: int, world: int):
    """counts[s][b] = states of hash partition s that fall into block b (Chapel's Block distribution of the global order),
    the block of this rank, and the position of every state of that block inside its owner's part of the block."""
    import torch

    from . import hdf5

    n = int(masks.numel())
    bounds = [hdf5.block_range(n, world, b) for b in range(world)]
    m = masks.to(torch.int64)
    counts = torch.zeros((world, world), dtype=torch.int64)
    for b, (lo, hi) in enumerate(bounds):
        if hi > lo:
            counts[:, b] = torch.bincount(m[lo:hi].cpu(), minlength=world)[:world]
    return counts, bounds[rank]


def hashed_to_block(part, masks, group=None):
    """arrFromHashedToBlock across processes: `part` = this rank's elements (its states in ascending order); returns this
    rank's block of the vector in global ascending order.  Every run a source sends is contiguous in its part (a part is
    ascending in the global index), so the send buffer is the part itself."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    counts, (lo, hi) = _layout_counts(masks, rank, world)
    if int(counts[rank].sum()) != int(part.numel()):
        raise ValueError(f"rank {rank} holds {part.numel()} elements, the masks give it {int(counts[rank].sum())}")
    recv = torch.empty(hi - lo, dtype=part.dtype, device=part.device)
    _Transport(group).all_to_all_single(recv, part.contiguous(), [int(c) for c in counts[:, rank]], [int(c) for c in counts[rank]])
    # recv = [elements owned by 0 | owned by 1 | ...] of this block, each run ascending: a stable sort of the block's owners
    # gives the position every global index of the block takes in that arrangement
    order = torch.sort(masks[lo:hi].to(torch.int64).to(part.device), stable=True).indices
    out = torch.empty_like(recv)
    out[order] = recv
    return out


def block_to_hashed(block, masks, group=None):
    """arrFromBlockToHashed across processes: the inverse of hashed_to_block"""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    counts, (lo, hi) = _layout_counts(masks, rank, world)
    if int(block.numel()) != hi - lo:
        raise ValueError(f"rank {rank} holds a block of {block.numel()} elements, the Block distribution gives it {hi - lo}")
    order = torch.sort(masks[lo:hi].to(torch.int64).to(block.device), stable=True).indices
    send = block.contiguous()[order]
    part = torch.empty(int(counts[rank].sum()), dtype=block.dtype, device=block.device)
    _Transport(group).all_to_all_single(part, send, [int(c) for c in counts[rank]], [int(c) for c in counts[:, rank]])
    return part


def _broadcast_int(value, group=None, device=None, signed=False):
    """rank 0's integer on every rank (None travels as -1; signed: negative values are returned as they are)"""
    import torch
    import torch.distributed as dist

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([-1 if value is None else int(value)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    v = int(t.item())
    return v if signed else (None if v < 0 else v)


def write_block_dataset(path: str, name: str, shape, dtype, rows, group=None):
    """writeDatasetAsBlocks across processes (MyHDF5.chpl:303-333): rank 0 creates dataset `name` of `shape` with its storage
    allocated, every rank then writes its own hyperslab of the last dimension -- all ranks AT THE SAME TIME, each straight to
    the bytes of its hyperslab (hdf5.write_hyperslab_raw: disjoint byte ranges of one contiguous dataset, no HDF5 metadata
    touched by anyone but rank 0), the `coforall loc in Locales` of the reference.  `rows` yields this rank's block of every
    leading index in order (rank-1 dataset: exactly one block).  A library that will not give the address of the storage
    falls back to one H5Dwrite at a time (LS_AMD_HDF5_RAW=0 forces that).  The file must be visible to every rank."""
    import os

    import numpy as np
    import torch.distributed as dist

    from . import hdf5

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shape = tuple(int(v) for v in shape)
    lo, hi = hdf5.block_range(shape[-1], world, rank)
    address = None
    dist.barrier(group)  # nobody still holds the file open (HDF5 locks it: a reader elsewhere makes rank 0's H5Fopen(RDWR) fail)
    failure = None
    if rank == 0:
        try:
            address = hdf5.create_dataset(path, name, shape, dtype, allocate=os.environ.get("LS_AMD_HDF5_RAW", "1") != "0")
        except Exception as e:  # noqa: BLE001 -- the peers wait in the broadcast below: they must hear about it
            failure = e
    # (also the barrier the reference needs before the first H5Dopen, MyHDF5.chpl:219-221)
    code = _broadcast_int(-2 if failure is not None else (-1 if address is None else address), group, signed=True)
    if code == -2:
        raise OSError(f"rank 0 could not create dataset '{name}' of '{path}'") from failure
    address = None if code < 0 else code
    lead = shape[:-1]
    for row, blk in enumerate(rows):
        blk = np.ascontiguousarray(blk, dtype=dtype)
        if blk.shape != (hi - lo,):
            raise ValueError(f"rank {rank}: block of {blk.shape} elements, the Block distribution gives it {hi - lo}")
        idx = tuple(int(v) for v in np.unravel_index(row, lead)) if lead else ()
        if address is not None:
            hdf5.write_hyperslab_raw(path, address, shape, idx + (lo,), blk.reshape((1,) * len(lead) + blk.shape))
        else:
            for r in range(world):  # serial HDF5 has no file locking across processes worth trusting: one writer at a time
                if r == rank:
                    hdf5.write_dataset_chunk(path, name, idx + (lo,), blk.reshape((1,) * len(lead) + blk.shape))
                dist.barrier(group)
    dist.barrier(group)
    return address is not None


def write_hashed_vectors(path: str, name: str, parts, masks, group=None):
    """writeDatasetAsBlocks for hash-partitioned vectors (Diagonalize.chpl:248-256 -> MyHDF5.chpl:303-333): `parts` = this
    rank's pieces of k vectors; dataset [k, N] in global ascending order, every rank writing its own hyperslab
    (write_block_dataset: all ranks at once).  The file must be visible to every rank (one node, or a shared file system --
    as in the reference)."""
    import numpy as np

    if any(p.is_complex() for p in parts):
        raise NotImplementedError("HDF5 output is implemented for real vectors (the reference's eltType is real(64))")
    n = int(masks.numel())
    # (a generator: the host holds one block at a time -- chain_40_symm vectors are 6.9 GB each)
    rows = (hashed_to_block(p, masks, group).cpu().numpy() for p in parts)
    return write_block_dataset(path, name, (len(parts), n), np.float64, rows, group)


def read_hashed_vector(path: str, name: str, row: int, masks, group=None, device=None):
    """readDatasetAsBlocks + arrFromBlockToHashed: every rank reads its block of row `row` of dataset [k, N] and the ranks
    exchange them into the hash partition"""
    import torch
    import torch.distributed as dist

    from . import hdf5

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = int(masks.numel())
    lo, hi = hdf5.block_range(n, world, rank)
    blk = torch.from_numpy(hdf5.read_dataset_chunk(path, name, (row, lo), (1, hi - lo))[0])
    if device is not None:
        blk = blk.to(device)
    return block_to_hashed(blk, masks, group)

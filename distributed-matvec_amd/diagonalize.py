"""Eigensolver driver: the caller of the matvec path in BASELINE config 5
(`/root/reference/src/Diagonalize.chpl:258-332`: load YAML -> basis states -> PRIMME `dprimme` with
`ls_chpl_primme_matvec` -> eigenpairs).

libprimme is not available here (headers only, `/root/reference/primme_headers/`), so the PRIMME-ABI
callbacks stay exported from the C library for a real PRIMME (include/ls_chpl.h) and
this module ships a small restarted Lanczos that keeps every vector in HBM — PRIMME hands host
pointers to its matvec callback, which costs a PCIe round trip of x and y per call,
so a device-resident solver is the MI355X-first way to drive this path.

The solver only needs three things from its operator: `matvec(x, y)`, a global dot product and the
local vector length — provided both by a single-process plan (all partitions on one GPU) and by
`DistributedOperator` (one partition per rank; dots = all_reduce = PRIMME's globalSumReal,
`/root/reference/src/PRIMME.chpl:267-311`).
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field


class LocalOperator:
    """H on one device: P logical partitions concatenated into flat per-partition vectors."""

    def __init__(self, matrix, representatives, dtype, mode="auto", slot_cache_bytes=0):
        import torch

        from .api import MatvecPlan

        self.torch = torch
        self.reps = list(representatives)
        self.plan = MatvecPlan(matrix, self.reps, dtype, mode=mode)
        # an eigensolver applies ONE plan hundreds of times: keep the resolved packet streams of a projected basis in HBM
        # when the caller has room for them (ls_amd_plan_cache_slots; 0 rows = the plan stays matrix-free)
        self.cached_rows = self.plan.cache_slots(slot_cache_bytes) if slot_cache_bytes > 0 and len(self.reps) == 1 else 0
        self.dtype = dtype
        self.sizes = [int(r.numel()) for r in self.reps]
        self.n_local = sum(self.sizes)
        self.device = self.reps[0].device
        self.matvecs = 0

    def _split(self, v):
        return list(v.split(self.sizes)) if len(self.sizes) > 1 else [v]

    def matvec(self, x, y):
        y.zero_()
        self.plan.matvec(self._split(x), self._split(y), check=False)
        self.matvecs += 1

    def check(self):
        self.plan.check()

    def dot(self, a, b):
        return self.torch.vdot(a, b) if a.is_complex() else self.torch.dot(a, b)

    def new_vector(self):
        return self.torch.zeros(self.n_local, dtype=self.dtype, device=self.device)

    def random_vector(self, seed):
        from .api import fillRandom

        return self.torch.cat([fillRandom(r, seed, self.dtype) for r in self.reps])


class RankOperator:
    """H with one partition per process (wraps distributed.DistributedOperator)."""

    def __init__(self, dist_op, representatives, dtype):
        import torch

        self.torch = torch
        self.op = dist_op
        self.reps = representatives
        self.dtype = dtype
        self.n_local = int(representatives.numel())
        self.device = representatives.device
        self.matvecs = 0

    def matvec(self, x, y):
        y.zero_()
        self.op.matvec(x, y, check=False)
        self.matvecs += 1

    def check(self):
        self.op.engine.check()

    def dot(self, a, b):
        return self.op.dot(a, b)

    def global_sum(self, t):
        if t.is_complex():
            r = self.torch.view_as_real(t).contiguous()
            self.op.global_sum(r)
            return self.torch.view_as_complex(r)
        return self.op.global_sum(t)

    def new_vector(self):
        return self.torch.zeros(self.n_local, dtype=self.dtype, device=self.device)

    def random_vector(self, seed):
        from .api import fillRandom

        return fillRandom(self.reps, seed, self.dtype)


@dataclass
class EigenResult:
    eigenvalues: list
    eigenvectors: list
    residual_norms: list
    matvecs: int
    restarts: int
    converged: bool
    seconds: float
    history: list = field(default_factory=list)


def lanczos_smallest(op, num_evals: int = 1, eps: float = 1e-6, max_basis: int = 24, max_restarts: int = 500,
                     seed: int = 1234, verbose: bool = False) -> EigenResult:
    """Smallest `num_evals` eigenpairs of the Hermitian operator `op` (target = primme_smallest, eps as
    in `/root/reference/src/Diagonalize.chpl:164-172,205`): thick-restart Lanczos (Wu & Simon) with
    full re-orthogonalisation (classical Gram-Schmidt twice, one GEMV per pass) inside a window of
    `max_basis` device vectors.  Residuals come from the Lanczos relation
    ||H y_i - theta_i y_i|| = |beta_m s_{m,i}|, so a step costs exactly one matvec.
    Converged when every wanted residual <= eps * ||H|| (estimated by the largest |Ritz value|)."""
    import numpy as np
    import torch

    t0 = time.perf_counter()
    k = num_evals
    m = max(max_basis, 2 * k + 4)
    v = op.random_vector(seed)
    n = v.numel()
    cplx = v.is_complex()
    V = torch.empty((m + 1, n), dtype=v.dtype, device=v.device)
    T = np.zeros((m + 1, m), dtype=complex if cplx else float)

    def gsum(t):
        return op.global_sum(t) if hasattr(op, "global_sum") else t

    def norm(u):
        # (no full-size temporaries: the vectors of chain_40_symm are 6.9 GB each and the HBM next to them belongs to the slot cache)
        return math.sqrt(float(gsum((torch.linalg.vector_norm(u) ** 2).reshape(1))[0]))

    torch.div(v, norm(v), out=V[0])
    del v
    w = op.new_vector()
    # fused orthogonalisation sweeps (real f64 vectors on the device; the partial sums of a rank go through the operator's global sum)
    fused, fused_rows = None, 0
    if not cplx and V.is_cuda and V.dtype == torch.float64 and os.environ.get("LS_AMD_FUSED_ORTH", "1") != "0":
        import ctypes as C

        from . import _lib
        from .api import _stream_ptr

        lib = _lib.load()
        fused_rows = int(lib.ls_amd_orth_max_rows())
        obuf = torch.zeros(fused_rows + 1, dtype=torch.float64, device=V.device)

        def fused(rows, Vb, wv, hin):
            _lib.check(lib.ls_amd_orth_pass(rows, n, C.c_void_p(Vb.data_ptr()), Vb.stride(0), C.c_void_p(wv.data_ptr()),
                                            C.c_void_p(hin.data_ptr()) if hin is not None else None, C.c_void_p(obuf.data_ptr()), _stream_ptr()))
            return gsum(obuf[: rows + 1].clone())
    j0 = 0  # number of locked (restarted) vectors at the front of V
    restarts = 0
    history = []
    prof = {"orth": 0.0, "normalise": 0.0, "restart": 0.0} if os.environ.get("LS_AMD_LANCZOS_PROFILE") else None

    def tick():
        if prof is None:
            return 0.0
        torch.cuda.synchronize()
        return time.perf_counter()

    while True:
        for j in range(j0, m):
            op.matvec(V[j], w)
            Vj = V[: j + 1]
            t_a = tick()
            if fused is not None and j + 1 <= fused_rows:
                # classical Gram-Schmidt with the fused sweeps of csrc/orth.hip: pass 1 -> h and ||w||^2; pass 2 applies h and returns
                # the overlaps that are left and the new norm in the same sweep.  A third sweep only when orthogonality was
                # really lost (the left-over overlaps are not at rounding level relative to what remains of w)
                out = fused(j + 1, Vj, w, None)
                h = out[: j + 1].clone()
                out = fused(j + 1, Vj, w, h)
                o2 = out.cpu().numpy()
                h2n, n1 = float(np.sqrt((o2[: j + 1] ** 2).sum())), math.sqrt(max(float(o2[j + 1]), 0.0))
                if h2n > 1e-11 * n1:
                    h2 = out[: j + 1].clone()
                    out = fused(j + 1, Vj, w, h2)
                    h = h + h2
                    n1 = math.sqrt(max(float(out[j + 1].item()), 0.0))
                h = h.cpu().numpy()
                beta = n1
            else:
                h = gsum(torch.mv(Vj.conj() if cplx else Vj, w))
                w -= torch.mv(Vj.t(), h)
                h2 = gsum(torch.mv(Vj.conj() if cplx else Vj, w))
                w -= torch.mv(Vj.t(), h2)
                h = (h + h2).cpu().numpy()
                beta = norm(w)
            if prof is not None:
                prof["orth"] += tick() - t_a
            T[: j + 1, j] = h
            T[j + 1, j] = beta
            if beta < 1e-14 * max(1.0, float(np.abs(h).max())):
                m_eff = j + 1
                break
            t_a = tick()
            torch.div(w, beta, out=V[j + 1])
            if prof is not None:
                prof["normalise"] += tick() - t_a
        else:
            m_eff = m
        Tm = T[:m_eff, :m_eff]
        Tm = 0.5 * (Tm + Tm.conj().T)
        theta, S = np.linalg.eigh(Tm)
        beta_m = T[m_eff, m_eff - 1].real if m_eff == m else 0.0
        kk = min(k, m_eff)
        res = [abs(beta_m * S[m_eff - 1, i]) for i in range(kk)]
        scale = max(float(np.abs(theta).max()), 1e-300)
        history.append((op.matvecs, [float(t) for t in theta[:kk]], res))
        if verbose:
            print(f"[lanczos] restart {restarts}: matvecs={op.matvecs} theta={theta[:kk]} res={res}", flush=True)
        done = all(r <= eps * scale for r in res) or m_eff < m
        if done or restarts >= max_restarts:
            St = torch.as_tensor(S[:, :kk], dtype=V.dtype, device=V.device)
            Y = torch.mm(St.t(), V[:m_eff])  # Ritz vectors, [kk, n]
            vecs = [Y[i].clone() for i in range(kk)]
            op.check()
            return EigenResult([float(t) for t in theta[:kk]], vecs, [float(r) for r in res], op.matvecs, restarts, bool(done),
                               time.perf_counter() - t0, history)
        # thick restart: keep `keep` Ritz vectors, then the next Lanczos vector
        t_a = tick()
        keep = min(m_eff - 2, kk + max(2, (m_eff - kk) // 3))
        St = torch.as_tensor(S[:, :keep], dtype=V.dtype, device=V.device)
        if fused is not None and m_eff <= fused_rows:
            # V[:keep] <- S^T V[:m_eff] in place, one read of m_eff and one write of keep vectors (csrc/orth.hip)
            Sc = St.contiguous()
            _lib.check(lib.ls_amd_basis_rotate(m_eff, keep, n, C.c_void_p(V.data_ptr()), V.stride(0), C.c_void_p(Sc.data_ptr()), _stream_ptr()))
        else:
            # ... in column blocks: the product of whole rows would be a temporary of `keep` vectors
            step = 1 << 24
            for c0 in range(0, n, step):
                c1 = min(n, c0 + step)
                V[:keep, c0:c1] = torch.mm(St.t(), V[:m_eff, c0:c1])
        V[keep] = V[m_eff]
        T[:, :] = 0
        for i in range(keep):
            T[i, i] = theta[i]
            T[keep, i] = beta_m * S[m_eff - 1, i]  # coupling of the kept Ritz vectors to the next vector
            T[i, keep] = np.conj(T[keep, i])
        j0 = keep
        restarts += 1
        if prof is not None:
            prof["restart"] += tick() - t_a
            print(f"[lanczos] profile after restart {restarts}: " + ", ".join(f"{k} {v:.2f} s" for k, v in prof.items()), flush=True)


def diagonalize(config, num_evals: int = 1, eps: float = 1e-6, num_partitions: int = 1, dtype=None, output: str | None = None,
                max_basis: int = 24, verbose: bool = False):
    """`Diagonalize.main` (Diagonalize.chpl:258-332) on one device: config (dict or YAML path) ->
    representatives (enumerated on the GPU; an existing HDF5 `output` that already holds basis/representatives is
    reused and extended, like makeBasisStates :227-246) ->
    eigenpairs; `output` (.npz) receives what the reference writes to its HDF5 groups:
    basis/representatives, hamiltonian/eigenvalues, hamiltonian/eigenvectors, hamiltonian/residuals."""
    import os

    import numpy as np
    import torch

    from . import api

    if isinstance(config, str):
        basis, h = api.loadConfigFromYaml(config, hamiltonian=True)
    else:
        basis, h = api.loadConfigFromDict(config, hamiltonian=True)
    dtype = dtype or torch.float64
    reps = None
    if output and output.endswith((".h5", ".hdf5")) and num_partitions == 1:
        # makeBasisStates (Diagonalize.chpl:227-246): representatives already stored in the output file are reused
        from . import hdf5

        if hdf5.has_dataset(output, "/basis/representatives"):
            stored = hdf5.read_dataset(output, "/basis/representatives").astype(np.uint64)
            # a stale file (another model, another sector, another Hamming weight, a truncated dataset) must not be
            # diagonalised silently.  The reference reuses stored representatives to skip an enumeration that takes its CPU path
            # hours; here the enumeration is a few seconds of GPU time even for chain_40_symm, so the stored array is simply
            # required to EQUAL what this basis enumerates to -- order, count and every state.
            fresh, _ = api.enumerateStates(basis, 1)
            ok = len(stored) == int(fresh[0].numel())
            if ok:
                ok = bool(torch.equal(torch.from_numpy(stored.view(np.int64)).cuda(), fresh[0]))
            del fresh
            if not ok:
                raise api.LsAmdError(f"halt: /basis/representatives of '{output}' does not belong to the configured basis "
                                     "(stale output file?); remove the dataset or the file")
            reps = [torch.from_numpy(stored.view(np.int64).copy()).cuda()]
            masks = torch.zeros(len(stored), dtype=torch.uint8, device="cuda")
    if reps is None:
        reps, masks = api.enumerateStates(basis, num_partitions)
    # one plan, hundreds of matvecs: what is left of HBM after the Krylov basis may hold the resolved packet streams
    # (ls_amd_plan_cache_slots; LS_AMD_SLOT_CACHE=0 keeps the solver matrix-free)
    cache_bytes = 0
    # (LS_AMD_SLOT_CACHE has one meaning everywhere: bytes; 0 = off; set = the C plan applies it at creation, nothing to add here)
    if num_partitions == 1 and os.environ.get("LS_AMD_SLOT_CACHE") is None:
        n_states = int(reps[0].numel())
        free, _total = torch.cuda.mem_get_info()
        cache_bytes = max(0, int(free) - (max_basis + 6) * n_states * (16 if dtype == torch.complex128 else 8) - (4 << 30))
    op = LocalOperator(h, reps, dtype, slot_cache_bytes=cache_bytes)
    if verbose and op.cached_rows:
        print(f"[diagonalize] slot cache: {op.cached_rows} rows, {op.plan.slot_cache[1] / 1e9:.2f} GB", flush=True)
    r = lanczos_smallest(op, num_evals=num_evals, eps=eps, max_basis=max_basis, verbose=verbose)
    if output and output.endswith((".h5", ".hdf5")):
        # same groups/datasets as the reference's output file (Diagonalize.chpl:241,248-256)
        from . import hdf5

        to_block = lambda v: api.arrFromHashedToBlock(list(v.split(op.sizes)), masks) if num_partitions > 1 else v  # noqa: E731
        if any(v.is_complex() for v in r.eigenvectors):
            raise NotImplementedError("HDF5 output is implemented for real eigenvectors (the reference's eltType is real(64))")
        hdf5.write_datasets(output, append=True, datasets={
            "/basis/representatives": (api.arrFromHashedToBlock(reps, masks) if num_partitions > 1 else reps[0]).cpu().numpy().view(np.uint64),
            "/hamiltonian/eigenvalues": np.array(r.eigenvalues),
            "/hamiltonian/residuals": np.array(r.residual_norms),
        })
        # the eigenvectors go out block by block (writeDatasetAsBlocks, MyHDF5.chpl:303-333: every locale its own hyperslab of
        # the last dimension): the host holds one block at a time -- chain_40_symm vectors are 6.9 GB each
        n_states = int(sum(op.sizes)) if num_partitions > 1 else int(reps[0].numel())
        num_blocks = max(num_partitions, -(-n_states // (1 << 27)))
        hdf5.create_dataset(output, "/hamiltonian/eigenvectors", (len(r.eigenvectors), n_states), np.float64)
        for row, v in enumerate(r.eigenvectors):
            vb = to_block(v)
            for b in range(num_blocks):
                lo, hi = hdf5.block_range(n_states, num_blocks, b)
                hdf5.write_dataset_chunk(output, "/hamiltonian/eigenvectors", (row, lo), vb[lo:hi].cpu().numpy()[None, :])
    elif output:
        parts_to_block = lambda v: api.arrFromHashedToBlock(list(v.split(op.sizes)), masks) if num_partitions > 1 else v  # noqa: E731
        np.savez(
            output,
            **{
                "basis/representatives": (api.arrFromHashedToBlock(reps, masks) if num_partitions > 1 else reps[0]).cpu().numpy().view(np.uint64),
                "hamiltonian/eigenvalues": np.array(r.eigenvalues),
                "hamiltonian/residuals": np.array(r.residual_norms),
                "hamiltonian/eigenvectors": np.stack([parts_to_block(v).cpu().numpy() for v in r.eigenvectors]),
            },
        )
    return r


def diagonalize_distributed(config, group=None, num_evals: int = 1, eps: float = 1e-6, dtype=None, output: str | None = None,
                            exchange: str = "auto", max_basis: int = 24, verbose: bool = False):
    """`Diagonalize.main` (Diagonalize.chpl:258-332) with one process per GPU: every rank enumerates the basis on its device,
    keeps its hash partition of the states and of the Lanczos vectors, the matvec goes through the C host's exchange
    (replicated-x for Hermitian operators, packets otherwise / on request), the reductions through RCCL, and the output file
    (HDF5, visible to every rank) is written block-distributed: /basis/representatives and /hamiltonian/eigenvectors [k, N]
    in global ascending order, every rank its own hyperslab, all ranks writing at once (MyHDF5.chpl:303-333;
    distributed.write_block_dataset), eigenvalues and residuals by rank 0.  An `output` that already holds
    /basis/representatives (an earlier run of this driver, or of the reference: makeBasisStates, Diagonalize.chpl:227-246) is
    extended, not truncated: every rank reads its block of the stored states (readDatasetAsBlocks), they must equal what
    the configured basis enumerates to, and the dataset is left as it is.
    Must be called by every rank of `group` (torch.distributed, backend "nccl")."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from . import api, hdf5
    from .distributed import RcclDistributedOperator, RcclReplicatedOperator, hashed_to_block, write_block_dataset, write_hashed_vectors

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if isinstance(config, str):
        basis, h = api.loadConfigFromYaml(config, hamiltonian=True)
    else:
        basis, h = api.loadConfigFromDict(config, hamiltonian=True)
    dtype = dtype or torch.float64
    if output and not output.endswith((".h5", ".hdf5")):
        raise ValueError("diagonalize_distributed writes HDF5 (the block-distributed format of the reference)")
    parts, masks = api.enumerateStates(basis, world)
    my_reps = parts[rank]
    stored_basis = False
    if output:
        from .distributed import _broadcast_int

        # rank 0 looks (one decision for all ranks); -1 = no such dataset
        n_stored = _broadcast_int(hdf5.dataset_shape(output, "/basis/representatives")[-1]
                                  if rank == 0 and hdf5.has_dataset(output, "/basis/representatives") else None, group)
        if n_stored is not None:
            # the reference trusts the stored states to skip an enumeration that costs its CPU path hours; the enumeration
            # above is seconds of GPU time, so here they are CHECKED instead (like diagonalize()): a stale file -- another
            # model, sector or Hamming weight, a truncated dataset -- must not be extended silently.  Every rank compares
            # its own block, read by itself, with its block of the fresh enumeration.
            n = int(masks.numel())
            fresh = hashed_to_block(my_reps, masks, group)
            ok = n_stored == n
            if ok:
                mine = hdf5.read_dataset_block(output, "/basis/representatives", world, rank, np.uint64)
                ok = bool(torch.equal(torch.from_numpy(mine.view(np.int64)), fresh.cpu()))
            del fresh
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=my_reps.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if not int(flag.item()):
                raise api.LsAmdError(f"halt: /basis/representatives of '{output}' does not belong to the configured basis "
                                     "(stale output file?); remove the dataset or the file")
            stored_basis = True
            if verbose and rank == 0:
                print(f"[diagonalize_distributed] /basis/representatives of '{output}': {n} states, verified, kept", flush=True)
    if exchange == "auto":
        # replicated x is the fast exchange, but its tables are O(N) on EVERY rank; the packets are the O(N / P) strategy and the
        # fallback when those tables do not fit next to the Krylov basis (exchange_memory_estimate; LS_AMD_EXCHANGE_HBM_CEILING)
        from .distributed import choose_exchange, exchange_memory_estimate

        free, _total = torch.cuda.mem_get_info() if my_reps.is_cuda else (1 << 62, 0)
        est = exchange_memory_estimate(int(masks.numel()), int(my_reps.numel()), world, 16 if dtype == torch.complex128 else 8,
                                       basis.requiresProjection(), h.numberOffDiagTerms(), krylov_vectors=max_basis + 4)
        exchange = choose_exchange(bool(h.isHermitian), est, int(free))
        if world > 1:  # one decision for all ranks: any rank without room for the replicated tables pulls everybody to the packets
            flag = torch.tensor([1 if exchange == "packets" else 0], dtype=torch.int64, device=my_reps.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            exchange = "packets" if int(flag.item()) else "replicated"
        if verbose and rank == 0:
            print(f"[diagonalize_distributed] exchange = {exchange} (per-rank HBM estimate: replicated {est['replicated'] / 1e9:.2f} GB, "
                  f"packets {est['packets'] / 1e9:.2f} GB, free {free / 1e9:.1f} GB)", flush=True)
    if exchange == "replicated":
        reps_global = api.arrFromHashedToBlock(parts, masks) if world > 1 else parts[0]
        op = RcclReplicatedOperator(h, reps_global, masks, dtype, group=group)
        del reps_global
        # one plan, hundreds of matvecs: this rank's rows keep their resolved packet streams in the HBM the Krylov basis leaves
        # (ls_amd_plan_cache_slots -- a local decision, no collective; rows that do not fit stay matrix-free; LS_AMD_SLOT_CACHE=0: off)
        plan = getattr(getattr(op, "engine", None), "plan", None)
        # (LS_AMD_SLOT_CACHE set: bytes, applied by ls_amd_repl_create itself -- 0 = off; unset: sized here)
        if plan is not None and hasattr(plan, "cache_slots") and my_reps.is_cuda and os.environ.get("LS_AMD_SLOT_CACHE") is None:
            free, _total = torch.cuda.mem_get_info()
            budget = int(free) - (max_basis + 6) * int(my_reps.numel()) * (16 if dtype == torch.complex128 else 8) - (4 << 30)
            if budget > 0:
                rows = plan.cache_slots(budget)
                if verbose and rows:
                    print(f"[diagonalize_distributed] rank {rank}: slot cache for {rows} rows, {plan.slot_cache[1] / 1e9:.2f} GB", flush=True)
    else:
        op = RcclDistributedOperator(h, my_reps, dtype, group=group)
    del parts
    r = lanczos_smallest(RankOperator(op, my_reps, dtype), num_evals=num_evals, eps=eps, max_basis=max_basis, verbose=verbose)
    if output:
        if rank == 0:
            # append: whatever else the file holds stays; hamiltonian/* of an earlier run is replaced (Diagonalize.chpl:248-256)
            hdf5.write_datasets(output, {"/hamiltonian/eigenvalues": np.array(r.eigenvalues),
                                         "/hamiltonian/residuals": np.array(r.residual_norms)}, append=os.path.exists(output))
        dist.barrier(group)
        if not stored_basis:
            write_block_dataset(output, "/basis/representatives", (int(masks.numel()),), np.uint64,
                                (hashed_to_block(my_reps, masks, group).cpu().numpy().view(np.uint64) for _ in range(1)), group)
        write_hashed_vectors(output, "/hamiltonian/eigenvectors", list(r.eigenvectors), masks, group)
    return r

"""The multi-rank logic of the C host on ONE GPU: a loop-back communicator group (ls_amd_comm_create_local) puts P ranks into
this process, one per host thread; every collective rendezvouses through a barrier and moves its bytes with
device-to-device copies.  Same calls, same buffers, same counts as over RCCL -- set-up collectives (round agreement,
counts matrix), the double-buffered round pipeline of ls_amd_dist_matvec, the block / owner layouts of ls_amd_repl_matvec,
the PRIMME reductions -- only the transport differs (RCCL refuses two ranks on one device).  The reference tests its
multi-locale code the same way, by oversubscribing one machine."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from helpers import model_config, oracle_for, oracle_reps

pytestmark = pytest.mark.gpu


def _run_ranks(P, body):
    """body(rank, comm) on P threads; re-raises the first exception"""
    import distributed_matvec_amd as D

    comms = D.Communicator.local_group(P)
    errors = [None] * P

    def run(r):
        try:
            import torch

            torch.cuda.set_device(0)
            body(r, comms[r])
        except BaseException as e:  # noqa: BLE001
            import sys
            import traceback

            errors[r] = e
            print(f"[loop-back rank {r}] raised:", file=sys.stderr, flush=True)
            traceback.print_exc()

    import time

    # daemon threads: a rank stuck in a collective must not keep the interpreter alive at exit
    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(P)]
    for t in threads:
        t.start()
    deadline = time.monotonic() + 300  # for ALL ranks together
    while any(t.is_alive() for t in threads) and time.monotonic() < deadline:
        threads[0].join(timeout=0.2) if threads[0].is_alive() else time.sleep(0.2)
        if any(e is not None for e in errors):  # a rank that raised leaves its peers in their next collective: no point in waiting
            deadline = min(deadline, time.monotonic() + 10)
    for e in errors:  # the rank that failed first explains the others' wait
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    return comms


@pytest.mark.parametrize("name,P,cplx", [("heisenberg_chain_16", 2, False), ("heisenberg_chain_16", 3, False),
                                         ("heisenberg_chain_24_symm", 2, False), ("heisenberg_chain_24_symm", 4, False),
                                         ("heisenberg_kagome_16", 3, True), ("issue_01", 2, False), ("heisenberg_chain_10", 8, False)])
@pytest.mark.parametrize("mode", ["packets", "replicated"])
def test_ranks_as_threads(name, P, cplx, mode):
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator
    from oracle import c_oracle as CO

    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    if mode == "replicated" and not h.isHermitian:
        pytest.skip("replicated-x needs a Hermitian operator")
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    dtype = torch.complex128 if cplx else torch.float64
    xs = [D.fillRandom(reps[p], 17, dtype) for p in range(P)]
    ys = [torch.full_like(x, 6.0) for x in xs]
    dots = [None] * P

    def body(rank, comm):
        if mode == "packets":
            op = RcclDistributedOperator(h, reps[rank], dtype, comm=comm, num_rounds=3)
            assert op.num_rounds == 3
            # exchange operators on unprojected fixed-weight bases leave the producers as sorted streams (window consumers, no atomics)
            streams = name in ("heisenberg_chain_16", "heisenberg_kagome_16")
            assert op.engine.plan.kernel == ("tile+streams" if streams else "tile"), (name, op.engine.plan.kernel)
        else:
            op = RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm)
        op.matvec(xs[rank], ys[rank], check=True)
        op.matvec(xs[rank], ys[rank], check=True)  # twice: double-buffered slots, cursors, staging buffers reused
        dots[rank] = complex(op.dot(xs[rank], xs[rank]).cpu().item())
        op.dm.destroy() if mode == "packets" else op.rm.destroy()

    comms = _run_ranks(P, body)
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    assert np.array_equal(CO.hashed_to_block([r.cpu().numpy().view(np.uint64) for r in reps], keys), want_reps)
    x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
    got = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
    want = oracle_for(name).local_matvec(want_reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for d in dots:  # globalSumReal over the ranks
        assert abs(d - np.vdot(x, x)) < 1e-9
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("case", ["heisenberg_chain_20/4/f64/0", "heisenberg_chain_20/8/c128/2", "heisenberg_kagome_16/3/f64/0", "heisenberg_chain_16/2/c128/5"])
def test_packet_exchange_sorted_streams_against_atomic_consumers(monkeypatch, case):
    """ls_amd_dist_matvec with >= 2 ranks: the sorted packet streams (the stream starts of every rank's segments all-gathered once, the
    own partition's segment consumed out of the send buffer, one window launch per round) against the atomic consumers
    (LS_AMD_PACKET_STREAMS=0) and the oracle; the default number of rounds (<= 3) and forced ones; the wire carries the same bytes."""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclDistributedOperator
    from oracle import c_oracle as CO

    name, P, dt, rounds = case.split("/")
    P, rounds = int(P), int(rounds)
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    dtype = torch.complex128 if dt == "c128" else torch.float64
    xs = [D.fillRandom(reps[p], 19, dtype) for p in range(P)]
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
    want = oracle_for(name).local_matvec(want_reps, x)
    results, volumes = {}, {}
    from distributed_matvec_amd import _lib

    # ("retry": the set-up of the streams fails on ONE rank -- every rank must come back with the atomic consumers)
    for label, env in (("streams", None), ("atomics", "0"), ("retry", None)):
        if env is None:
            monkeypatch.delenv("LS_AMD_PACKET_STREAMS", raising=False)
        else:
            monkeypatch.setenv("LS_AMD_PACKET_STREAMS", env)
        ys = [torch.full_like(v, 2.5) for v in xs]
        info = [None] * P

        def body(rank, comm):
            _lib.load().ls_amd_test_fail_dist_streams(1 if label == "retry" and rank == P - 1 else 0)  # (thread-local: ranks are threads)
            try:
                op = RcclDistributedOperator(h, reps[rank], dtype, comm=comm, num_rounds=rounds)
            finally:
                _lib.load().ls_amd_test_fail_dist_streams(0)
            for _ in range(2):
                op.matvec(xs[rank], ys[rank], check=True)
            info[rank] = (op.engine.plan.kernel, op.num_rounds, op.exchange_bytes_per_matvec)
            op.dm.destroy()

        comms = _run_ranks(P, body)
        for c in comms:
            c.destroy()
        assert all(k == ("tile+streams" if label == "streams" else "tile") for k, _, _ in info), info
        assert len({r for _, r, _ in info}) == 1 and (info[0][1] == rounds if rounds else info[0][1] <= 3)
        volumes[label] = sum(b for _, _, b in info)
        results[label] = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
        assert np.abs(results[label] - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), label
    assert volumes["streams"] == volumes["atomics"] > 0  # pre-indexed 12 / 20-byte packets either way; the own segment never travels
    assert np.abs(results["streams"] - results["atomics"]).max() <= 1e-13 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("case", ["heisenberg_chain_20/4/f64/-7", "heisenberg_chain_20/8/c128/-6", "heisenberg_chain_16/3/f64/-5",
                                  "heisenberg_kagome_16/4/f64/-6", "heisenberg_chain_24/8/f64/-10", "heisenberg_chain_24/8/f64/10",
                                  "heisenberg_chain_24/2/f64/0"])
def test_replicated_exchange_sends_only_what_the_rows_reach(monkeypatch, case):
    """Unprojected bases in the replicated-x exchange: a rank's contiguous rows read their own neighbourhood and the partner
    blocks of the top bonds, not the whole vector.  The reach is found at plan time (every off-diagonal partner of every row,
    marked per block of 2^s rows), kept as <= 16 intervals, and each owner sends the contiguous piece of its block that falls into
    each interval -- one grouped exchange, no packing; the permutation into global order runs on the intervals only.  Result
    == the oracle; the loop-back transport checks that every sender's byte counts are what the receiver computed; fewer bytes
    arrive than N - N/P elements.  (-s forces the layout on bases too small for the 80 % rule, s applies the rule, 0 = off.)"""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclReplicatedOperator
    from oracle import c_oracle as CO

    name, P, dt, reach = case.split("/")
    P = int(P)
    monkeypatch.setenv("LS_AMD_REPL_REACH", reach)
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    dtype = torch.complex128 if dt == "c128" else torch.float64
    xs = [D.fillRandom(reps[p], 31, dtype) for p in range(P)]
    others = [D.fillRandom(reps[p], 32, dtype) for p in range(P)]
    ys = [torch.full_like(x, 4.0) for x in xs]
    x_in = [None] * P

    def body(rank, comm):
        op = RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm)
        x_in[rank] = op.x_bytes_in
        op.matvec(others[rank], ys[rank], check=True)  # stale values in the unread parts of the global buffer must not matter
        op.matvec(xs[rank], ys[rank], check=True)
        op.rm.destroy()

    comms = _run_ranks(P, body)
    n = int(masks.numel())
    es = 16 if dt == "c128" else 8
    full = [(n - int(reps[p].numel())) * es for p in range(P)]
    if int(reach) < 0:
        assert all(a <= b for a, b in zip(x_in, full)) and sum(x_in) < sum(full), (x_in, full)
    elif int(reach) == 0:
        assert x_in == full
    else:  # the 80 % rule decides, for all ranks alike
        assert x_in == full or all(a < b for a, b in zip(x_in, full)), (x_in, full)
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
    got = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
    want = oracle_for(name).local_matvec(want_reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("indexed", ["1", "0", "fused", "part", "cached", "cached-part"])
@pytest.mark.parametrize("case", ["heisenberg_chain_24_symm/4/f64", "heisenberg_chain_24_symm/3/c128", "issue_01/2/f64",
                                  "heisenberg_kagome_12_symm/8/f64", "translation_12_5/3/c128"])
def test_replicated_exchange_indexed_and_value_table(monkeypatch, case, indexed):
    """ls_amd_repl_matvec on projected bases, both ways of reading x: INDEXED (default: static {rep -> slot} table, x stays
    in the owner-major order it arrives in, owners prescale -- no per-rank O(N) pass) and the value table + permutation pass
    (LS_AMD_REPL_INDEXED=0), against the oracle; trivial sectors (prescaled), a -1 character and complex characters.
    The indexed mode runs as resolve (on the compute stream, while the blocks of x are exchanged on the communicator's
    stream) + gather by default; "fused": one kernel after the exchange (LS_AMD_PULL_SPLIT=0); "part": a packet buffer that
    only holds the first rows -- the others take the fused kernel."""
    if indexed == "fused":
        monkeypatch.setenv("LS_AMD_PULL_SPLIT", "0")
        indexed = "1"
    elif indexed == "part":
        monkeypatch.setenv("LS_AMD_PULL_SPLIT", "70000")
        indexed = "1"
    cached = indexed.startswith("cached")  # slot cache: resolve once, later matvecs gather only ("-part": a buffer for some rows)
    if cached:
        if indexed == "cached-part":
            monkeypatch.setenv("LS_AMD_PULL_SPLIT", "70000")
        indexed = "1"
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclReplicatedOperator
    from helpers import complex_translation_config
    from oracle import c_oracle as CO
    from oracle import model as M

    name, P, dt = case.split("/")
    P = int(P)
    monkeypatch.setenv("LS_AMD_REPL_INDEXED", indexed)
    if name.startswith("translation"):
        _, L, sector = name.split("_")
        cfg = complex_translation_config(int(L), int(sector))
        o = CO.COracle(M.model_from_config(cfg))
        want_reps = o.enumerate()
    else:
        cfg, o, want_reps = model_config(name), oracle_for(name), oracle_reps(name)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    dtype = torch.complex128 if dt == "c128" else torch.float64
    xs = [D.fillRandom(reps[p], 23, dtype) for p in range(P)]
    ys = [torch.full_like(x, -3.0) for x in xs]
    kernels = [None] * P

    others = [D.fillRandom(reps[p], 77, dtype) for p in range(P)]

    def body(rank, comm):
        op = RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm)
        if cached:
            assert op.engine.plan.cache_slots(70000 if "LS_AMD_PULL_SPLIT" in os.environ else 0) > 0
            op.matvec(others[rank], ys[rank], check=True)  # resolves; the checked calls below reuse the streams on another x
        kernels[rank] = op.engine.plan.kernel
        for _ in range(2):
            op.matvec(xs[rank], ys[rank], check=True)
        op.rm.destroy()

    comms = _run_ranks(P, body)
    want_kernel = ("replicated-tile-pull+indexed" if indexed == "1" else "replicated-tile-pull") + ("+cached" if cached else "")
    assert all(k == want_kernel for k in kernels), kernels
    keys = CO.locale_idx_of(want_reps, P)
    x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
    got = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
    want = o.local_matvec(want_reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for c in comms:
        c.destroy()


def test_round_agreement_and_primme_reductions_across_ranks(monkeypatch):
    """every rank must run the same number of rounds (all-reduce MAX of the local counts), and PRIMME's host-buffer
    reductions (/root/reference/src/PRIMME.chpl:267-373) sum / broadcast across the ranks of primme->commInfo"""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib
    from distributed_matvec_amd.distributed import RcclDistributedOperator

    monkeypatch.setenv("LS_AMD_ROWS_PER_ROUND", "700")
    name, P = "heisenberg_chain_16", 3
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    rounds = [None] * P
    sums = [None] * P
    bcast = [None] * P
    L = _lib.load()

    class View(C.Structure):  # just enough of primme_params for the callbacks: see include/ls_chpl.h
        pass

    def body(rank, comm):
        op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm)
        rounds[rank] = op.num_rounds
        # the reductions take the communicator from primme->commInfo: build a zeroed params view and set that field
        buf = (C.c_char * 4096)()
        off = L.ls_amd_test_primme_comminfo_offset()
        C.c_void_p.from_buffer(buf, off).value = comm.h.value
        C.c_int.from_buffer(buf, L.ls_amd_test_primme_sumtype_offset()).value = 3  # primme_op_double
        send = np.array([1.0 + rank, -2.0 * rank, 0.5])
        recv = np.zeros(3)
        n, ierr = C.c_int(3), C.c_int(7)
        L.primmeGlobalSumReal(send.ctypes.data, recv.ctypes.data, C.byref(n), buf, C.byref(ierr))
        assert ierr.value == 0
        sums[rank] = recv.copy()
        b = np.array([10.0 + rank, 20.0 + rank])
        n2 = C.c_int(2)
        L.primmeBroadcastReal(b.ctypes.data, C.byref(n2), buf, C.byref(ierr))
        assert ierr.value == 0
        bcast[rank] = b.copy()
        op.dm.destroy()

    comms = _run_ranks(P, body)
    want_rounds = max(-(-int(r.numel()) // 700) for r in reps)
    assert rounds == [want_rounds] * P
    for s in sums:
        assert np.allclose(s, [1 + 2 + 3, -2.0 * (0 + 1 + 2), 1.5])
    for b in bcast:
        assert np.array_equal(b, [10.0, 20.0])
    for c in comms:
        c.destroy()


def test_primme_matvec_callback_across_ranks():
    """ls_chpl_primme_matvec (/root/reference/src/Diagonalize.chpl:134-162) with one locale per rank: every rank's basis
    holds ITS block of the hashed representatives, x and y are the matching host blocks (ldx > n, two columns), the
    communicator comes from primme->commInfo; the callback runs on all ranks in lock-step."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib
    from oracle import c_oracle as CO

    name, P = "heisenberg_chain_16", 3
    L = _lib.load()
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    parts = CO.block_to_hashed(want_reps, keys, P)
    rs = np.random.RandomState(3)
    x_block = [rs.rand(len(want_reps)) - 0.5 for _ in range(2)]
    x_parts = [CO.block_to_hashed(v, keys, P) for v in x_block]
    y_parts = [[None] * P for _ in range(2)]

    def body(rank, comm):
        basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)  # every rank has its own objects
        basis.uncheckedSetRepresentatives(parts[rank])
        n = len(parts[rank])
        ld = n + 5
        X = np.zeros((2, ld)); Y = np.full((2, ld), 3.0)
        for k in range(2):
            X[k, :n] = x_parts[k][rank]
        buf = (C.c_char * 4096)()
        C.c_void_p.from_buffer(buf, L.ls_amd_test_primme_comminfo_offset()).value = comm.h.value
        C.c_int64.from_buffer(buf, L.ls_amd_test_primme_nlocal_offset()).value = n
        C.c_void_p.from_buffer(buf, L.ls_amd_test_primme_matrix_offset()).value = C.cast(h.payload, C.c_void_p).value
        ldx, ldy, bs, ierr = C.c_int64(ld), C.c_int64(ld), C.c_int(2), C.c_int(5)
        L.ls_chpl_primme_matvec(X.ctypes.data, C.byref(ldx), Y.ctypes.data, C.byref(ldy), C.byref(bs), buf, C.byref(ierr))
        _lib.raise_pending_halt()
        assert ierr.value == 0
        assert np.all(Y[:, n:] == 3.0)  # padding untouched
        for k in range(2):
            y_parts[k][rank] = Y[k, :n].copy()
        del h, basis

    comms = _run_ranks(P, body)
    o = oracle_for(name)
    for k in range(2):
        got = CO.hashed_to_block(y_parts[k], keys)
        want = o.local_matvec(want_reps, x_block[k])
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("mode", ["packets", "replicated"])
def test_operator_without_diagonal_accumulates_across_ranks(mode):
    """DMV:1062-1063: without diagonal terms y is not assigned first, so y += H x -- also when the rows of H x arrive from
    other ranks (packets: atomics into y; replicated: the returned pieces are added to y)"""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator
    from oracle import c_oracle as CO
    from oracle import model as M

    name, P = "heisenberg_chain_16", 3
    cfg = model_config(name)
    cfg2 = {"basis": cfg["basis"], "hamiltonian": {"terms": [t for t in cfg["hamiltonian"]["terms"] if "ᶻ" not in t["expression"]]}}
    basis, h = D.loadConfigFromDict(cfg2, hamiltonian=True)
    assert h.numberDiagTerms() == 0
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    xs = [D.fillRandom(reps[p], 23, torch.float64) for p in range(P)]
    ys = [torch.full_like(x, 4.0) for x in xs]

    def body(rank, comm):
        op = (RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm, num_rounds=2) if mode == "packets"
              else RcclReplicatedOperator(h, reps_global, masks, torch.float64, comm=comm))
        op.matvec(xs[rank], ys[rank], check=True)
        op.dm.destroy() if mode == "packets" else op.rm.destroy()

    for c in _run_ranks(P, body):
        c.destroy()
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
    got = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
    want = CO.COracle(M.model_from_config(cfg2)).local_matvec(want_reps, x, y=np.full(len(want_reps), 4.0))
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("name,P,cplx", [("heisenberg_chain_20", 4, False), ("heisenberg_chain_24_symm", 3, False),
                                         ("heisenberg_chain_16", 2, True)])
@pytest.mark.parametrize("mode", ["packets", "replicated", "replicated-chunked"])
def test_self_verification_catches_a_misplaced_segment(monkeypatch, name, P, cplx, mode):
    """The check `bench.py --gpus N` attaches to every exchange strategy (distributed-matvec_amd/verify.py; the reference's
    multi-locale check, test/TestMatrixVectorProduct.chpl:41-59): every rank's block of y against its rows of the ONE-partition
    kernel on x = u(hash(sigma, seed)), element-wise, plus all-reduced invariants.  Clean run: ok, error <= 1e-12.  Then one
    segment offset of the exchange layout is shifted by one element (ls_amd_test_corrupt_*: the exchange still completes, the
    data is misplaced) and the same check must say so -- on every rank."""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import verify
    from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator

    if mode == "replicated-chunked":  # the chunked return of the projected bases' driver (forced: the default chunks >= 2^16 rows)
        if "symm" not in name:
            pytest.skip("the chunked return belongs to the indexed (projected) driver")
        monkeypatch.setenv("LS_AMD_REPL_RETURN_CHUNKS", "3")
        mode = "replicated"
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    dtype = torch.complex128 if cplx else torch.float64
    refs = [verify.reference_block(h, reps_global, masks, p, dtype) for p in range(P)]
    clean, faulty, injected = [None] * P, [None] * P, [None] * P

    def body(rank, comm):
        def allsum(v):
            t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
            comm.allreduce_sum(t)
            return float(t.item())

        def allmax(v):  # (the loop-back all-reduce only carries sums and i64 maxima: scale into an integer)
            t = torch.tensor([int(v * 2.0**40)], dtype=torch.int64, device="cuda")
            comm.allreduce_max(t)
            return float(t.item()) / 2.0**40

        op = (RcclDistributedOperator(h, reps[rank], dtype, comm=comm, num_rounds=2) if mode == "packets"
              else RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm))
        if mode == "packets":  # one partition per process: pre-indexed packets wherever a closed form ranks the basis
            assert op.engine.plan.key_bytes == (8 if "symm" in name else 4)
        x = D.fillRandom(reps[rank], 42, dtype)
        y = torch.zeros_like(x)
        xr, yr, ymax, kern = refs[rank]
        assert torch.equal(x, xr)
        op.matvec(x, y, check=True)
        clean[rank] = verify.parity_object(y, x, yr, ymax, allsum=allsum, allmax=allmax, reference_kernel=kern)
        injected[rank] = allsum(1.0 if op.inject_fault() else 0.0)
        y.zero_()
        op.matvec(x, y, check=False)
        faulty[rank] = verify.parity_object(y, x, yr, ymax, allsum=allsum, allmax=allmax, reference_kernel=kern)
        op.dm.destroy() if mode == "packets" else op.rm.destroy()

    comms = _run_ranks(P, body)
    for r in range(P):
        assert clean[r]["ok"] and clean[r]["max_rel_err"] <= 1e-12 and clean[r]["rows_off"] == 0, clean[r]
        assert injected[r] >= 1, "no rank had a segment to corrupt"
        assert not faulty[r]["ok"], faulty[r]            # every rank learns about it (the reductions are collective)
        assert faulty[r]["rows_off"] > 0 and faulty[r]["max_rel_err"] > 1e-6
    for c in comms:
        c.destroy()


# ---- never hang (VERDICT r5 #1b): mismatched exchange layouts end with an error on EVERY rank, quickly -------------------------

@pytest.mark.parametrize("mode", ["packets", "replicated"])
@pytest.mark.parametrize("name,P", [("heisenberg_chain_16", 3), ("heisenberg_chain_24_symm", 2)])
def test_layout_mismatch_is_caught_at_setup_on_every_rank(name, P, mode):
    """One rank announces 8 bytes less for its right neighbour than that neighbour expects (ls_amd_test_skew_exchange, late = 0).
    The set-up cross-check (all-gather of every rank's send AND receive counts, every pair verified on every rank) must fail
    ls_amd_dist_create / ls_amd_repl_create on ALL ranks, name the two ranks and both byte counts, and start no exchange."""
    import time

    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib
    from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator

    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    messages = [None] * P
    L = _lib.load()
    L.ls_amd_test_skew_exchange(1, -8, 0)
    t0 = time.monotonic()
    try:
        def body(rank, comm):
            try:
                if mode == "packets":
                    RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm, num_rounds=2)
                else:
                    RcclReplicatedOperator(h, reps_global, masks, torch.float64, comm=comm)
            except D.LsAmdError as e:
                messages[rank] = str(e)

        comms = _run_ranks(P, body)
    finally:
        L.ls_amd_test_skew_exchange(-1, 0, 0)
    assert time.monotonic() - t0 < 30
    for r, m in enumerate(messages):
        assert m is not None, f"rank {r} built its operator on a layout its peers contradict"
        assert "exchange layouts disagree" in m and f"rank 1 sends" in m and f"to rank {2 % P}" in m and "which expects" in m, m
    # the same group still works afterwards: nothing was left half-open
    ys = [None] * P

    def body2(rank, comm):
        op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm, num_rounds=2)
        x = D.fillRandom(reps[rank], 3, torch.float64)
        ys[rank] = torch.zeros_like(x)
        op.matvec(x, ys[rank], check=True)
        op.dm.destroy()

    for c in comms:
        c.destroy()
    comms = _run_ranks(P, body2)
    for c in comms:
        c.destroy()


def test_run_time_mismatch_ends_with_an_error_not_a_hang(monkeypatch):
    """A fault AFTER the set-up check (late = 1): one rank's send segment shrinks by 8 bytes.  The loop-back transport cross-checks
    every exchange (the receiver sees the announced count), the receiver fails, and its peers -- who would wait for it in their
    next rendezvous for ever -- give up at the deadline (LS_AMD_COMM_WATCHDOG_S): every rank ends with rc != 0 in well under 30 s."""
    import time

    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib
    from distributed_matvec_amd.distributed import RcclDistributedOperator

    monkeypatch.setenv("LS_AMD_COMM_WATCHDOG_S", "3")
    name, P = "heisenberg_chain_16", 3
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, _masks = D.enumerateStates(basis, P)
    L = _lib.load()
    outcome = [None] * P
    L.ls_amd_test_skew_exchange(0, -8, 1)
    t0 = time.monotonic()
    try:
        def body(rank, comm):
            op = RcclDistributedOperator(h, reps[rank], torch.float64, comm=comm, num_rounds=2)
            x = D.fillRandom(reps[rank], 3, torch.float64)
            y = torch.zeros_like(x)
            try:
                for _ in range(3):
                    op.matvec(x, y, check=True)
                outcome[rank] = "completed"
            except D.LsAmdError as e:
                outcome[rank] = str(e)

        _run_ranks(P, body)
    finally:
        L.ls_amd_test_skew_exchange(-1, 0, 0)
    took = time.monotonic() - t0
    assert took < 30, took
    assert all(o is not None and o != "completed" for o in outcome), outcome
    assert any("sends" in o and "which expects" in o for o in outcome), outcome          # the receiver names the mismatch
    assert any("gave up" in o and "ranks arrived" in o for o in outcome), outcome          # its peers hit the deadline
    # (the group is broken for good: its communicators are abandoned, not destroyed -- a destroy would synchronise streams)


def test_rccl_watchdog_ends_a_stalled_exchange(tmp_path):
    """The watchdog thread of an RCCL communicator: an exchange-stream event that has not completed by the deadline ends the process
    with exit code 86 and says which rank, what was in flight and for how long.  (One RCCL rank on this box: the stall is a host
    callback on the exchange stream, armed exactly like an exchange -- lsk_comm_test_stall.)  ls_amd_comm_wait, the polite form,
    returns an error for the same stall instead of hanging in a synchronisation."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import distributed_matvec_amd as D
from distributed_matvec_amd import _lib
torch.cuda.set_device(0)
mode = sys.argv[1]
comm = D.Communicator(1, 0, D.Communicator.unique_id())
L = _lib.load()
assert L.ls_amd_comm_test_stall(comm.h, 12.0) == 0
if mode == "polite":
    t0 = time.monotonic()
    try:
        comm.wait(2.0)
    except D.LsAmdError as e:
        print("POLITE", round(time.monotonic() - t0, 1), str(e), flush=True)
        os._exit(0)
    print("NO ERROR", flush=True)
    os._exit(1)
torch.cuda.synchronize()   # what a hung job does: sits in a synchronisation
print("SURVIVED", flush=True)
''' % root
    env = dict(os.environ, LS_AMD_COMM_WATCHDOG_S="3")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code, "watchdog"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 86, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "libls_amd watchdog: rank 0 of 1" in p.stderr and "not complete" in p.stderr and "test stall" in p.stderr, p.stderr[-1500:]
    assert "SURVIVED" not in p.stdout
    env["LS_AMD_COMM_WATCHDOG_S"] = "60"
    p = subprocess.run([sys.executable, "-c", code, "polite"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0 and "POLITE" in p.stdout and "has not drained" in p.stdout, (p.returncode, p.stdout[-800:], p.stderr[-800:])


# ---- replicated x, projected bases: chunked return of y + adaptive split (VERDICT r5 #1a) ------------------------------------------

@pytest.mark.parametrize("name,P,cplx", [("heisenberg_chain_24_symm", 4, False), ("heisenberg_chain_24_symm", 3, True),
                                         ("heisenberg_kagome_12_symm", 2, False)])
@pytest.mark.parametrize("chunks,adapt", [("0", "0"), ("3", "1"), ("8", "1"), ("2", "0")])
def test_replicated_chunked_return_and_adaptive_split(monkeypatch, name, P, cplx, chunks, adapt):
    """ls_amd_repl_matvec on a projected basis: the rows of a rank are computed in chunks and every chunk's rows travel back to their
    owners while the next chunk is gathered (LS_AMD_REPL_RETURN_CHUNKS; forced here -- the default only chunks >= 2^16 rows), and
    only as many rows are resolved ahead as the exchange of x takes, the rest running fused afterwards (LS_AMD_REPL_ADAPT; the
    number changes from matvec to matvec: six matvecs in a row must all equal the oracle).  One return for all rows and no
    adaptation ("0", "0") is the round-5 path."""
    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib
    from distributed_matvec_amd.distributed import RcclReplicatedOperator
    from oracle import c_oracle as CO

    monkeypatch.setenv("LS_AMD_REPL_RETURN_CHUNKS", chunks)
    monkeypatch.setenv("LS_AMD_REPL_ADAPT", adapt)
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, P)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    dtype = torch.complex128 if cplx else torch.float64
    want_reps = oracle_reps(name)
    keys = CO.locale_idx_of(want_reps, P)
    L = _lib.load()
    active = [[] for _ in range(P)]
    results = []
    for it in range(6):
        xs = [D.fillRandom(reps[p], 100 + it, dtype) for p in range(P)]
        results.append((xs, [torch.full_like(v, -3.0) for v in xs]))

    def body(rank, comm):
        op = RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm)
        assert op.engine.plan.kernel == "replicated-tile-pull+indexed"
        for xs, ys in results:
            op.matvec(xs[rank], ys[rank], check=True)
            active[rank].append(int(L.ls_amd_internal_plan_split_active(op.engine.plan.h)))
        op.rm.destroy()

    comms = _run_ranks(P, body)
    for xs, ys in results:
        x = CO.hashed_to_block([v.cpu().numpy() for v in xs], keys)
        got = CO.hashed_to_block([v.cpu().numpy() for v in ys], keys)
        want = oracle_for(name).local_matvec(want_reps, x)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for r in range(P):  # whole 256-row tiles, or every row of the rank
        n_r = len(want_reps) * (r + 1) // P - len(want_reps) * r // P
        assert all(a == n_r or (a % 256 == 0 and 0 < a < n_r) for a in active[r]), (active[r], n_r)
        if adapt == "0":
            assert all(a == n_r for a in active[r])
    for c in comms:
        c.destroy()

"""The exchange inside the C host (ls_amd_comm / ls_amd_dist, RCCL loaded with dlopen; include/ls_amd.h).

On the one-GPU box RCCL accepts a single rank per device, so here the communicator has one rank: that still runs
the whole native path -- dlopen + ncclCommInitRank, the set-up collectives (all-reduce of the local counts,
all-gather of the counts matrix), the double-buffered round pipeline on two streams, the PRIMME reductions.
With >= 2 visible GPUs the same test body runs with one process per GPU (backend-free: the unique id travels
through a file, as a C caller would do it with MPI_Bcast)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_comm_single_rank_collectives():
    import torch

    import distributed_matvec_amd as D

    comm = D.Communicator(1, 0, D.Communicator.unique_id())
    t = torch.arange(5, dtype=torch.float64, device="cuda")
    comm.allreduce_sum(t)
    assert torch.equal(t.cpu(), torch.arange(5, dtype=torch.float64))
    m = torch.tensor([7, -3], dtype=torch.int64, device="cuda")
    comm.allreduce_max(m)
    assert m.tolist() == [7, -3]
    b = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
    comm.broadcast(b, 0)
    assert b.tolist() == [1.5, 2.5]
    comm.destroy()


@pytest.mark.parametrize("name,cplx", [("heisenberg_chain_16", False), ("heisenberg_chain_24_symm", False),
                                       ("heisenberg_kagome_16", True), ("heisenberg_chain_10", False)])
def test_dist_matvec_one_rank_vs_oracle(name, cplx, monkeypatch):
    """numLocales == 1 through the per-rank packet path (no test hook): 3 rounds, so the double-buffered
    slots are reused; y starts dirty; called twice."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclDistributedOperator
    from helpers import model_config, oracle_for, oracle_reps

    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    want_reps = oracle_reps(name)
    assert np.array_equal(reps[0].cpu().numpy().view(np.uint64), want_reps)
    dtype = torch.complex128 if cplx else torch.float64
    x = D.fillRandom(reps[0], 11, dtype)
    y = torch.full_like(x, 5.0)
    op = RcclDistributedOperator(h, reps[0], dtype, comm=D.Communicator(1, 0, D.Communicator.unique_id()), num_rounds=3)
    assert op.num_rounds == 3 and op.engine.plan.kernel == "tile"
    op.matvec(x, y, check=True)
    y.fill_(-2.0)
    op.matvec(x, y, check=True)
    want = oracle_for(name).local_matvec(want_reps, x.cpu().numpy())
    assert np.abs(y.cpu().numpy() - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    # automatic round agreement (all-reduce of the local counts)
    monkeypatch.setenv("LS_AMD_ROWS_PER_ROUND", "1000")
    op2 = RcclDistributedOperator(h, reps[0], dtype, comm=op.comm)
    assert op2.num_rounds == max(1, -(-len(want_reps) // 1000))
    y.zero_()
    op2.matvec(x, y, check=True)
    assert np.abs(y.cpu().numpy() - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    d = op.dot(x, x)
    assert abs(complex(d.cpu().item()) - np.vdot(x.cpu().numpy(), x.cpu().numpy())) < 1e-9


def test_primme_reductions_single_rank_and_default_comm():
    """primmeGlobalSumReal / primmeBroadcastReal (/root/reference/src/PRIMME.chpl:267-373): with one locale the sum
    is a copy (aliasing allowed), with a default communicator installed the same host buffers go through RCCL."""
    import distributed_matvec_amd as D
    from distributed_matvec_amd import _lib

    L = _lib.load()
    send = np.array([1.0, -2.0, 3.5])
    recv = np.zeros(3)
    n, ierr = C.c_int(3), C.c_int(99)
    L.primmeGlobalSumReal(send.ctypes.data, recv.ctypes.data, C.byref(n), None, C.byref(ierr))
    assert ierr.value == 0 and np.array_equal(recv, send)
    comm = D.Communicator(1, 0, D.Communicator.unique_id())
    comm.set_default()
    assert L.ls_amd_default_comm() == comm.h.value
    recv[:] = 0
    L.primmeGlobalSumReal(send.ctypes.data, recv.ctypes.data, C.byref(n), None, C.byref(ierr))
    assert ierr.value == 0 and np.array_equal(recv, send)
    L.primmeBroadcastReal(recv.ctypes.data, C.byref(n), None, C.byref(ierr))
    assert ierr.value == 0 and np.array_equal(recv, send)
    comm.destroy()
    assert not L.ls_amd_default_comm()


def _worker(rank, world, id_path, name, cplx, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time

    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclDistributedOperator
    from helpers import model_config

    torch.cuda.set_device(rank)
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(D.Communicator.unique_id())
        os.replace(id_path + ".tmp", id_path)
    while not os.path.exists(id_path):
        time.sleep(0.05)
    with open(id_path, "rb") as f:
        uid = f.read()
    comm = D.Communicator(world, rank, uid)
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, world)
    my_reps = reps[rank].clone()
    dtype = torch.complex128 if cplx else torch.float64
    x = D.fillRandom(my_reps, 7, dtype)
    y = torch.full_like(x, 9.0)
    op = RcclDistributedOperator(h, my_reps, dtype, comm=comm, num_rounds=3)
    op.matvec(x, y, check=True)
    op.matvec(x, y, check=True)
    nrm = op.dot(x, x)
    np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array([complex(nrm.cpu().item())]))
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x.cpu().numpy())
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y.cpu().numpy())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), my_reps.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("name,world,cplx", [("heisenberg_chain_16", 2, False), ("heisenberg_chain_24_symm", 2, False),
                                             ("heisenberg_kagome_16", 3, True)])
def test_dist_matvec_multi_gpu_rccl(tmp_path, name, world, cplx):
    """one process per GPU, packets over RCCL (no torch.distributed anywhere); skipped on the one-GPU box"""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses two ranks on one device)")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_for, oracle_reps
    from oracle import c_oracle as CO

    mp.spawn(_worker, args=(world, str(tmp_path / "uid"), name, cplx, str(tmp_path)), nprocs=world, join=True)
    reps = oracle_reps(name)
    keys = CO.locale_idx_of(reps, world)
    load = lambda k: [np.load(os.path.join(str(tmp_path), f"{k}{r}.npy")) for r in range(world)]  # noqa: E731
    assert np.array_equal(CO.hashed_to_block(load("r"), keys), reps)
    x = CO.hashed_to_block(load("x"), keys)
    got = CO.hashed_to_block(load("y"), keys)
    want = oracle_for(name).local_matvec(reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    for v in load("n"):
        assert abs(v[0] - np.vdot(x, x)) < 1e-9


@pytest.mark.parametrize("name,cplx", [("heisenberg_chain_16", False), ("heisenberg_chain_20", True), ("heisenberg_chain_24_symm", False),
                                       ("heisenberg_kagome_16", True), ("issue_01", False)])
def test_repl_matvec_one_rank_vs_oracle(name, cplx):
    """the replicated-x exchange of the C host (ls_amd_repl_*) with a communicator of one rank: permutation tables from
    masks, pull plan over the contiguous row range, owner regrouping of y; y starts dirty; called twice"""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclReplicatedOperator
    from helpers import model_config, oracle_for, oracle_reps

    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    want_reps = oracle_reps(name)
    dtype = torch.complex128 if cplx else torch.float64
    x = D.fillRandom(reps[0], 13, dtype)
    y = torch.full_like(x, 4.0)
    op = RcclReplicatedOperator(h, reps[0], masks, dtype, comm=D.Communicator(1, 0, D.Communicator.unique_id()))
    assert op.engine.plan.kernel.startswith("replicated-")
    op.matvec(x, y, check=True)
    y2 = torch.full_like(x, -9.0)
    op.matvec(x, y2, check=True)
    want = oracle_for(name).local_matvec(want_reps, x.cpu().numpy())
    assert np.abs(y.cpu().numpy() - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    # (bitwise equality is not promised: the projected-basis kernel accumulates with LDS atomics in arrival order)
    assert float((y - y2).abs().max()) <= 1e-12 * max(1.0, float(y.abs().max()))
    # an operator without diagonal terms: y is accumulated into (DMV:1062-1063)
    if name in ("heisenberg_chain_16", "heisenberg_chain_24_symm"):
        from oracle import c_oracle as CO
        from oracle import model as M

        cfg = model_config(name)
        cfg2 = {"basis": cfg["basis"], "hamiltonian": {"terms": [t for t in cfg["hamiltonian"]["terms"] if "ᶻ" not in t["expression"]]}}
        basis2, h2 = D.loadConfigFromDict(cfg2, hamiltonian=True)
        assert h2.numberDiagTerms() == 0
        op2 = RcclReplicatedOperator(h2, reps[0], masks, dtype, comm=op.comm)
        y3 = torch.full_like(x, 4.0)
        op2.matvec(x, y3, check=True)
        want2 = CO.COracle(M.model_from_config(cfg2)).local_matvec(want_reps, x.cpu().numpy(), y=np.full(len(want_reps), 4.0))
        assert np.abs(y3.cpu().numpy() - want2).max() <= 1e-12 * max(1.0, np.abs(want2).max())


def _repl_worker(rank, world, id_path, name, cplx, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time

    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import RcclReplicatedOperator
    from helpers import model_config

    torch.cuda.set_device(rank)
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(D.Communicator.unique_id())
        os.replace(id_path + ".tmp", id_path)
    while not os.path.exists(id_path):
        time.sleep(0.05)
    with open(id_path, "rb") as f:
        uid = f.read()
    comm = D.Communicator(world, rank, uid)
    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    reps, masks = D.enumerateStates(basis, world)
    reps_global = D.arrFromHashedToBlock(reps, masks)
    my_reps = reps[rank].clone()
    dtype = torch.complex128 if cplx else torch.float64
    x = D.fillRandom(my_reps, 7, dtype)
    y = torch.full_like(x, 9.0)
    op = RcclReplicatedOperator(h, reps_global, masks, dtype, comm=comm)
    op.matvec(x, y, check=True)
    op.matvec(x, y, check=True)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x.cpu().numpy())
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y.cpu().numpy())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), my_reps.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("name,world,cplx", [("heisenberg_chain_16", 2, False), ("heisenberg_chain_24_symm", 2, False),
                                             ("heisenberg_kagome_16", 3, True)])
def test_repl_matvec_multi_gpu_rccl(tmp_path, name, world, cplx):
    """one process per GPU, replicated-x exchange over RCCL inside the C host; skipped on the one-GPU box"""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses two ranks on one device)")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_for, oracle_reps
    from oracle import c_oracle as CO

    mp.spawn(_repl_worker, args=(world, str(tmp_path / "uid"), name, cplx, str(tmp_path)), nprocs=world, join=True)
    reps = oracle_reps(name)
    keys = CO.locale_idx_of(reps, world)
    load = lambda k: [np.load(os.path.join(str(tmp_path), f"{k}{r}.npy")) for r in range(world)]  # noqa: E731
    assert np.array_equal(CO.hashed_to_block(load("r"), keys), reps)
    x = CO.hashed_to_block(load("x"), keys)
    got = CO.hashed_to_block(load("y"), keys)
    want = oracle_for(name).local_matvec(reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def _run_bench(*extra):
    import json
    import subprocess

    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-distributed", "--model", "heisenberg_chain_24",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", *extra],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stderr[-2000:]
    return p.returncode, json.loads(lines[-1]), p.stderr


def test_bench_distributed_path_is_self_verifying():
    """`bench.py --gpus N` (here: the same code path with one RCCL rank) reports the communicator size as RCCL gives it and, for
    EVERY exchange strategy, y against the one-partition kernel; a deliberately misplaced segment makes the run fail.  The
    driver's multi-GPU run is the only place RCCL's transport runs with > 1 rank: its number must carry this evidence."""
    rc, out, err = _run_bench()
    assert rc == 0, err[-2000:]
    assert out["rccl"]["comm_count"] == 1 and out["rccl"]["ranks_by_allreduce"] == 1 and out["rccl"]["world_size"] == 1
    assert set(out["exchanges"]) == {"packets", "replicated"} and not out["failed_exchanges"]
    for name, ex in out["exchanges"].items():
        par = ex["parity"]
        assert par["ok"] and par["max_rel_err"] <= 1e-12 and par["rows_off"] == 0, (name, par)
        for inv in par["invariants"].values():
            assert inv["rel_err"] <= 1e-12
    assert out["parity"]["ok"] and out["parity"]["failed"] == [] and len(out["parity"]["checked"]) == 2
    # one rank: the packet plan has no remote segment to corrupt, the replicated-x layout has (the own rows of y)
    rc, out, err = _run_bench("--inject-fault")
    assert rc != 0
    assert "PARITY FAILURE" in err
    assert not out["exchanges"]["replicated"]["parity_after_fault"]["ok"]
    assert "exchanges.replicated" in out["parity"]["fault_injection"]["detected"]
    assert out["parity"]["fault_injection"]["missed"] == []
    assert out["exchanges"]["replicated"]["parity"]["ok"]  # the check before the fault was clean


def _run_bench_one_gpu(*extra, model="heisenberg_chain_24"):
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stderr[-2000:]
    return p.returncode, json.loads(lines[-1]), p.stderr


@pytest.mark.parametrize("model", ["heisenberg_chain_24", "heisenberg_chain_24_symm"])
def test_bench_one_gpu_line_carries_its_own_parity(model):
    """The N = 1 line checks the y it times (the reference's single-locale check, test/TestMatrixVectorProduct.chpl:25-39):
    the measured kernel against kernels that share no device code with it -- generic pull + push with atomics (unprojected), push +
    value-table pull (projected) --, element by element; a row kernel that skips ONE row (--inject-fault) ends the run with rc 3."""
    rc, out, err = _run_bench_one_gpu(model=model)
    assert rc == 0, err[-2000:]
    par = out["parity"]
    assert par["ok"] and par["failed"] == [] and "main" in par["checked"], par
    main = par["main"]
    assert main["ok"] and main["rows_off"] == 0 and main["max_rel_err"] <= 1e-12 and len(main["against"]) == 2, main
    for o in main["against"].values():
        assert o["ok"] and o["independent_of_measured_kernel"], o
        for inv in o["invariants"].values():
            assert inv["rel_err"] <= 1e-12
    if model == "heisenberg_chain_24":  # the other dtype's leg (c128: the north star's) carries its own object
        legs = [k for k in par["checked"] if k.startswith("extra.c128/")]
        assert legs, par["checked"]
    rc, out, err = _run_bench_one_gpu("--no-extra", "--inject-fault", model=model)
    assert rc == 3 and "PARITY FAILURE" in err, (rc, err[-1500:])
    assert out["parity"]["main"]["ok"]  # clean before the fault ...
    fi = out["parity"]["fault_injection"]
    assert "main" in fi["detected"] and fi["missed"] == [] and fi["nothing_to_corrupt"] == [], fi

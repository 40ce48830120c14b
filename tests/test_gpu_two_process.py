"""Two (three) real processes driving the HIP kernels on the one GPU, exchanging over gloo (RCCL refuses
two ranks on one device: "Duplicate GPU detected").  Everything except the transport is the product
path: HipEngine / HipReplicatedEngine, plans with one partition per process, packet layout produced
by one process and scattered by another, the replicated-x block/permutation logic."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, cplx, mode, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import distributed_matvec_amd as D
    from distributed_matvec_amd.distributed import DistributedOperator, ReplicatedOperator
    from helpers import model_config

    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
        reps, masks = D.enumerateStates(basis, world)
        reps_global = D.arrFromHashedToBlock(reps, masks)
        my_reps = reps[rank].clone()
        dtype = torch.complex128 if cplx else torch.float64
        x = D.fillRandom(my_reps, 7, dtype)
        y = torch.full_like(x, 9.0)
        if mode == "packets":
            op = DistributedOperator(h, my_reps, dtype, num_rounds=3)
        else:
            op = ReplicatedOperator(h, my_reps, reps_global, masks, dtype)
        op.matvec(x, y, check=True)
        op.matvec(x, y, check=True)  # twice: buffers / cursors must be reusable
        if name == "heisenberg_chain_16" and not cplx:
            # the eigensolver driver on top of the distributed operator (dots = all_reduce, PRIMME's globalSumReal)
            from distributed_matvec_amd.diagonalize import RankOperator, lanczos_smallest

            r = lanczos_smallest(RankOperator(op, my_reps, dtype), num_evals=1, eps=1e-9)
            assert r.converged
            np.save(os.path.join(out_dir, f"e{rank}.npy"), np.array(r.eigenvalues))
        np.save(os.path.join(out_dir, f"x{rank}.npy"), x.cpu().numpy())
        np.save(os.path.join(out_dir, f"y{rank}.npy"), y.cpu().numpy())
        np.save(os.path.join(out_dir, f"r{rank}.npy"), my_reps.cpu().numpy().view(np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,cplx", [("heisenberg_chain_16", 2, False), ("heisenberg_chain_24_symm", 2, False),
                                             ("heisenberg_kagome_16", 3, True), ("issue_01", 2, False)])
@pytest.mark.parametrize("mode", ["packets", "replicated"])
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_processes_one_gpu(tmp_path, name, world, cplx, mode, backend):
    """backend "gloo": every rank on the one GPU, host-staged transport.  backend "nccl" (RCCL, one rank per GPU): the
    production transport of the torch drivers -- device tensors handed to all_to_all_single / batch_isend_irecv as they
    are, async_op in the depth-2 pipeline (num_rounds = 3: buffer reuse); needs as many GPUs as ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_for, oracle_reps
    from oracle import c_oracle as CO

    if backend == "nccl":
        import torch

        if torch.cuda.device_count() < world:
            pytest.skip(f"RCCL needs one GPU per rank ({world}); this box has {torch.cuda.device_count()}")
    port = 30100 + (os.getpid() % 300) + world + (7 if mode == "packets" else 0) + (13 if backend == "nccl" else 0)
    mp.spawn(_worker, args=(world, port, name, cplx, mode, str(tmp_path), backend), nprocs=world, join=True)
    reps = oracle_reps(name)
    keys = CO.locale_idx_of(reps, world)
    parts_r = [np.load(os.path.join(str(tmp_path), f"r{r}.npy")) for r in range(world)]
    assert np.array_equal(CO.hashed_to_block(parts_r, keys), reps)
    x = CO.hashed_to_block([np.load(os.path.join(str(tmp_path), f"x{r}.npy")) for r in range(world)], keys)
    got = CO.hashed_to_block([np.load(os.path.join(str(tmp_path), f"y{r}.npy")) for r in range(world)], keys)
    want = oracle_for(name).local_matvec(reps, x)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    if name == "heisenberg_chain_16" and not cplx:
        import scipy.sparse.linalg as spla

        o = oracle_for(name)
        A = spla.LinearOperator((len(reps), len(reps)), matvec=lambda v: o.local_matvec(reps, np.ascontiguousarray(v, dtype=np.float64)), dtype=np.float64)
        e0 = spla.eigsh(A, k=1, which="SA", tol=1e-10)[0][0]
        for r in range(world):
            assert abs(np.load(os.path.join(str(tmp_path), f"e{r}.npy"))[0] - e0) < 1e-6

#!/usr/bin/env python3
"""P ranks as host threads on ONE GPU over the loop-back transport (ls_amd_comm_create_local): the multi-rank logic of
the C host at full size -- partitioning, set-up collectives, buffers, the round pipeline, both exchange strategies -- with a
per-stage timing tree of rank 0.  NOT a scaling measurement: the ranks share one device, so the wall time is roughly the sum
of their work.  Checks the result against the one-partition kernel."""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import distributed_matvec_amd as D  # noqa: E402
from distributed_matvec_amd import config  # noqa: E402
from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32)
ap.add_argument("--symm", action="store_true")
ap.add_argument("--P", type=int, default=8)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--mode", default="both", choices=["packets", "replicated", "both"])
args = ap.parse_args()

basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(args.L, symm=args.symm), hamiltonian=True)
P = args.P
reps, masks = D.enumerateStates(basis, P)
reps_global = D.arrFromHashedToBlock(reps, masks)
n = int(masks.numel())
xs = [D.fillRandom(reps[p], 42, torch.float64) for p in range(P)]
x_block = D.arrFromHashedToBlock(xs, masks)
y_ref = torch.empty_like(x_block)
ref = D.MatvecPlan(h, [reps_global], torch.float64, mode="pull")
ref.matvec([x_block], [y_ref])
_e0, _e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_e0.record()
for _ in range(2):
    ref.matvec([x_block], [y_ref], check=False)
_e1.record()
torch.cuda.synchronize()
one_gpu_ms = _e0.elapsed_time(_e1) / 2
ref_kernel = ref.kernel
ref.destroy()
del x_block
print(f"chain_{args.L}{'_symm' if args.symm else ''}: N = {n}, P = {P} loop-back ranks on {torch.cuda.get_device_name(0)}; one-partition kernel "
      f"({ref_kernel}) {one_gpu_ms:.2f} ms per matvec in this job; LS_AMD_REPL_RETURN_CHUNKS={os.environ.get('LS_AMD_REPL_RETURN_CHUNKS', 'default')} "
      f"LS_AMD_REPL_ADAPT={os.environ.get('LS_AMD_REPL_ADAPT', 'default')}", flush=True)

for mode in (["packets", "replicated"] if args.mode == "both" else [args.mode]):
    comms = D.Communicator.local_group(P)
    ys = [torch.zeros_like(v) for v in xs]
    out = {}
    errors = []

    def body(rank):
        try:
            torch.cuda.set_device(0)
            op = (RcclDistributedOperator(h, reps[rank], torch.float64, comm=comms[rank]) if mode == "packets"
                  else RcclReplicatedOperator(h, reps_global, masks, torch.float64, comm=comms[rank]))
            op.matvec(xs[rank], ys[rank], check=True)
            plan = op.engine.plan
            plan.enable_stage_timing(1 << 16)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(args.steps):
                op.matvec(xs[rank], ys[rank], check=False)
            torch.cuda.synchronize()
            wall_r = (time.perf_counter() - t) / args.steps
            report = plan.timing_report()
            stages, mv = plan.stage_times()
            out[rank] = (wall_r, report, getattr(op, "exchange_bytes_per_matvec", 0), getattr(op, "num_rounds", 1),
                         {k: v[0] / max(1, mv) for k, v in stages.items()}, plan.kernel, getattr(op, "x_bytes_in", None))
            (op.dm if mode == "packets" else op.rm).destroy()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=body, args=(r,)) for r in range(P)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    if errors:
        raise errors[0]
    got = D.arrFromHashedToBlock(ys, masks)
    err = float((got - y_ref).abs().max()) / float(y_ref.abs().max())
    wall = max(v[0] for v in out.values())
    xb = sum(v[2] for v in out.values())
    print(f"[{mode}] {P} ranks sharing one GPU: {wall * 1e3:.2f} ms per matvec (all ranks) = {wall * 1e3 / one_gpu_ms:.3f} x the one-partition kernel, "
          f"rounds = {out[0][3]}, exchange {xb / 1e9:.2f} GB per matvec over all ranks, max |dy| / max |y| vs one partition = {err:.1e}")
    if out[0][6] is not None:
        es = 8
        full = [(int(masks.numel()) - int(reps[r].numel())) * es for r in range(P)]
        print("x received per rank [GB]: " + " ".join(f"{out[r][6] / 1e9:.3f}" for r in range(P)) +
              f"   (whole vector minus own block: {full[0] / 1e9:.3f}; share {sum(out[r][6] for r in range(P)) / max(1, sum(full)):.3f})")
    print("rank 0: " + out[0][1], flush=True)
    # device time of every rank's stages (HIP events; the ranks share ONE device, so stages of different ranks overlap and
    # each is slower than it would be alone): the aggregate is what P GPUs would have to do in total, transport excluded
    # ("exchangeWait" here is device-to-device copies behind a host barrier, not xGMI)
    names = list(out[0][4])
    print(f"per-rank stage device time per matvec [ms] ({out[0][5]}):")
    print("  rank  " + "  ".join(f"{n[:14]:>14s}" for n in names) + "     compute")
    agg = 0.0
    for r in range(P):
        st = out[r][4]
        compute = sum(v for k, v in st.items() if k != "exchangeWait")
        agg += compute
        print(f"  {r:4d}  " + "  ".join(f"{st[n]:14.3f}" for n in names) + f"  {compute:10.3f}")
    print(f"aggregate compute-stage device time of the {P} ranks: {agg:.2f} ms per matvec -> {agg / P:.2f} ms per rank; "
          f"wall on the shared device {wall * 1e3:.2f} ms -> {wall * 1e3 / P:.2f} ms per rank", flush=True)
    assert err <= 1e-12
    for c in comms:
        c.destroy()
    del ys, got
    torch.cuda.empty_cache()

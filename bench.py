#!/usr/bin/env python3
"""bench.py -- matvecs/s of y <- H x on the reference's Heisenberg chain (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model heisenberg_chain_32] [--dtype f64|c128]

One "step" = one full matrix-vector product (diagonal + off-diagonal, exchange included when N > 1)
with sigma, x, y resident in HBM.  N == 1: the fused single-partition kernel on heisenberg_chain_32
(601 080 390 states).  N > 1 (launched by torch.distributed.run, one rank per GPU): the same vector,
hash-partitioned over the ranks (strong scaling), all-to-all-v over RCCL/xGMI.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      algorithmic HBM bytes per matvec / HIP-event time of the dominant kernel
  cpu_baseline  oracle (C/OpenMP restatement of the reference's numLocales=1 algorithm) on the host
                cores, bounded sample; N == 1 only.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_model(name):
    # heisenberg_chain_<L>[_symm]
    parts = name.split("_")
    assert parts[0] == "heisenberg" and parts[1] == "chain", "bench workloads are the reference's chain configs"
    return int(parts[2]), len(parts) > 3 and parts[3] == "symm"


def chain_nnz(L, n_states):
    """exact number of off-diagonal non-zeros of the periodic chain in the half-filling sector:
    each of the L bonds is anti-aligned in a fraction L / (2 (L - 1)) of the states."""
    return n_states * L * L // (2 * (L - 1))


def source_sha():
    """sha256 of the kernel sources the library was built from (the k_*.hip translation units + what they share): ties a PMC
    measurement to the code it measured.  Same recipe as scripts/kernel_isa_sha.py::source_sha."""
    import hashlib

    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "distributed-matvec_amd", "csrc")
    for f in ("k_runtime.hip", "k_rows.hip", "k_packets.hip", "k_pull.hip", "k_plan.hip", "lsk_dev.hpp", "lsk.h"):
        path = os.path.join(csrc, f)
        if os.path.exists(path):
            with open(path, "rb") as fh:
                hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def kernel_isa_sha(family):
    """fingerprint of the machine code of one kernel family of this build (distributed-matvec_amd/kernel_isa.json, written
    by the build from the device assembly: scripts/kernel_isa_sha.py), or None when the file does not belong to the
    source in the tree"""
    try:
        with open(os.path.join(ROOT, "distributed-matvec_amd", "kernel_isa.json")) as f:
            isa = json.load(f)
    except (OSError, ValueError):
        return None
    if isa.get("source_sha") != source_sha():
        return None
    return isa.get("families", {}).get(family, {}).get("isa_sha")


def kernel_instance_sha(instance):
    """fingerprint of ONE template instantiation of this build ("k_chain_t<unsigned int, unsigned int, false, 1024, true>")"""
    try:
        with open(os.path.join(ROOT, "distributed-matvec_amd", "kernel_isa.json")) as f:
            isa = json.load(f)
    except (OSError, ValueError):
        return None
    if isa.get("source_sha") != source_sha():
        return None
    return isa.get("kernels", {}).get(instance)


PULL_KERNELS = ("direct-pull", "tile-pull", "replicated-")

# ---- model of the replicated-x exchange on P GPUs, printed next to every measured N > 1 number -----------
# The one-GPU time the model starts from is MEASURED IN THE SAME RUN: every rank times the one-partition kernel on the whole
# basis while it computes the parity reference (verify.reference_block).  What stays fixed are RATIOS of the split form of the
# projected pull kernel measured on one GPU in round 4 (profiles/r4_split_vs_fused_chain{36,40}symm_*.txt): resolve (stage A +
# K4 + slot look-ups, needs no x) = 0.85 x fused, gather (slots -> x -> y) = 0.23 x fused, owner-side prescaling = 0.016 x fused.
MODEL_RATIOS = {"resolve": 0.85, "gather": 0.23, "prescale": 0.0165}
MODEL_STATES = {"heisenberg_chain_40_symm": 861725794, "heisenberg_chain_36_symm": 63068876, "heisenberg_chain_32": 601080390}
XGMI_IN_GBPS = (7 * 50.0, 7 * 64.0)  # what one GPU receives from its 7 peers at once: 7 links x 50-64 GB/s achievable of 153 nominal
REACH_SHARE_MAX = 0.56               # 2.358 / 4.208 GB: the rank in the middle of the basis; the average over ranks is 0.44


def reach_share(P):
    """largest share of the other ranks' x a rank receives: at 2 ranks the rows of a rank reach everything, at 4 the middle ranks
    reach 85 % and the 80 % rule keeps the whole-vector exchange (profiles/r4_loopback_chain28_subrange_share_P2_P4.txt)"""
    return REACH_SHARE_MAX if P >= 8 else 1.0


REPL_OVERHEAD = 1.27                 # upper bound of the per-row cost of a rank's kernels relative to one GPU: the wall time of EIGHT loop-back
                                     # ranks sharing one device / the one-partition kernel (profiles/r6_loopback_chain40symm_chunked_return_
                                     # adaptive_split_ab.txt: 1.26-1.27 whatever the driver does -- eight row kernels at eight places of the basis
                                     # share one L2 / Infinity Cache there, which eight GPUs would not)
RETURN_CHUNKS = 4                    # rows of y travel back chunk by chunk while the next chunk is gathered (csrc/dist.c, repl_rows_chunked)
SPLIT_MARGIN = 1.15                  # adaptive split: rows resolved ahead = 1.15 x what hides the exchange of x (csrc/dist.c, adapt_split)


def scaling_model(model, P, w=8, fused_ms=None):
    """predicted ms per matvec of the replicated-x exchange on P GPUs and the speed-up over one GPU it implies.  Projected bases
    (round 6: adaptive split + chunked return): a share phi = min(1, 1.15 X / R) of the rows is resolved while x travels, the rest
    runs the fused kernel afterwards, and only the last chunk's rows of y return un-overlapped:
        t(P) = prescale / P + max(phi R, X) + phi G + (1 - phi) F + return / chunks,
        R, G, F = resolve, gather, fused per rank (x f), X = N w (P - 1) / P / B_in.
    The unprojected chain has no resolve step to hide the exchange behind and pays a permutation pass over what it received.
    fused_ms = the one-GPU matvec of this build measured in this run (None: no model)."""
    n = MODEL_STATES.get(model)
    if not n or P < 2 or not fused_ms:
        return None
    projected = model.endswith("_symm")
    m = {"fused": fused_ms, "n": n}
    if projected:
        m.update({k: r * fused_ms for k, r in MODEL_RATIOS.items()})
    xbytes = n * w * (P - 1) / P
    if not projected:
        # unprojected bases: the sub-range exchange (dist.c::setup_reach) -- the largest share a rank receives, measured with
        # loop-back ranks on chain_28 / chain_32 at P = 8 (profiles/r4_loopback_chain32_8ranks_subrange_exchange.txt: 2.36 of 4.21 GB)
        xbytes *= reach_share(P)
    out = {"inputs_ms_one_gpu": m, "inputs_source": "fused: measured in this run (one-partition kernel, this GPU); split ratios: round 4",
           "assumed_in_GBps": list(XGMI_IN_GBPS), "assumed_kernel_overhead": [1.0, REPL_OVERHEAD], "x_bytes_in_per_rank": xbytes}
    lo_hi = []
    for b in XGMI_IN_GBPS:
        for f in ((1.0, REPL_OVERHEAD) if projected else (1.0,)):  # from "no overhead" to the loop-back figure
            xch = xbytes / b / 1e6  # ms
            ret = n * w / P / 1.0e9 * 1e3 / 3000.0 + (n * w / P) * (P - 1) / P / b / 1e6  # group rows by owner (~3 TB/s) + send back
            if projected:
                R, G, F = m["resolve"] / P * f, m["gather"] / P * f, m["fused"] / P * f
                phi = min(1.0, SPLIT_MARGIN * xch / R)
                t = m["prescale"] / P + max(phi * R, xch) + phi * G + (1.0 - phi) * F + ret / RETURN_CHUNKS
            else:
                perm = reach_share(P) * n * 2 * w / 1.0e9 * 1e3 / 2500.0  # the hashed -> block permutation of what arrived: random reads + writes
                t = xch + perm + m["fused"] / P + ret
            lo_hi.append(t)
    out["predicted_ms_per_matvec"] = [min(lo_hi), max(lo_hi)]
    out["predicted_speedup_over_one_gpu"] = [m["fused"] / max(lo_hi), m["fused"] / min(lo_hi)]
    if projected:
        # with the slot cache the resolve step runs once per plan: a matvec is exchange + gather (+ prescale, + return)
        c = []
        for b in XGMI_IN_GBPS:
            for f in (1.0, REPL_OVERHEAD):
                ret = n * w / P / 1.0e9 * 1e3 / 3000.0 + (n * w / P) * (P - 1) / P / b / 1e6
                c.append(m["prescale"] / P + xbytes / b / 1e6 + m["gather"] / P * f + ret)
        out["predicted_ms_per_matvec_slot_cache"] = [min(c), max(c)]
    return out


def model_config(name):
    """(config dict, where it came from): the reference's own YAML input as parsed into tests/golden/models.json when the model
    is one of those (data/heisenberg_chain_{10,...,40_symm}.yaml; /root/reference does not exist on the GPU box), else the same
    model regenerated (periodic ring, sigma.sigma per bond)"""
    from distributed_matvec_amd import config

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "models.json")
    try:
        with open(path) as f:
            m = json.load(f)["models"].get(name)
        if m:
            return m["config"], f"data/{name}.yaml (the reference's input, parsed: tests/golden/models.json)"
    except (OSError, ValueError, KeyError):
        pass
    L, symm = parse_model(name)
    return (config.heisenberg_chain_config(L, symm=symm, spin_inversion=-1 if (L == 10 and not symm) else None),
            f"data/{name}.yaml (regenerated: periodic ring, sigma.sigma per bond)")


def slot_cache_leg(plan, run, check, kernel_times, time_steps, steps, allsum=None, world=1):
    """the same plan with ls_amd_plan_cache_slots: the packet streams (slot of every partner: 5 B per non-zero here) are resolved
    by the first matvec and kept in HBM, later matvecs only gather -- what an eigensolver that applies one plan hundreds of
    times runs.  Opt-in and NOT matrix-free, therefore a separate leg and never the headline."""
    rows = plan.cache_slots(0)
    # (with more than one rank the matvecs below are collective: every rank takes the leg, or none does)
    enabled = rows > 0 if allsum is None else allsum(1.0 if rows > 0 else 0.0) >= world
    if not enabled:
        return {"rows": 0, "note": "not enabled (no room for the packet streams on some rank, or nothing to cache)"}
    run()  # resolves
    check()
    kernel_times()
    dt = time_steps(run, steps, 1)
    check()
    ks = kernel_times()
    crow, cbytes = plan.slot_cache
    return {"rows": crow, "hbm_bytes": cbytes, "kernel": plan.kernel, "matvecs_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps,
            "kernel_ms_avg": sum(ks) / max(1, len(ks)), "matrix_free": False}


_REGISTERED_KEEP_ALIVE = []


def boundary_extra(D, torch, h, basis, reps, y_device, x_device, calls=2):
    """The drop-in entry itself -- `ls_chpl_matrix_vector_product(matrix, 1, double *x, double *y)` (DMV:1095-1110), what Diagonalize /
    PRIMME call -- timed on the benchmark workload with the caller's vectors in every kind of memory (include/ls_amd.h, "The
    host-pointer boundary"): pageable host memory (numpy; through the pinned bounce pipeline, and with LS_AMD_STAGE=0 through plain
    hipMemcpy), registered host memory (ls_amd_host_register), device memory (zero copies).  Wall time per call after one untimed
    call that builds and caches the plan; GB/s = bytes that crossed PCIe / wall time (kernel included).  Never the headline."""
    import ctypes as C

    import numpy as np

    from distributed_matvec_amd import _lib

    L = _lib.load()
    n = int(reps.numel())
    reps_h = reps.cpu().numpy().view(np.uint64)
    basis.uncheckedSetRepresentatives(reps_h)  # ls_hs_unchecked_set_representatives: the built basis the reference's callers hold
    x_h = x_device.cpu().numpy()
    y_h = np.empty(n)
    y_h.fill(0.0)
    f64p = _lib.c_f64p
    st = _lib.BoundaryStats()

    def call(xp, yp):
        L.ls_chpl_matrix_vector_product(h.payload, 1, C.cast(xp, f64p), C.cast(yp, f64p))
        _lib.raise_pending_halt()

    def timed(xp, yp, label):
        call(xp, yp)  # first call of a variant: plan cached, staging buffers / pool warm
        L.ls_amd_boundary_stats_get(C.byref(st), 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            call(xp, yp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        L.ls_amd_boundary_stats_get(C.byref(st), 1)
        moved = (st.bytes_h2d + st.bytes_d2h) / calls
        return {"ms_per_call": 1e3 * dt, "pcie_bytes_per_call": moved, "pcie_GBps": moved / dt / 1e9 if moved else None,
                "x": ("device" if st.device_x else label), "y": ("device" if st.device_y else label)}

    out = {"entry": "ls_chpl_matrix_vector_product (host-pointer ABI of the reference, DMV:1095-1110)", "n": n, "calls_timed": calls}
    t0 = time.perf_counter()
    call(x_h.ctypes.data, y_h.ctypes.data)  # uploads the representatives, builds and caches the plan
    out["first_call_seconds"] = time.perf_counter() - t0
    out["pageable_staged"] = timed(x_h.ctypes.data, y_h.ctypes.data, "pageable")
    err = float((torch.from_numpy(y_h).cuda() - y_device).abs().max()) / max(float(y_device.abs().max()), 1e-300)
    out["max_rel_err_vs_device_plan"] = err
    os.environ["LS_AMD_STAGE"] = "0"
    try:
        out["pageable_plain_hipMemcpy"] = timed(x_h.ctypes.data, y_h.ctypes.data, "pageable")
    finally:
        del os.environ["LS_AMD_STAGE"]
    _REGISTERED_KEEP_ALIVE.extend((x_h, y_h))  # (their addresses are not handed back: the runtime keeps state per registered range)
    for a in (x_h, y_h):
        _lib.check(L.ls_amd_host_register(C.c_void_p(a.ctypes.data), a.nbytes))
    try:
        out["registered"] = timed(x_h.ctypes.data, y_h.ctypes.data, "pinned")
    finally:
        for a in (x_h, y_h):
            L.ls_amd_host_unregister(C.c_void_p(a.ctypes.data))
    # the PRIMME callback on a block of 4 columns in pageable memory (Diagonalize.chpl:134-162): one pipeline -- column k + 1 goes up
    # and column k - 1 comes down while column k computes -- against column-by-column plain copies (LS_AMD_STAGE=0)
    try:
        bs = 4
        Xb = np.empty((bs, n))
        Yb = np.empty((bs, n))
        for k in range(bs):
            Xb[k] = x_h
        Yb.fill(0.0)
        pbuf = (C.c_char * 512)()
        C.c_int64.from_buffer(pbuf, 0).value = n
        C.c_int64.from_buffer(pbuf, L.ls_amd_test_primme_nlocal_offset()).value = n
        C.c_void_p.from_buffer(pbuf, L.ls_amd_test_primme_matrix_offset()).value = C.cast(h.payload, C.c_void_p).value

        def block_call():
            ldx, ldy, blk, ierr = C.c_int64(n), C.c_int64(n), C.c_int(bs), C.c_int(0)
            L.ls_chpl_primme_matvec(C.c_void_p(Xb.ctypes.data), C.byref(ldx), C.c_void_p(Yb.ctypes.data), C.byref(ldy), C.byref(blk),
                                    C.cast(pbuf, C.c_void_p), C.byref(ierr))
            _lib.raise_pending_halt()
            assert ierr.value == 0

        res = {}
        for label, env in (("pipelined", "1"), ("plain_hipMemcpy", "0")):
            if env is not None:
                os.environ["LS_AMD_STAGE"] = env
            try:
                block_call()
                L.ls_amd_boundary_stats_get(C.byref(st), 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                block_call()
                torch.cuda.synchronize()
                dtb = time.perf_counter() - t0
                L.ls_amd_boundary_stats_get(C.byref(st), 1)
                res[label] = {"ms_per_block": 1e3 * dtb, "ms_per_column": 1e3 * dtb / bs, "pcie_GBps_both_directions": (st.bytes_h2d + st.bytes_d2h) / dtb / 1e9}
            finally:
                if env is not None:
                    del os.environ["LS_AMD_STAGE"]
        res["max_rel_err_vs_device_plan"] = float((torch.from_numpy(Yb[bs - 1]).cuda() - y_device).abs().max()) / max(float(y_device.abs().max()), 1e-300)
        res["block_size"] = bs
        out["primme_block_pageable"] = res
        del Xb, Yb
    except Exception as e:  # reported, never hidden
        out["primme_block_pageable"] = {"error": repr(e)[:300]}
    y_d = torch.zeros_like(y_device)
    out["device_pointers"] = timed(x_device.data_ptr(), y_d.data_ptr(), "device")
    out["device_pointers"]["max_rel_err_vs_device_plan"] = float((y_d - y_device).abs().max()) / max(float(y_device.abs().max()), 1e-300)
    out["staging_threads"] = int(os.environ.get("LS_AMD_STAGE_THREADS", 0)) or "min(16, cores / 4)"
    basis.uncheckedSetRepresentatives(np.zeros(0, dtype=np.uint64))  # drops the cached plan, its staging buffers and the device copy
    return out


def packet_path_extra(D, torch, time_steps, L=28, P=8, steps=5):
    """The reference's own formulation -- term expansion, hash -> owner, per-destination buffers, local scatter (DMV:663-853) -- on
    one device: heisenberg_chain_L over P logical partitions through ls_amd_matvec (the "exchange" is a pointer hand-off), once with
    the sorted packet streams + window consumers (csrc/k_packets.hip, k_tile_sd / k_window) and once with the atomic consumers
    (LS_AMD_PACKET_STREAMS=0), each checked element-wise against the one-partition pull kernel on the same x."""
    from distributed_matvec_amd import config

    basis, h = D.loadConfigFromDict(config.heisenberg_chain_config(L), hamiltonian=True)
    reps1, _ = D.enumerateStates(basis, 1)
    x1 = D.fillRandom(reps1[0], 42, torch.float64)
    y1 = torch.empty_like(x1)
    ref = D.MatvecPlan(h, reps1, torch.float64, mode="pull")
    ref.matvec([x1], [y1], check=True)
    ref.destroy()
    reps, masks = D.enumerateStates(basis, P)
    xs = D.arrFromBlockToHashed(x1, masks, P)
    out = {"workload": f"heisenberg_chain_{L} over {P} logical partitions on one device (ls_amd_matvec; f64)", "states": int(x1.numel())}
    scale = float(y1.abs().max())
    saved = os.environ.get("LS_AMD_PACKET_STREAMS")
    try:
        for label, env in (("sorted_streams", None), ("atomic_consumers", "0")):
            if env is None:
                os.environ.pop("LS_AMD_PACKET_STREAMS", None)
            else:
                os.environ["LS_AMD_PACKET_STREAMS"] = env
            ys = [torch.full_like(v, 1.5) for v in xs]
            pl = D.MatvecPlan(h, reps, torch.float64)
            t = time_steps(lambda: pl.matvec(xs, ys, check=False), steps, 2)
            pl.check()
            err = float((D.arrFromHashedToBlock(ys, masks) - y1).abs().max()) / scale
            out[label] = {"kernel": pl.kernel, "ms_per_matvec": t / steps * 1e3, "matvecs_per_s": steps / t, "nnz": pl.nnz, "rounds": pl.num_rounds,
                          "packet_bytes": pl.packet_bytes, "max_rel_err_vs_one_partition": err, "ok": err <= 1e-12}
            pl.destroy()
            del ys
    finally:
        if saved is None:
            os.environ.pop("LS_AMD_PACKET_STREAMS", None)
        else:
            os.environ["LS_AMD_PACKET_STREAMS"] = saved
    a, b = out["sorted_streams"], out["atomic_consumers"]
    out["speedup"] = b["ms_per_matvec"] / a["ms_per_matvec"]
    # SURVEY 8(d) push formula: rows (8 + w) + nnz 2w bytes per matvec, against the HBM line
    b_alg = int(x1.numel()) * 16 + a["nnz"] * 16
    out["survey_push_formula"] = {"bytes_per_matvec": b_alg, "GBps": b_alg / (a["ms_per_matvec"] * 1e-3) / 1e9,
                                  "frac_of_8TBps": b_alg / (a["ms_per_matvec"] * 1e-3) / 8e12}
    return out


def eigensolve_extra(D, torch, name, max_basis=12, eps=1e-7):
    """the caller of the path (BASELINE config 5: Diagonalize): ground state of one of the projected chains with the device-resident
    thick-restart Lanczos of diagonalize.py -- enumeration, plan, the slot cache in whatever HBM the Krylov basis leaves, fused
    Gram-Schmidt sweeps; wall seconds from the YAML-equivalent config to the converged eigenpair"""
    from distributed_matvec_amd.diagonalize import LocalOperator, lanczos_smallest

    t0 = time.perf_counter()
    basis, h = D.loadConfigFromDict(model_config(name)[0], hamiltonian=True)
    reps, _masks = D.enumerateStates(basis, 1)
    n = int(reps[0].numel())
    free, _total = torch.cuda.mem_get_info()
    op = LocalOperator(h, reps, torch.float64, slot_cache_bytes=max(0, int(free) - (max_basis + 6) * n * 8 - (8 << 30)))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = lanczos_smallest(op, num_evals=1, eps=eps, max_basis=max_basis, max_restarts=200)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = {"states": n, "E0": res.eigenvalues[0], "E0_per_site": res.eigenvalues[0] / basis.numberSites(), "converged": bool(res.converged),
           "residual": res.residual_norms[0], "matvecs": res.matvecs, "restarts": res.restarts, "max_basis": max_basis, "eps": eps,
           "seconds_setup": t1 - t0, "seconds_solve": t2 - t1, "seconds_per_lanczos_step": (t2 - t1) / max(1, res.matvecs),
           "slot_cache_rows": op.cached_rows, "slot_cache_bytes": op.plan.slot_cache[1], "kernel": op.plan.kernel}
    op.plan.destroy()
    return out


def verify_operator(op, x, y, ref, allsum, allmax, inject_fault=False):
    """the `parity` object of one exchange strategy (distributed-matvec_amd/verify.py): a fresh matvec of the N-rank operator on
    the benchmark vector against this rank's rows of the one-partition kernel, element by element, plus all-reduced invariants.
    inject_fault (test hook): misplace one segment of the exchange first -- the object must then say ok = false."""
    from distributed_matvec_amd import verify

    x_ref, y_ref, ymax, ref_kernel = ref
    injected = None
    if inject_fault:
        injected = allsum(1.0 if op.inject_fault() else 0.0)
    y.zero_()
    op.matvec(x, y, check=False)
    halted = None
    try:  # the device error flag (a misplaced state outside the basis halts, DMV:115-118) is local: every rank still takes part
        op.engine.check()  # in the reductions below, and all of them learn about it
    except Exception as e:
        halted = str(e)[:300]
    n_halted = allsum(1.0 if halted else 0.0)
    out = verify.parity_object(y, x, y_ref, ymax, allsum=allsum, allmax=allmax, reference_kernel=ref_kernel)
    same_x = x.numel() == x_ref.numel() and bool((x == x_ref).all())
    out["x_equals_reference_x"] = bool(allsum(0.0 if same_x else 1.0) == 0)
    out["ok"] = bool(out["ok"] and out["x_equals_reference_x"] and n_halted == 0)
    if n_halted:
        out["halted_on_ranks"] = int(n_halted)
        if halted:
            out["halt"] = halted
    if injected is not None:
        out["fault_injected_on_ranks"] = int(injected)
    return out


def one_gpu_parity(D, torch, h, reps, tdtype, x, y, plan, projected, make_for_fault=None):
    """(parity, parity_after_fault) of a one-GPU leg: one more matvec of the measured plan on the benchmark vector, compared element
    by element with every alternative kernel (verify.alternative_kernels).  make_for_fault (--inject-fault): a second plan of the
    same kind whose row kernel is made to skip one row (ls_amd_test_corrupt_plan) -- the same check must then fail."""
    from distributed_matvec_amd import verify

    t0 = time.perf_counter()
    refs = verify.single_gpu_references(h, reps, tdtype, x, projected)
    y.zero_()
    plan.matvec([x], [y], check=True)
    par = verify.single_gpu_parity(y, x, refs, plan.kernel)
    par["seconds"] = time.perf_counter() - t0
    fault = None
    if make_for_fault is not None:
        p2 = make_for_fault()
        injected = p2.inject_fault()
        y.zero_()
        p2.matvec([x], [y], check=True)
        fault = verify.single_gpu_parity(y, x, refs, p2.kernel)
        fault["fault_injected_on_ranks"] = int(injected)
        p2.destroy()
        y.zero_()
        plan.matvec([x], [y], check=True)
    del refs
    torch.cuda.empty_cache()
    return par, fault


def projected_extra(D, torch, dist, name, rank, world, time_steps, allsum, steps=3, warmup=2, distributed=False, allmax=None,
                    inject_fault=False, cpu=None):
    """one of the symmetry-projected BASELINE chains, measured inside the default run (see main)"""
    from distributed_matvec_amd import config

    L, symm = parse_model(name)
    basis, h = D.loadConfigFromDict(model_config(name)[0], hamiltonian=True)
    t0 = time.perf_counter()
    parts, masks = D.enumerateStates(basis, world)
    n_total = int(masks.numel())
    out = {"states": n_total, "steps": steps, "warmup": warmup, "dtype": "f64"}
    if world == 1 and not distributed:
        reps = parts[0]
        x = [D.fillRandom(reps, 42, torch.float64)]
        y = [torch.zeros_like(x[0])]
        pl = D.MatvecPlan(h, [reps], torch.float64)
        out["setup_seconds"] = time.perf_counter() - t0
        pl.enable_timing(64)
        for _ in range(warmup):
            pl.matvec(x, y, check=False)
        pl.check()
        pl.kernel_times_ms()
        dt = time_steps(lambda: pl.matvec(x, y, check=False), steps, 0)
        pl.check()
        ks = pl.kernel_times_ms()
        kms = sum(ks) / max(1, len(ks))
        out.update({"matvecs_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "kernel": pl.kernel, "kernel_ms_avg": kms})
        ent, note = pmc_entry(name, "f64", pl.kernel)
        ia = int_alu_object(ent, kms * 1e-3)
        if ia:
            out["int_alu"] = ia
        if ent and ent.get("traffic_bytes"):
            # what bounds this kernel: random 64-byte requests (index-table probes, partner values), not bytes
            out["requests_64B_per_s"] = ent["traffic_bytes"] / 64.0 / (kms * 1e-3)
        out["pmc_note"] = note
        make2 = (lambda: D.MatvecPlan(h, [reps], torch.float64)) if inject_fault else None
        out["parity"], fault = one_gpu_parity(D, torch, h, reps, torch.float64, x[0], y[0], pl, True, make2)
        if fault is not None:
            out["parity_after_fault"] = fault
        out["slot_cache"] = slot_cache_leg(pl, lambda: pl.matvec(x, y, check=False), pl.check, pl.kernel_times_ms, time_steps, steps)
        pl.destroy()
        if cpu is not None:
            # "next to the reference CPU path timed on the same box" (north_star) for the projected configs too: the oracle on
            # this box's host cores -- chain_36_symm measured, chain_40_symm scaled from it and labelled so
            pp = D.MatvecPlan(h, [reps], torch.float64, mode="push")
            out["nnz"] = int(pp.nnz)
            pp.destroy()
            take = L <= 36
            out["cpu_baseline"] = cpu_baseline_projected(name, reps.cpu().numpy().view("uint64") if take else None, out["nnz"],
                                                         probe=cpu.get("probe"))
            if out["cpu_baseline"].get("probe"):
                cpu["probe"] = out["cpu_baseline"].pop("probe")
            out["gpu_over_cpu"] = out["matvecs_per_s"] / out["cpu_baseline"]["value"]
        return out
    from distributed_matvec_amd.distributed import RcclReplicatedOperator

    my = parts[rank].clone()
    reps_global = D.arrFromHashedToBlock(parts, masks)
    del parts
    from distributed_matvec_amd import verify

    comm = D.Communicator.from_torch()
    x = D.fillRandom(my, 42, torch.float64)
    y = torch.zeros_like(x)
    out["setup_seconds"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = verify.reference_block(h, reps_global, masks, rank, torch.float64)  # this rank's rows of the one-partition kernel
    out["reference_seconds"] = time.perf_counter() - t0
    one_gpu_ms = verify.LAST_REFERENCE_MS[0]  # the one-partition matvec of this build on this GPU, timed while it made the reference
    out["one_gpu_ms_per_matvec_this_run"] = one_gpu_ms
    t0 = time.perf_counter()
    op = RcclReplicatedOperator(h, reps_global, masks, torch.float64, comm=comm)
    out["setup_seconds"] += time.perf_counter() - t0
    plan = op.engine.plan
    plan.enable_stage_timing(4096)
    for _ in range(warmup):
        op.matvec(x, y, check=False)
    plan.check()
    plan.stage_times()
    dt = time_steps(lambda: op.matvec(x, y, check=False), steps, 0)
    plan.check()
    stages, mv = plan.stage_times()
    out.update({"matvecs_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "kernel": plan.kernel, "exchange": "replicated",
                "n_gpus": world, "exchange_bytes_per_matvec": allsum(getattr(op, "exchange_bytes_per_matvec", 0)),
                "x_bytes_in_this_rank": op.x_bytes_in,
                "rank0_stage_ms_per_matvec": {k: v[0] / max(1, mv) for k, v in stages.items()},
                "model": scaling_model(name, world, fused_ms=one_gpu_ms)})
    if one_gpu_ms:
        out["speedup_over_one_gpu_this_run"] = one_gpu_ms / out["ms_per_step"]
    out["parity"] = verify_operator(op, x, y, ref, allsum, allmax)
    out["slot_cache"] = slot_cache_leg(plan, lambda: op.matvec(x, y, check=False), plan.check, plan.kernel_times_ms, time_steps, steps,
                                       allsum=allsum, world=world)
    if out["slot_cache"].get("rows"):  # the cached streams serve the same matvec: verified the same way
        out["slot_cache"]["parity"] = verify_operator(op, x, y, ref, allsum, allmax)
    if inject_fault:
        out["parity_after_fault"] = verify_operator(op, x, y, ref, allsum, allmax, inject_fault=True)
    op.rm.destroy()
    return out


def pmc_entry(model, dtype, kernel_name):
    """the committed PMC entry (profiles/pmc_traffic.json) of (model, dtype, plan kernel) if it was measured on the machine code
    of this build's kernel; else (None, why)"""
    sha = source_sha()
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(f"{model}/{dtype}/{kernel_name}")
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if not ent:
        return None, "no PMC entry for this workload / kernel"
    isa = kernel_isa_sha(ent.get("device_kernel", "").rstrip("<"))
    same = (ent.get("isa_sha") is not None and ent.get("isa_sha") == isa) or ent.get("source_sha") == sha
    if ent.get("instance_isa_sha"):
        same = same or kernel_instance_sha(ent.get("instance", "")) == ent["instance_isa_sha"]
    if not same:
        return None, f"PMC entry is stale (measured on {ent.get('device_kernel')} ISA {ent.get('isa_sha')}, source_sha {ent.get('source_sha')})"
    return ent, "profiles/pmc_traffic.json"


def int_alu_object(ent, t):
    """integer-ALU view of the projected bases: wave64 VALU instructions x 64 lanes against the spec rate, and as
    issue time at the ~4 cycles the shifts / counts / multiplies / f64 ops of these kernels take (profiles/r3_valu_issue_rates.txt)"""
    if not ent or not ent.get("valu_insts") or not t:
        return None
    peak = 256 * 4 * 32 * 2.4e9
    ach = ent["valu_insts"] * 64 / t
    return {"valu_wave_insts_per_launch": ent["valu_insts"], "achieved_lane_ops_per_s": ach, "peak_lane_ops_per_s": peak,
            "frac": ach / peak, "issue_time_frac_at_4_cycles": ent["valu_insts"] * 4 / (256 * 4 * 2.4e9) / t}


def roofline_object(args, kernel_name, kernel_ms, launches_per_step, rows_here, n_total, nnz, w, world, sec_per_step, symm,
                    row_bytes=8, dtype=None):
    """HBM roofline of the dominant kernel, per launch.

    ALGORITHMIC bytes = the compulsory traffic of the formulation the kernel executes:
      push (direct-push, tile: the reference's formulation):
          rows (8 + w) [+ w for the y write of the separate diagonal pass] + nnz 2w   (RMW of y_j per non-zero)
      pull (Hermitian operators; y written once, no RMW): rows (row_bytes + 2w), row_bytes = the per-row basis / plan data
          the measured instantiation streams (ls_amd_plan_row_bytes: the 8-byte fused sigma|partner record of the staged
          f64 kernel -- 24 B per row --, 4 + 4 for the c128 / unfused instantiations, 8 + 8 B norm for projected bases).
    `achieved` = those bytes / average launch time, `frac` = achieved / peak: cannot exceed 1.
    `traffic` = fabric bytes of the same launch from the committed PMC passes (profiles/pmc_traffic.json), attached
    only when the entry was measured on the very machine code of this build's kernel (isa_sha of the kernel family, or the
    sha of the whole kernel source for entries without one); `frac_traffic` = traffic / t / peak,
    `wasted_traffic` = traffic / algorithmic bytes (> 1 = re-reads the caches did not absorb).
    `survey_formula` keeps the SURVEY.md 8(d) push-formula figure for cross-reference; for a pull kernel it is NOT a
    bandwidth (the kernel never performs the 2w read-modify-write per non-zero the formula charges)."""
    nnz_here = nnz * rows_here // max(1, n_total)
    pull = kernel_name.startswith(PULL_KERNELS)
    if pull:
        alg = rows_here * (row_bytes + 2 * w)
        formulation = f"pull: rows ({row_bytes} B state / plan data + 2w); y written once"
    elif kernel_name.startswith("direct-push"):
        alg = rows_here * (8 + w) + nnz_here * 2 * w  # (k_direct: the diagonal pass is a separate, tiny kernel; k_push_t adds it itself)
        formulation = "push: rows (8 + w) + nnz 2w (SURVEY 8(d))" + (
            "; staged: near targets and the diagonal part meet in an LDS window of y per tile, one atomic per touched row"
            if kernel_name.endswith("+staged") else "")
    else:  # tile: staged push; remote packets are written (8 + w) and scattered by k_scatter
        alg = rows_here * (8 + w) + nnz_here * 2 * w
        formulation = "push via packets: rows (8 + w) + nnz 2w (SURVEY 8(d)); all launches of one matvec"
    alg_per_launch = alg / launches_per_step
    t = kernel_ms * 1e-3 if kernel_ms else None
    achieved = alg_per_launch / t / 1e9 if t else None
    sha = source_sha()
    traffic = int_alu = None
    traffic_note = "no PMC entry for this workload / kernel"
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(f"{args.model}/{dtype or args.dtype}/{kernel_name}")
        if ent and world == 1:
            isa = kernel_isa_sha(ent.get("device_kernel", "").rstrip("<"))
            same_code = (ent.get("isa_sha") is not None and ent.get("isa_sha") == isa) or ent.get("source_sha") == sha
            if ent.get("instance_isa_sha"):  # the measured instantiation itself, whatever happened to its siblings
                inst = kernel_instance_sha(ent.get("instance", ""))
                same_code = same_code or inst == ent["instance_isa_sha"]
                if inst == ent["instance_isa_sha"]:
                    isa = f"{inst} ({ent['instance']})"
            if same_code:
                traffic = ent["traffic_bytes"]
                traffic_note = (f"profiles/pmc_traffic.json ({ent.get('source')}), measured on source_sha "
                                f"{ent.get('source_sha')}; {ent.get('device_kernel')} ISA {ent.get('isa_sha')} == this build's")
                if ent.get("valu_insts"):
                    # integer-ALU roofline of the projected bases: wave64 VALU instructions x 64 lanes
                    peak_lane_ops = 256 * 4 * 32 * 2.4e9  # CUs x SIMDs x lanes per clock x Hz
                    ach = ent["valu_insts"] * 64 / t if t else None
                    # ... against the spec rate (2 cycles per wave64 instruction), and as issue time at the ~4 cycles the shifts,
                    # bit-field / count instructions, multiplies and f64 ops of these kernels measure on this part
                    # (profiles/r3_valu_issue_rates.txt; = the VALUBusy formula of the SQ counters)
                    busy4 = ent["valu_insts"] * 4 / (256 * 4 * 2.4e9) / t if t else None
                    int_alu = {"valu_wave_insts_per_launch": ent["valu_insts"], "achieved_lane_ops_per_s": ach,
                               "peak_lane_ops_per_s": peak_lane_ops, "frac": ach / peak_lane_ops if ach else None,
                               "issue_time_frac_at_4_cycles": busy4}
            else:
                traffic_note = (f"PMC entry is stale: measured on {ent.get('device_kernel')} ISA {ent.get('isa_sha')} (source_sha "
                                f"{ent.get('source_sha')}), this build is ISA {isa} (source_sha {sha}); re-run "
                                f"scripts/gpu_pmc_traffic.sh")
    except (OSError, ValueError):
        pass
    b_alg_push = n_total * (8 + 2 * w) + nnz * 2 * w
    out = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic,
        "frac_compulsory": achieved / HBM_PEAK_GBPS if achieved else None,
        "frac_traffic": traffic / t / 1e9 / HBM_PEAK_GBPS if (traffic and t) else None,
        "wasted_traffic": traffic / alg_per_launch if traffic else None,
        "traffic_note": traffic_note,
        "kernel": kernel_name, "formulation": formulation, "kernel_ms_avg": kernel_ms,
        "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": alg_per_launch,
        "guide_stream_copy_GBps": 6290.0,  # float4 copy on this part (MI355X_MICROARCH.md); measured live below
        "survey_formula": {"B_alg_push_bytes_per_matvec": b_alg_push,
                           "matvec_over_B_alg_GBps": b_alg_push / sec_per_step / 1e9 / world,
                           "note": "push-formula bytes / time per GPU; a bandwidth only for push kernels"},
    }
    if int_alu:
        out["int_alu"] = int_alu
    return out


def cpu_baseline(sample_L, threads=0, repeats=3):
    """oracle ("port") timed on the host cores: heisenberg_chain_<sample_L>, full matvec, STEADY STATE -- before the timed call
    the OpenMP pool has run a matvec on a small chain, x and y have been written (every page touched: `np.zeros` alone hands out
    untouched pages, and the 4.8 GB of page faults of chain_32 would land inside the diagonal pass) and the look-up table of the
    representatives has been built (oracle `prepare_index`: it belongs to the basis, not to a matvec).  Large samples (the benchmark
    workload itself: chain_32 is ~15 s per matvec on 128 threads) are then timed ONCE so that the default run stays bounded;
    small ones `repeats` times after one untimed matvec, best time reported."""
    import numpy as np

    from oracle import c_oracle as CO
    from oracle import model as M

    big = sample_L >= 30
    cores = CO.lib().lso_num_threads() if threads <= 0 else threads
    if big:  # spin the OpenMP pool up on something small
        ow = CO.COracle(M.model_from_config(M.heisenberg_chain_config(20)))
        rw = ow.enumerate()
        ow.local_matvec(rw, np.random.RandomState(1).rand(len(rw)) - 0.5, num_threads=cores)
    cfg = M.heisenberg_chain_config(sample_L)
    o = CO.COracle(M.model_from_config(cfg))
    reps = np.ascontiguousarray(o.enumerate(), dtype=np.uint64)
    n = len(reps)
    x = np.random.RandomState(42).rand(n) - 0.5
    y = np.empty(n)
    y.fill(0.0)  # touched
    CO.COracle.prepare_index(reps)
    if not big:
        o.local_matvec(reps, x, y, num_threads=cores)  # warm-up
    times = []
    for _ in range(1 if big else repeats):
        t = time.perf_counter()
        o.local_matvec(reps, x, y, num_threads=cores)
        times.append(time.perf_counter() - t)
    best = min(times)
    return {"seconds_per_matvec": best, "n": n, "nnz": chain_nnz(sample_L, n), "cores": int(cores), "L": sample_L,
            "timed_matvecs": len(times)}


REFERENCE_RECORDED = {  # the only figures the reference records for its own matvec (hardware "cn20", core count not stated)
    "heisenberg_chain_36_symm": 38.897812843322754, "heisenberg_chain_40_symm": 682.9306001663208}


def cpu_baseline_projected(name, reps_np, nnz, budget_s=60.0, threads=0, probe=None):
    """the oracle on one of the symmetry-projected chains (BASELINE configs 4 / 5), on the host cores of this box: ONE timed
    matvec of the whole basis in steady state (pool warm, x / y touched, index table built) when a probe on
    heisenberg_chain_28_symm says it fits `budget_s`, else that probe with the cost per (non-zero x group element) scaled.
    `reps_np`: the representatives (the GPU enumeration's, bit-identical to the oracle's by the parity tests -- enumerating
    6e7 orbits on the host would cost more than the matvec)."""
    import numpy as np

    from oracle import c_oracle as CO
    from oracle import model as M

    cores = CO.lib().lso_num_threads() if threads <= 0 else threads
    L, _ = parse_model(name)

    def timed(cfg, reps):
        o = CO.COracle(M.model_from_config(cfg))
        reps = np.ascontiguousarray(reps if reps is not None else o.enumerate(), dtype=np.uint64)
        x = np.random.RandomState(42).rand(len(reps)) - 0.5
        y = np.empty(len(reps))
        y.fill(0.0)
        CO.COracle.prepare_index(reps)
        t = time.perf_counter()
        o.local_matvec(reps, x, y, num_threads=cores)
        return time.perf_counter() - t, len(reps)

    if probe is None:
        probe_L = 28
        timed(M.heisenberg_chain_config(20, symm=True), None)  # thread pool
        t_probe, n_probe = timed(M.heisenberg_chain_config(probe_L, symm=True), None)
        nnz_probe = chain_nnz(probe_L, n_probe)  # anti-aligned bonds per representative: what the row expansion generates
        probe_name = f"heisenberg_chain_{probe_L}_symm"
    else:  # a larger chain measured earlier in this run
        t_probe, n_probe, nnz_probe, probe_L, probe_name = probe
    # cost model: every non-zero runs state_info over the 4 L group elements (2 L with the reflection, x 2 for the spin flip)
    per_unit = t_probe / (nnz_probe * 4 * probe_L)
    est = per_unit * nnz * 4 * L
    out = {"unit": "matvecs/s", "cores": int(cores), "kind": "port",
           "reference_recorded_seconds_per_matvec": REFERENCE_RECORDED.get(name),
           "reference_recorded_note": "/root/reference/example/Example05.chpl:97-102 (\"On cn20 ... spent in matrix-vector using OpenMP\"): "
                                      "the reference's own figure on its authors' node, core count not stated; context, not a measurement of this box"}
    if reps_np is not None and est <= budget_s:
        t, n = timed(model_config(name)[0], reps_np)
        out.update({"value": 1.0 / t, "sample_seconds_per_matvec": t, "scaled": False, "probe": (t, n, nnz, L, name),
                    "sample": f"{name} itself: one full matvec (f64, {n} representatives, {nnz} non-zeros), {t:.2f} s on {cores} threads, "
                              f"steady state, measured (no extrapolation)"})
    else:
        out.update({"value": 1.0 / est, "sample_seconds_per_matvec": t_probe, "scaled": True,
                    "sample": f"SCALED: {probe_name} full matvec ({n_probe} representatives, {nnz_probe} non-zeros) "
                              f"{t_probe:.2f} s on {cores} threads; cost per (non-zero x group element) scaled to {name} "
                              f"({nnz} non-zeros x {4 * L} elements) = {est:.1f} s per matvec"})
    return out


def default_cpu_sample(L):
    """the benchmark workload itself when the host can hold and finish it (chain_32 f64 = 14.4 GB, ~17 s per matvec on the
    GPU box's 128 threads); otherwise chain_28 with the per-non-zero cost scaled (and labelled so)"""
    try:
        cores = os.cpu_count() or 1
        with open("/proc/meminfo") as f:
            mem_kb = int(next(line for line in f if line.startswith("MemAvailable")).split()[1])
    except (OSError, StopIteration, ValueError):
        return min(L, 28)
    if L <= 28:
        return L
    need_gb = 24.0 * math.comb(L, L // 2) / 1e9 * 1.5
    return L if (cores >= 64 and mem_kb / 1e6 >= need_gb and L <= 32) else 28


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="heisenberg_chain_32")
    ap.add_argument("--dtype", default="f64", choices=["f64", "c128"])
    ap.add_argument("--mode", default=os.environ.get("LS_AMD_MODE", "auto"), choices=["auto", "push", "pull"])
    ap.add_argument("--exchange", default="auto", choices=["auto", "packets", "replicated"],
                    help="N > 1: all-to-all-v of packets (reference formulation), all-gather of x + pull, or (auto) both")
    ap.add_argument("--distributed-extras", action="store_true",
                    help="with --force-distributed: also run the projected-chain extras through the N > 1 code path (one rank)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the N > 1 code path (process group, exchange) even with one rank (test hook)")
    ap.add_argument("--inject-fault", action="store_true",
                    help="test hook: after the timed steps misplace one segment of every exchange strategy's layout (N > 1 code path, "
                         "ls_amd_test_corrupt_dist / _repl) or make the row kernel skip one row (N = 1, ls_amd_test_corrupt_plan); the parity "
                         "check must catch it and the run must exit non-zero")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="chain length of the CPU-baseline sample (0 = the workload itself when the host has >= 64 cores and "
                         "the memory for it, else 28)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (other dtype / other mode) measurements")
    ap.add_argument("--kDisplayTimings", action="store_true",
                    help="print the per-stage timing tree (the reference's flag, DMV:1028-1052) to stderr after the timed steps")
    args = ap.parse_args()

    import torch

    import distributed_matvec_amd as D
    from distributed_matvec_amd import config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked exactly like the N = 1 line (`python bench.py --gpus N ...`): become the launcher, one rank per GPU
        # over RCCL, rendezvous on 127.0.0.1 (what the driver's own torch.distributed.run line does)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {have} HIP device(s) visible")
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    torch.cuda.set_device(local_rank)
    dist = None
    distributed = world > 1 or args.force_distributed
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    L, symm = parse_model(args.model)
    cfg, cfg_source = model_config(args.model)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    tdtype = torch.float64 if args.dtype == "f64" else torch.complex128
    w = 8 if args.dtype == "f64" else 16

    comm_box = []  # the C host's communicator once it exists (distributed runs)

    def drain():
        """device idle before anything else is issued.  With the C host's communicator: a wait WITH A DEADLINE on its streams first
        (ls_amd_comm_wait), so that a dead peer or mismatched counts end this rank with a message and a non-zero exit code instead
        of leaving it in torch.cuda.synchronize() until the driver's timeout; and torch's own collectives (another communicator
        of the same RCCL) are only ever issued on an idle device -- the two never have operations in flight at the same time."""
        if comm_box:
            try:
                comm_box[0].wait()
            except D.LsAmdError as e:
                print(f"bench.py: rank {rank}: EXCHANGE DID NOT COMPLETE: {e}", file=sys.stderr, flush=True)
                os._exit(4)
        torch.cuda.synchronize()

    def barrier():
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_setup = time.perf_counter()
    reps_parts, masks = D.enumerateStates(basis, world)
    n_total = int(masks.numel())
    if distributed:
        my_reps = reps_parts[rank].clone()
        reps_global = D.arrFromHashedToBlock(reps_parts, masks) if world > 1 else reps_parts[0]
    else:
        my_reps = reps_parts[0]
        reps_global = None
    del reps_parts
    torch.cuda.empty_cache()
    x = D.fillRandom(my_reps, 42, tdtype)
    y = torch.zeros_like(x)

    def time_steps(run, steps, warmup):
        for _ in range(warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def allsum(v):
        if dist is None:
            return v
        drain()
        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        return float(t.item())

    def allmax(v):
        if dist is None:
            return v
        drain()
        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    nnz = chain_nnz(L, n_total) if not symm else None
    extra = {}
    extra_rooflines = {}

    def measure(make_op, steps, warmup, label):
        """returns (dt, kernel_ms, launches_per_step, kernel_name, exchange_bytes, plan)"""
        op = make_op()
        plan = op if isinstance(op, D.MatvecPlan) else op.engine.plan
        plan.enable_timing(8192)
        if args.kDisplayTimings:
            plan.enable_stage_timing(65536)
        if isinstance(op, D.MatvecPlan):
            run = lambda: op.matvec([x], [y], check=False)  # noqa: E731
        else:
            run = lambda: op.matvec(x, y, check=False)  # noqa: E731
        for _ in range(warmup):
            run()
        plan.check()
        plan.kernel_times_ms(8192)  # drop warm-up samples
        dt = time_steps(run, steps, 0)
        plan.check()
        samples = plan.kernel_times_ms(8192)
        if args.kDisplayTimings and rank == 0:
            print(f"[{label}] " + plan.timing_report(), file=sys.stderr, flush=True)
        lps = max(1, len(samples) // max(1, steps))
        kms = sum(samples) / max(1, len(samples))  # average duration of ONE launch of the dominant kernel
        xb = allsum(getattr(op, "exchange_bytes_per_matvec", 0))
        x_in = getattr(op, "x_bytes_in", None)
        if x_in is not None:  # replicated-x: what this rank and all ranks receive of x per matvec (sub-range exchange: < N - N/P elements)
            x_in_info[label] = {"rank0": x_in, "all_ranks": allsum(x_in), "whole_vector_all_ranks": allsum((n_total - int(my_reps.numel())) * w)}
        return dt, kms, lps, plan.kernel, xb, plan, op

    exchanges = {}
    failed = {}
    x_in_info = {}
    if not distributed:
        make = lambda: D.MatvecPlan(h, [my_reps], tdtype, mode=args.mode)  # noqa: E731
        exchange = "none"
        setup_t0 = time.perf_counter()
        dt, kernel_ms, launches_per_step, kernel_name, exchange_bytes, plan, op_obj = measure(make, args.steps, args.warmup, "main")
        # the y that was timed, checked in the same run (the reference's single-locale check,
        # test/TestMatrixVectorProduct.chpl:25-39, with the stored /y replaced by kernels that share no device code with the
        # measured one): outside the timed region, < 1 s
        parity_main, parity_main_fault = one_gpu_parity(D, torch, h, my_reps, tdtype, x, y, plan, bool(symm), make if args.inject_fault else None)
    else:
        from distributed_matvec_amd.distributed import RcclDistributedOperator, RcclReplicatedOperator

        # Both exchange strategies are first-class numbers (same hash-partitioned x / y / representatives at the
        # interface, same result):
        #   packets     the reference's formulation: (sigma_j, c_j x_i) packets, all-to-all-v inside the C host
        #               (ls_amd_dist_matvec: grouped ncclSend/ncclRecv over xGMI, double-buffered rounds)
        #   replicated  Hermitian operators: exchange x itself (N w bytes instead of nnz (8 + w)) and pull locally
        #               (ls_amd_repl_matvec: grouped ncclSend/ncclRecv of the blocks + permutation + pull + all-to-all-v of y)
        # A failure of either is an error of the run, never a footnote.
        from distributed_matvec_amd import verify

        comm = D.Communicator.from_torch()
        comm_box.append(comm)
        # Evidence of the transport and of the result: the communicator size as RCCL reports it, the ranks an
        # all-reduce through it counts, and -- per strategy, below -- y against the one-partition kernel on the same vector.
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        comm.allreduce_sum(one)
        torch.cuda.synchronize()
        rccl_info = {"comm_count": comm.rccl_count(), "ranks_by_allreduce": int(round(float(one.item()))), "world_size": world,
                     "transport": "rccl (ncclSend/ncclRecv inside the C host)" if comm.rccl_count() > 0 else "loop-back"}
        t_ref = time.perf_counter()
        reference = verify.reference_block(h, reps_global, masks, rank, tdtype)
        rccl_info["reference_seconds"] = time.perf_counter() - t_ref
        one_gpu_ms_main = verify.LAST_REFERENCE_MS[0]
        rccl_info["one_gpu_ms_per_matvec_this_run"] = one_gpu_ms_main  # the one-partition kernel on the whole basis, this GPU
        makers = {"packets": lambda: RcclDistributedOperator(h, my_reps, tdtype, comm=comm)}
        if h.isHermitian:
            makers["replicated"] = lambda: RcclReplicatedOperator(h, reps_global, masks, tdtype, comm=comm)
        wanted = list(makers) if args.exchange == "auto" else [args.exchange]
        # per-rank HBM of the two strategies: replicated x keeps O(N) tables on EVERY rank, the packets O(N / P);
        # `auto` measures both while both fit and only the packets when the replicated tables do not (LS_AMD_EXCHANGE_HBM_CEILING)
        from distributed_matvec_amd.distributed import choose_exchange, exchange_memory_estimate

        free_b, _tot = torch.cuda.mem_get_info()
        hbm_est = exchange_memory_estimate(n_total, int(my_reps.numel()), world, w, bool(symm), h.numberOffDiagTerms())
        rccl_info["exchange_hbm_estimate_bytes"] = hbm_est
        # (one decision for all ranks: a rank with less room pulls everybody to the packets)
        if args.exchange == "auto" and "replicated" in makers and allsum(1.0 if choose_exchange(True, hbm_est, int(free_b)) == "packets" else 0.0) > 0:
            wanted = ["packets"]
            rccl_info["exchange_auto"] = "packets only: the replicated-x tables do not fit this rank's HBM"
        setup_t0 = time.perf_counter()
        results = {}
        failed = {}
        for name in wanted:
            torch.cuda.empty_cache()
            r, err = None, None
            try:
                r = measure(makers[name], args.steps, args.warmup, name)
            except Exception as e:  # reported at the TOP level of the JSON line (failed_exchanges) and on stderr, never hidden
                import traceback

                traceback.print_exc()
                err = repr(e)[:400]
            # every rank takes the same view of a strategy: failed anywhere = failed everywhere (a rank that raised outside a
            # collective -- allocation, plan creation -- would otherwise leave its peers with a number it never produced)
            if allsum(1.0 if err else 0.0) > 0:
                failed[name] = err or "failed on another rank"
                exchanges[name] = {"error": failed[name]}
                r = None
                continue
            results[name] = r
            exchanges[name] = {"matvecs_per_s": args.steps / r[0], "ms_per_step": 1e3 * r[0] / args.steps, "kernel": r[3],
                               "kernel_ms_avg": r[1], "launches_per_step": r[2], "exchange_bytes_per_matvec": r[4]}
            exchanges[name]["parity"] = verify_operator(r[6], x, y, reference, allsum, allmax)
            if args.inject_fault:
                exchanges[name]["parity_after_fault"] = verify_operator(r[6], x, y, reference, allsum, allmax, inject_fault=True)
            if name in x_in_info:
                exchanges[name]["x_bytes_in"] = x_in_info[name]
            if len(wanted) > 1:  # keep only the numbers; the plans of the other strategy would pin HBM
                results[name] = r[:5] + (None, None)
                del r
        del reference
        if not results:
            raise SystemExit(f"every exchange strategy failed: {failed}")
        # `value` is the faster strategy, named in config.exchange; the times are max-over-ranks (all-reduced in
        # time_steps), so every rank picks the same one
        exchange = min(sorted(results), key=lambda k: results[k][0])
        dt, kernel_ms, launches_per_step, kernel_name, exchange_bytes, plan, op_obj = results[exchange]
    setup_s = (setup_t0 - t_setup)
    if symm:
        # non-zeros of the projected matrix: the packet count of a push plan's count pass
        if not distributed:
            pp = D.MatvecPlan(h, [my_reps], tdtype, mode="push")
        else:
            pp = D.MatvecPlan(h, my_reps, tdtype, my_partition=rank, num_partitions=world, mode="push")
        nnz = int(allsum(pp.nnz))
        pp.destroy()

    ms_per_step = 1e3 * dt / args.steps
    value = args.steps / dt

    roofline = roofline_object(args, kernel_name, kernel_ms, launches_per_step, int(my_reps.numel()), n_total, nnz, w, world,
                               dt / args.steps, symm, row_bytes=plan.row_bytes if plan is not None else 8)
    # attainable streaming rate on THIS box ("measure the attainable peak with a device copy"): x -> y, 16-byte
    # lanes, read + written bytes over HIP-event time
    try:
        nb = x.numel() * x.element_size()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import ctypes as C

        from distributed_matvec_amd import _lib

        def copy():
            _lib.check(_lib.load().ls_amd_stream_copy(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), nb,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))

        copy()
        ev0.record()
        for _ in range(5):
            copy()
        ev1.record()
        torch.cuda.synchronize()
        roofline["measured_stream_copy_GBps"] = 2 * nb * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        if roofline.get("achieved"):
            roofline["frac_of_measured_stream"] = roofline["achieved"] / roofline["measured_stream_copy_GBps"]
        # ... and the read-only rate: the pull kernels' traffic is > 90 % reads, so this is the line that traffic can reach
        sink = torch.zeros(4, dtype=torch.int32, device=x.device)

        def read():
            _lib.check(_lib.load().ls_amd_stream_read(C.c_void_p(x.data_ptr()), nb, 2, C.c_void_p(sink.data_ptr()),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))

        read()
        ev0.record()
        for _ in range(5):
            read()
        ev1.record()
        torch.cuda.synchronize()
        roofline["measured_stream_read_GBps"] = nb * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        if roofline.get("traffic") and kernel_ms:
            roofline["traffic_over_measured_read_rate"] = roofline["traffic"] / (kernel_ms * 1e-3) / 1e9 / roofline["measured_stream_read_GBps"]
    except RuntimeError:
        pass

    if not distributed and not args.no_extra and not symm and args.dtype == "f64" and args.mode == "auto":
        # the reference's own host-pointer entry on this workload, per kind of caller memory
        try:
            plan.matvec([x], [y], check=True)
            extra["boundary_host_ptr"] = boundary_extra(D, torch, h, basis, my_reps, y, x)
        except Exception as e:  # reported, never hidden
            import traceback

            traceback.print_exc()
            extra["boundary_host_ptr"] = {"error": repr(e)[:400]}
    if not distributed and not args.no_extra and not symm and args.dtype == "f64" and args.mode == "auto" and args.model == "heisenberg_chain_32":
        # the packet path (the reference's formulation; the O(N / P)-memory exchange) on one device, self-verified
        try:
            extra["packet_path"] = packet_path_extra(D, torch, time_steps)
        except Exception as e:  # reported, never hidden
            import traceback

            traceback.print_exc()
            extra["packet_path"] = {"error": repr(e)[:400]}
    if not distributed and not args.no_extra and not symm:
        # secondary numbers in the same run: the other scatter/gather mode and the other dtype
        for label, mode2 in (("f64" if args.dtype == "c128" else "c128", args.mode),
                             (args.dtype, "pull" if kernel_name.startswith("direct-push") else "push")):
            try:
                td = torch.float64 if label == "f64" else torch.complex128
                x2 = D.fillRandom(my_reps, 42, td)
                y2 = torch.zeros_like(x2)
                p2 = D.MatvecPlan(h, [my_reps], td, mode=mode2)
                steps2 = max(3, args.steps // 2)
                p2.enable_timing(4096)
                for _ in range(2):
                    p2.matvec([x2], [y2], check=False)
                p2.check()
                p2.kernel_times_ms(4096)  # drop the warm-up samples
                t2 = time_steps(lambda: p2.matvec([x2], [y2], check=False), steps2, 0)
                p2.check()
                ks2 = p2.kernel_times_ms(4096)
                w2 = 8 if label == "f64" else 16
                extra[f"{label}/{p2.kernel}"] = {
                    "matvecs_per_s": steps2 / t2,
                    "whole_matvec_GBps": (n_total * (8 + 2 * w2) + nnz * 2 * w2) / (t2 / steps2) / 1e9,
                }
                # first-class roofline objects of the north star's dtype (c128) and formulation (push: atomics), built like the
                # headline's -- HIP-event time of the dominant kernel, algorithmic bytes of the formulation it executes, the PMC
                # entry of ITS machine code (profiles/pmc_traffic.json, profiles/r6_*_rocprof_summary.txt): VERDICT r5 #8
                if ks2:
                    ro2 = roofline_object(args, p2.kernel, sum(ks2) / len(ks2), max(1, len(ks2) // steps2), int(my_reps.numel()), n_total, nnz, w2,
                                          world, t2 / steps2, symm, row_bytes=p2.row_bytes, dtype=label)
                    ro2["matvecs_per_s"] = steps2 / t2
                    ro2["dtype"] = label
                    extra_rooflines["roofline_c128" if label == "c128" and label != args.dtype else
                                    ("roofline_push" if p2.kernel.startswith("direct-push") else f"roofline_{label}_{p2.kernel}")] = ro2
                if label != args.dtype:  # the other dtype (c128 on the default run: the north star's): its own parity object
                    extra[f"{label}/{p2.kernel}"]["parity"], _f = one_gpu_parity(D, torch, h, my_reps, td, x2, y2, p2, False)
                p2.destroy()
                del x2, y2
            except D.LsAmdError as e:  # e.g. pull on a non-Hermitian operator
                extra[f"{label}/{mode2}"] = {"error": str(e)}

    # BASELINE configs 4 and 5 (the symmetry-projected chains) in the same run: 3 timed matvecs each, so that the driver's
    # record carries projected-basis numbers too -- on one GPU the fused indexed pull kernel (with the integer-ALU and the
    # request-rate view of it; an HBM-byte fraction means nothing there), on N > 1 GPUs the replicated-x exchange (slot
    # resolution overlapped with the all-gather of x) next to what the model of DESIGN.md section 4 predicts.  The headline
    # `value` / `config` stay those of --model.
    if args.model == "heisenberg_chain_32" and not args.no_extra and args.dtype == "f64" and (not args.force_distributed or args.distributed_extras):
        try:
            del plan, op_obj
        except NameError:
            pass
        x = y = my_reps = reps_global = masks = None
        torch.cuda.empty_cache()
        cpu_state = {}
        for name in ("heisenberg_chain_36_symm", "heisenberg_chain_40_symm"):
            try:
                extra[name] = projected_extra(D, torch, dist, name, rank, world, time_steps, allsum, distributed=distributed,
                                              allmax=allmax, inject_fault=args.inject_fault,
                                              cpu=cpu_state if (rank == 0 and not distributed and not args.no_cpu_baseline) else None)
            except Exception as e:  # reported, never hidden
                import traceback

                traceback.print_exc()
                extra[name] = {"error": repr(e)[:400]}
            if allsum(1.0 if "error" in extra[name] else 0.0) > 0 and "error" not in extra[name]:
                extra[name] = {"error": "failed on another rank"}
            torch.cuda.empty_cache()
        if world > 1:
            extra["model_heisenberg_chain_32"] = scaling_model("heisenberg_chain_32", world, fused_ms=one_gpu_ms_main)
        elif not distributed:
            for name in ("heisenberg_chain_36_symm", "heisenberg_chain_40_symm"):
                try:
                    extra["eigensolve_" + name] = eigensolve_extra(D, torch, name)
                except Exception as e:  # reported, never hidden
                    import traceback

                    traceback.print_exc()
                    extra["eigensolve_" + name] = {"error": repr(e)[:400]}
                torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        s = cpu_baseline(min(args.cpu_sample, L) if args.cpu_sample > 0 else default_cpu_sample(L))
        # scale the measured per-non-zero cost of the sample to the benchmark workload
        per_nnz = s["seconds_per_matvec"] / s["nnz"]
        est = per_nnz * nnz
        cpu = {
            "value": 1.0 / est, "unit": "matvecs/s", "cores": s["cores"], "kind": "port",
            "sample": f"heisenberg_chain_{s['L']} full matvec (f64, {s['n']} states, {s['nnz']} nnz) "
                      f"{s['seconds_per_matvec']:.3f} s on {s['cores']} threads; per-nnz cost scaled to "
                      f"{args.model} ({nnz} nnz)" if s["L"] != L else
                      f"{args.model} itself: one full matvec (f64, {s['n']} states, {s['nnz']} nnz), "
                      f"{s['seconds_per_matvec']:.3f} s on {s['cores']} threads, steady state (pool warm, x / y touched, index "
                      f"table built before the timed call), measured (no extrapolation)",
            "sample_seconds_per_matvec": s["seconds_per_matvec"],
        }

    # every `parity` object of the run (exchange strategies of --model, the projected extras and their cached legs): one
    # verdict, and a non-zero exit code when any of them failed (or, with --inject-fault, when a corrupted layout went unnoticed)
    checks = {f"exchanges.{k}": v["parity"] for k, v in exchanges.items() if isinstance(v, dict) and "parity" in v}
    faults = {f"exchanges.{k}": v["parity_after_fault"] for k, v in exchanges.items() if isinstance(v, dict) and "parity_after_fault" in v}
    for k, v in extra.items():
        if isinstance(v, dict):
            if "parity" in v:
                checks[f"extra.{k}"] = v["parity"]
            if isinstance(v.get("slot_cache"), dict) and "parity" in v["slot_cache"]:
                checks[f"extra.{k}.slot_cache"] = v["slot_cache"]["parity"]
            if "parity_after_fault" in v:
                faults[f"extra.{k}"] = v["parity_after_fault"]
    if not distributed:  # N = 1: the headline's own object, next to the c128 leg's and the projected extras'
        checks["main"] = parity_main
        if parity_main_fault is not None:
            faults["main"] = parity_main_fault
    parity_failed = sorted(k for k, v in checks.items() if not v.get("ok"))
    fault_missed = sorted(k for k, v in faults.items() if v.get("ok") and v.get("fault_injected_on_ranks", 0) > 0)
    parity_summary = None
    if distributed or checks:
        parity_summary = {"checked": sorted(checks), "failed": parity_failed,
                          "max_rel_err": max([v.get("max_rel_err", float("inf")) for v in checks.values()] or [None]),
                          "ok": not parity_failed and bool(checks)}
        if not distributed:
            parity_summary["main"] = parity_main
        if args.inject_fault:
            parity_summary["fault_injection"] = {"detected": sorted(k for k, v in faults.items() if not v.get("ok")), "missed": fault_missed,
                                                 "nothing_to_corrupt": sorted(k for k, v in faults.items() if v.get("fault_injected_on_ranks", 0) == 0)}
    if rank == 0:
        out = {
            "metric": "matvecs/sec", "value": value, "unit": "matvecs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": f"{args.model}: y <- H x, {n_total} basis states, {nnz} off-diagonal non-zeros, "
                            f"{args.dtype} vectors, sigma/x/y resident in HBM",
                "model_yaml": cfg_source,
                "partitions": world, "partitioning": "hash64_01(sigma) % n_gpus" if world > 1 else "single",
                "exchange": exchange,
                "kernel": kernel_name, "x": "u(hash(sigma, 42)) - 0.5",
                "exchange_bytes_per_matvec": exchange_bytes,
            },
            "exchanges": exchanges,
            "value_is": (f"best of {len(exchanges)} exchange strategies ({exchange})" if distributed and len(exchanges) > 1
                         else "the single configuration measured"),
            "failed_exchanges": sorted(failed) if distributed else [],
            "rccl": rccl_info if distributed else None,
            "parity": parity_summary,
            "roofline": roofline,
            **extra_rooflines,
            "cpu_baseline": cpu,
            "setup_seconds": setup_s,
            "extra": extra,
        }
        if cpu:
            out["gpu_over_cpu"] = value / cpu["value"]
        # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit, AFTER the JSON line
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if (parity_failed or not checks or (args.inject_fault and (fault_missed or any(not v.get("ok") for v in faults.values())))):
        # wrong y on some rank (or a deliberately corrupted exchange, which must end the same way): the number above is not a result
        print(f"bench.py: PARITY FAILURE: {parity_failed or 'injected fault'}", file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()

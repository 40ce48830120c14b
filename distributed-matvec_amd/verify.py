"""Self-verification of a hash-partitioned matvec: this rank's block of y against the ONE-partition kernel.

The reference checks its multi-locale product against the stored single-locale result
(/root/reference/test/TestMatrixVectorProduct.chpl:41-59: every locale's block of `y` is converted hashed -> block and compared with
the `/y` of the input file).  Here the stored result is replaced by the one-partition kernel of this library on the same vector
x = u(hash(sigma, seed)) - 0.5 -- a function of the basis state alone, so every partitioning sees the same logical vector -- and the
comparison is element by element on the rows this rank owns (selected through `masks`, the owner of every state in global
ascending order), plus three all-reduced invariants (sum y, <x, y>, ||y||^2) that tie the ranks' blocks together.

`bench.py --gpus N` attaches the resulting object to every exchange strategy it times and exits non-zero on a mismatch: the
driver's multi-GPU run is the only place the RCCL transport executes with more than one rank, so the number it produces must
carry its own correctness evidence.  One-partition kernel and N-rank path share the term tables but neither the row kernel
(staged pull vs packets / replicated pull over hashed blocks) nor the index structures nor, of course, the exchange.
"""
from __future__ import annotations

TOLERANCE = 1e-12  # max |dy| / max |y_ref|: sums of ~20 terms of O(1) in another order differ at 1e-15
LAST_REFERENCE_MS = [None]  # wall time (HIP events) of the last reference matvec: one partition, whole basis, this GPU


def reference_block(matrix, reps_global, masks, part: int, dtype, seed: int = 42):
    """(x_part, y_part) of the one-partition kernel: x = fillRandom over the WHOLE basis in one partition, y = H x computed by a
    single-partition plan, both restricted to the states `masks` gives to partition `part` (ascending = the order of that
    partition's block).  Frees everything else before it returns."""
    import torch

    from . import api

    x = api.fillRandom(reps_global, seed, dtype)
    y = torch.zeros_like(x)
    plan = api.MatvecPlan(matrix, [reps_global], dtype)
    kernel = plan.kernel
    try:
        plan.matvec([x], [y], check=True)
        # the same matvec once more, timed: the one-GPU time of THIS build on THIS box, which bench.py's scaling model starts from
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.matvec([x], [y], check=True)
        e1.record()
        torch.cuda.synchronize()
        LAST_REFERENCE_MS[0] = float(e0.elapsed_time(e1))
    finally:
        plan.destroy()
    n = int(masks.numel())
    # rows of this partition, chunked so that the index tensors stay small next to 6e8-element vectors
    xs, ys = [], []
    step = 1 << 27
    for lo in range(0, n, step):
        sel = (masks[lo:lo + step] == part).nonzero(as_tuple=True)[0]
        xs.append(x[lo:lo + step][sel])
        ys.append(y[lo:lo + step][sel])
        del sel
    xp, yp = torch.cat(xs), torch.cat(ys)
    ymax = float(y.abs().max()) if n else 0.0
    del x, y, xs, ys
    torch.cuda.empty_cache()
    return xp, yp, ymax, kernel


def parity_object(y_part, x_part, y_ref_part, ymax_ref: float, allsum=None, allmax=None, reference_kernel: str = ""):
    """what `bench.py` prints per exchange strategy.  allsum / allmax reduce a python float over the ranks (identity when None)."""
    import torch

    allsum = allsum or (lambda v: v)
    allmax = allmax or (lambda v: v)
    # (a collective verdict: a rank whose block has the wrong size must not leave while its peers sit in the reductions below)
    differ = y_part.numel() != y_ref_part.numel()
    if allsum(1.0 if differ else 0.0) > 0:
        return {"ok": False, "error": f"block sizes differ: {y_part.numel()} vs {y_ref_part.numel()} reference rows" if differ
                else "block sizes differ on another rank"}
    d = (y_part - y_ref_part).abs()
    max_abs = allmax(float(d.max()) if d.numel() else 0.0)
    bad = allsum(float((d > TOLERANCE * max(ymax_ref, 1e-300)).sum()) if d.numel() else 0.0)
    del d

    def dot(a, b):
        v = torch.vdot(a, b) if a.is_complex() else torch.dot(a, b)
        return complex(v.item()) if a.is_complex() else float(v.item())

    def red(v):  # all-reduce a real or complex python scalar
        return complex(allsum(v.real), allsum(v.imag)) if isinstance(v, complex) else allsum(v)

    inv = {}
    for name, got, ref in (("sum_y", y_part.sum().item(), y_ref_part.sum().item()),
                           ("dot_x_y", dot(x_part, y_part), dot(x_part, y_ref_part)),
                           ("norm2_y", dot(y_part, y_part), dot(y_ref_part, y_ref_part))):
        g, r = red(complex(got) if isinstance(got, complex) else float(got)), red(complex(ref) if isinstance(ref, complex) else float(ref))
        scale = max(abs(r), ymax_ref, 1e-300)
        inv[name] = {"value": [g.real, g.imag] if isinstance(g, complex) else g, "reference": [r.real, r.imag] if isinstance(r, complex) else r,
                     "rel_err": abs(g - r) / scale}
    rel = max_abs / max(ymax_ref, 1e-300)
    ok = bool(rel <= TOLERANCE and bad == 0 and all(v["rel_err"] <= 1e-9 for v in inv.values()))
    return {"ok": ok, "max_abs_err": max_abs, "max_rel_err": rel, "rows_off": int(bad), "tolerance": TOLERANCE,
            "max_abs_y_reference": ymax_ref, "invariants": inv,
            "reference": f"one-partition kernel ({reference_kernel}) on x = u(hash(sigma, seed)) - 0.5, compared element-wise on every rank's rows"}


class _Env:
    """set / restore environment knobs the C host reads at plan creation"""

    def __init__(self, **kv):
        self.kv, self.saved = kv, {}

    def __enter__(self):
        import os

        for k, v in self.kv.items():
            self.saved[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        import os

        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def alternative_kernels(matrix, projected: bool):
    """the independent formulations a one-GPU result is held against, as (label, plan mode, environment) -- none of them shares the
    measured kernel's device code:
      unprojected bases   the GENERIC row kernel k_direct in pull form (LS_AMD_ROW_KERNEL=generic: one row per lane, term loop,
                          index by combinadic rank or prefix table) and in PUSH form (the reference's formulation: atomics);
      projected bases     PUSH (packets of state-carrying keys -> search -> atomics) and the VALUE-TABLE pull kernel
                          (LS_AMD_PULL_INDEXED=0: block-wide lists, binary-search window, {rep -> x n(rep)} hash table)."""
    if projected:
        out = [("push_atomics", "push", {})]
        if matrix.isHermitian:
            out.append(("pull_value_table", "pull", {"LS_AMD_PULL_INDEXED": "0"}))
        return out
    out = [("generic_push_atomics", "push", {"LS_AMD_ROW_KERNEL": "generic"})]
    if matrix.isHermitian:
        out.insert(0, ("generic_pull", "pull", {"LS_AMD_ROW_KERNEL": "generic"}))
    return out


def single_gpu_references(matrix, reps, dtype, x, projected: bool):
    """y of every alternative kernel on the same x: {label: (y_ref, kernel name)} -- the reference's single-locale check
    (/root/reference/test/TestMatrixVectorProduct.chpl:25-39) with the stored `/y` replaced by independent device kernels."""
    import torch

    from . import api

    refs = {}
    for label, mode, env in alternative_kernels(matrix, projected):
        with _Env(**env):
            plan = api.MatvecPlan(matrix, [reps], dtype, mode=mode)
        try:
            y = torch.zeros_like(x)
            plan.matvec([x], [y], check=True)
            refs[label] = (y, plan.kernel)
        finally:
            plan.destroy()
    return refs


def single_gpu_parity(y, x, refs, measured_kernel: str):
    """the `parity` object of a one-GPU bench leg: the measured kernel's y against every reference of single_gpu_references,
    element by element, plus the invariants; ok = all of them agree."""
    out = {"measured_kernel": measured_kernel, "against": {}}
    ok = bool(refs)
    worst = 0.0
    rows_off = 0
    for label, (y_ref, kname) in refs.items():
        ymax = float(y_ref.abs().max()) if y_ref.numel() else 0.0
        o = parity_object(y, x, y_ref, ymax, reference_kernel=kname)
        o["reference"] = f"{kname} on the same x, all {y_ref.numel()} rows"
        o["independent_of_measured_kernel"] = kname != measured_kernel
        ok = ok and o["ok"] and kname != measured_kernel
        worst = max(worst, o["max_rel_err"])
        rows_off = max(rows_off, o["rows_off"])
        out["against"][label] = o
    first = next(iter(out["against"].values()), None)
    out.update({"ok": bool(ok), "max_rel_err": worst, "rows_off": rows_off, "tolerance": TOLERANCE,
                "sum_y": first["invariants"]["sum_y"]["value"] if first else None,
                "dot_x_y": first["invariants"]["dot_x_y"]["value"] if first else None})
    return out

"""examples/c_matvec.c: the C ABI from C -- no Python, no torch, no HIP headers on the caller's side.  Compiled with gcc
against include/*.h and libls_amd.so, run as its own process, checked against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L", [12, 18, "yaml-16"])
def test_c_caller(tmp_path, L):
    from oracle import c_oracle as CO
    from oracle import model as M

    arg = str(L)
    if isinstance(L, str):  # the model comes from a YAML file in the reference's layout, loaded by the C library itself
        from test_yaml_loader import reference_style

        L = int(L.split("-")[1])
        arg = str(tmp_path / f"heisenberg_chain_{L}.yaml")
        with open(arg, "w", encoding="utf-8") as f:
            f.write(reference_style(M.heisenberg_chain_config(L)))
    exe = str(tmp_path / "c_matvec")
    libdir = os.path.join(ROOT, "distributed-matvec_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_matvec.c"), "-L", libdir, "-lls_amd", f"-Wl,-rpath,{libdir}", "-lm",
                           "-o", exe])
    dump = str(tmp_path / "y.bin")
    out = subprocess.run([exe, arg, dump], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout
    m = re.search(r"N = (\d+), <x\|H\|x>/<x\|x> = (-?[\d.]+)", out.stdout)
    o = CO.COracle(M.model_from_config(M.heisenberg_chain_config(L)))
    reps = o.enumerate()
    assert int(m.group(1)) == len(reps)
    x = np.sin(0.37 * np.arange(len(reps))) + 0.1
    y = o.local_matvec(reps, x)
    assert abs(float(m.group(2)) - float(x @ y) / float(x @ x)) < 1e-9
    # every element of every path against the oracle, tolerance of the reference's own check (test/TestMatrixVectorProduct.chpl:15-20)
    raw = np.fromfile(dump, dtype=np.int64, count=2)
    n, have_rccl = int(raw[0]), int(raw[1])
    assert n == len(reps)
    body = np.fromfile(dump, dtype=np.uint64, offset=16)
    assert body.size == 5 * n
    assert np.array_equal(body[:n], reps)  # ls_hs_basis_build from C == the oracle's enumeration, bit for bit
    assert np.array_equal(body[n:2 * n].view(np.float64), x)
    from helpers import approx_equal

    for k, label in ((2, "plug-in (ls_chpl_matrix_vector_product, host pointers)"), (3, "ls_amd_matvec, 3 partitions"), (4, "ls_amd_dist_matvec, one rank")):
        if k == 4 and not have_rccl:
            continue
        got = body[k * n:(k + 1) * n].view(np.float64)
        close = approx_equal(got, y)
        assert close.all(), (label, int((~close).sum()), float(np.abs(got - y).max()))
    assert "one-rank RCCL path: 3 rounds, kernel tile" in out.stdout

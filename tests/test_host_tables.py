"""CPU-side checks of the product's host logic (no kernels run): the C-ABI library loads and exports
every symbol include/*.h declares; YAML -> term tables; symmetry groups, characters and the compiled
bit-permutation networks; loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import distributed_matvec_amd as D
from distributed_matvec_amd import _lib, config
from helpers import (CHECK_MODELS, SMALL_MODELS, apply_terms_python, complex_translation_config, golden,
                     model_config, oracle_for, product_terms)
from oracle import model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("ls_hs.h", "ls_chpl.h", "ls_amd.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src):
            name = m.group(1)
            if name.startswith(("ls_", "primme")):
                names.add(name)
    return names


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 60
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    # and the typed binding covers them
    _lib.load()


@pytest.mark.parametrize("header", ["ls_hs.h", "ls_chpl.h", "ls_amd.h"])
def test_every_header_stands_alone_in_c_and_cxx(header, tmp_path):
    """the boundary is what a maintainer includes from C (Chapel's C backend, PRIMME glue) or C++: every header of include/ must
    compile on its own, -pedantic clean, as C11 and as C++17"""
    import subprocess

    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    src = tmp_path / "one.c"
    src.write_text(f'#include "{header}"\nint ls_header_probe(void) {{ return 0; }}\n')
    for cmd in (["gcc", "-std=c11"], ["g++", "-std=c++17", "-x", "c++"]):
        r = subprocess.run(cmd + ["-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "one.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_integration_guide_names_only_symbols_the_library_exports():
    """INTEGRATION.md is the binding a maintainer would write: every ls_* / primme* function it CALLS must be an export of the library"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    called = set(re.findall(r"\b((?:ls_amd_|ls_hs_|ls_chpl_)[a-z0-9_]+|primme(?:GlobalSumReal|BroadcastReal))\s*\(", text))
    quoted = set(re.findall(r'dlsym\(h, "([a-z_0-9]+)"\)', text))
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(root, "distributed-matvec_amd", "libls_amd.so")], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if line.strip()}
    assert len(called) >= 10 and quoted
    assert not sorted((called | quoted) - exported), sorted((called | quoted) - exported)


def test_no_gpu_means_loud_failure():
    L = _lib.load()
    if L.ls_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    basis, h = D.loadConfigFromDict(model_config("heisenberg_chain_10"), hamiltonian=True)
    with pytest.raises(D.LsAmdError):
        D.enumerateStates(basis)
    with pytest.raises(D.LsAmdError):
        basis.build()
    with pytest.raises(D.LsAmdError):
        h @ np.zeros(126)


def test_plain_c_caller_links_and_halts_without_a_gpu(tmp_path):
    """examples/c_matvec.c needs nothing but include/*.h and libls_amd.so to build (no HIP, no torch on the caller's side); without a
    device its first call, ls_chpl_init, halts like the reference's `halt` -- a message and a non-zero exit, no CPU fallback"""
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible: tests/test_gpu_c_example.py runs the example")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "distributed-matvec_amd")
    exe = str(tmp_path / "c_matvec")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_matvec.c"),
                           "-L", libdir, "-lls_amd", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe])
    out = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "no HIP device" in out.stdout + out.stderr


@pytest.mark.parametrize("name", sorted(golden()["models"].keys()))
def test_basis_flags_and_groups_match_oracle(name):
    cfg = model_config(name)
    basis = D.loadConfigFromDict(cfg)
    m = M.model_from_config(cfg)
    assert basis.numberSites() == m.number_sites
    assert basis.requiresProjection() == m.requires_projection
    assert basis.isStateIndexIdentity() == m.state_index_is_identity
    assert basis.hasPermutationSymmetries() == m.has_permutations
    assert basis.hasSpinInversionSymmetry() == (m.spin_inversion != 0)
    assert basis.isHammingWeightFixed() == (m.hamming_weight >= 0)
    assert basis.groupOrder() == m.group.perms.shape[0]
    assert basis.numberWords() == 1
    # min/max state estimates bracket the candidates the enumerator walks
    if m.hamming_weight >= 0:
        assert bin(basis.minStateEstimate()).count("1") == m.hamming_weight
        assert bin(basis.maxStateEstimate()).count("1") == m.hamming_weight
    # same set of (permutation action, character): compare through the action on random states
    L = _lib.load()
    rs = np.random.RandomState(5)
    states = [int(v) & m.mask for v in rs.randint(0, 2**62, size=16, dtype=np.int64)]
    want = {}
    for p, ch in zip(m.group.perms, m.group.chars):
        key = tuple(M.apply_perm(p, s) for s in states)
        want[key] = ch
    got = {}
    for g in range(basis.groupOrder()):
        key = tuple(int(L.ls_amd_basis_apply_group_element(basis.payload, g, C.c_uint64(s))) for s in states)
        re_, im_ = C.c_double(), C.c_double()
        assert L.ls_amd_basis_group_character(basis.payload, g, C.byref(re_), C.byref(im_)) == 0
        got[key] = complex(re_.value, im_.value)
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) < 1e-12


@pytest.mark.parametrize("L,sector", [(8, 1), (12, 5), (10, 3)])
def test_complex_characters(L, sector):
    cfg = complex_translation_config(L, sector)
    basis = D.loadConfigFromDict(cfg)
    m = M.model_from_config(cfg)
    lib = _lib.load()
    chars_oracle = sorted([(round(c.real, 12), round(c.imag, 12)) for c in m.group.chars])
    chars = []
    for g in range(basis.groupOrder()):
        re_, im_ = C.c_double(), C.c_double()
        lib.ls_amd_basis_group_character(basis.payload, g, C.byref(re_), C.byref(im_))
        chars.append((round(re_.value, 12), round(im_.value, 12)))
    assert sorted(chars) == chars_oracle


def test_incompatible_sectors_are_rejected():
    # reflection sector 1 together with translation sector 1 on a ring is not a 1-D representation
    L = 6
    cfg = M.heisenberg_chain_config(L)
    cfg["basis"]["symmetries"] = [
        {"permutation": [(i + 1) % L for i in range(L)], "sector": 1},
        {"permutation": [L - 1 - i for i in range(L)], "sector": 1},
    ]
    with pytest.raises(D.LsAmdError):
        D.loadConfigFromDict(cfg)
    with pytest.raises(ValueError):
        M.model_from_config(cfg)


def test_generators_of_a_huge_group_are_refused_quickly():
    """one wrong entry in a permutation of a YAML file is enough to generate (nearly) the symmetric group: the closure must end
    with an error in milliseconds -- membership is a hash look-up, the order is capped -- not scan ~2^40 pairs (found by
    mutation-fuzzing the loader: 6 of 6 runs hung on such an input).  A large but legitimate group still closes."""
    import time

    L = 40
    cfg = M.heisenberg_chain_config(L)
    swapped = [L - 1 - i for i in range(L)]
    swapped[10], swapped[12] = swapped[12], swapped[10]  # the reflection with two entries exchanged
    cfg["basis"]["symmetries"] = [{"permutation": [(i + 1) % L for i in range(L)], "sector": 0}, {"permutation": swapped, "sector": 0}]
    t0 = time.perf_counter()
    with pytest.raises(D.LsAmdError, match="group too large"):
        D.loadConfigFromDict(cfg)
    assert time.perf_counter() - t0 < 5.0
    # 4 x 4 x 4 torus: translations, one axis permutation cycle and one axis swap -> 64 * 6 = 384 elements
    idx = lambda x, y, z: (x % 4) + 4 * ((y % 4) + 4 * (z % 4))  # noqa: E731
    sites = [(x, y, z) for z in range(4) for y in range(4) for x in range(4)]
    cfg = M.heisenberg_chain_config(64)
    cfg["basis"]["hamming_weight"] = 32
    cfg["basis"]["symmetries"] = [
        {"permutation": [idx(x + 1, y, z) for x, y, z in sites], "sector": 0},
        {"permutation": [idx(y, z, x) for x, y, z in sites], "sector": 0},
        {"permutation": [idx(y, x, z) for x, y, z in sites], "sector": 0},
    ]
    basis = D.loadConfigFromDict(cfg)
    basis = basis[0] if isinstance(basis, tuple) else basis
    assert basis.groupOrder() == 384


def test_random_permutation_networks():
    """Benes compilation for arbitrary generators (not just rotations / reflections)."""
    rs = np.random.RandomState(123)
    lib = _lib.load()
    for L in (5, 12, 31, 32, 33, 48, 64):
        perm = list(rs.permutation(L))
        spec = config.BasisSpec(number_sites=L, hamming_weight=-1, permutations=[[int(v) for v in perm]], sectors=[0])
        order = 1
        q = list(perm)
        while q != list(range(L)):
            q = [q[i] for i in perm]
            order += 1
            if order > 5000:
                break
        if order > 5000:
            continue
        basis = D.Basis.fromSpec(spec)
        assert basis.groupOrder() == order
        mask = (1 << L) - 1
        states = [int(rs.randint(0, 2**62, dtype=np.int64)) * 4 + int(rs.randint(0, 4)) & mask for _ in range(8)]
        actions = set()
        for g in range(order):
            actions.add(tuple(int(lib.ls_amd_basis_apply_group_element(basis.payload, g, C.c_uint64(s))) for s in states))
        # the generator itself must be among the elements
        assert tuple(M.apply_perm(perm, s) for s in states) in actions
        # every element is a bijection on bits: popcount preserved
        for act in actions:
            assert all(bin(a).count("1") == bin(s).count("1") for a, s in zip(act, states))


@pytest.mark.parametrize("name", CHECK_MODELS + ["heisenberg_chain_32", "heisenberg_chain_40_symm", "heisenberg_square_6x6"])
def test_term_tables_equal_oracle_terms(name):
    """config.py (symbolic Pauli compile) + C grouping vs the oracle's matrix-element compile: two
    different decompositions, same operator -- compared as functions on random basis states."""
    cfg = model_config(name)
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    m = M.model_from_config(cfg)
    diag, off = product_terms(h)
    assert h.numberDiagTerms() == len(diag)
    assert h.numberOffDiagTerms() == m.max_off_diag == len(set(t[3] for t in off))
    assert h.isHermitian and h.isReal
    rs = np.random.RandomState(9)
    for _ in range(40):
        a = int(rs.randint(0, 2**62, dtype=np.int64)) & m.mask
        d_want = M._apply_terms(m.diag, a).get(a, 0)
        d_got = apply_terms_python(diag, a).get(a, 0)
        assert abs(d_want - d_got) < 1e-12
        want = {k: v for k, v in M._apply_terms(m.offdiag, a).items() if v != 0}
        got = apply_terms_python(off, a)
        assert set(want) == set(got)
        for k in want:
            assert abs(want[k] - got[k]) < 1e-12


def test_non_hermitian_and_complex_operators_are_flagged():
    spec = config.BasisSpec(number_sites=4, hamming_weight=-1)
    basis = D.Basis.fromSpec(spec)
    raise_only = config.OperatorSpec(config.monomial_terms("σ⁺₀ σ⁻₁", [0, 1]))
    op = D.Operator.fromSpec(basis, raise_only)
    assert not op.isHermitian and op.isReal
    herm = config.OperatorSpec(config.monomial_terms("σ⁺₀ σ⁻₁", [0, 1]) + config.monomial_terms("σ⁻₀ σ⁺₁", [0, 1]))
    op2 = D.Operator.fromSpec(basis, herm)
    assert op2.isHermitian
    # sigma^x sigma^y is Hermitian with purely imaginary matrix elements
    op3 = D.Operator.fromSpec(basis, config.OperatorSpec(config.monomial_terms("σˣ₀ σʸ₁", [2, 3])))
    assert op3.isHermitian and not op3.isReal
    # dense cross-check of the symbolic compile on a non-trivial monomial
    k, mat = M.local_matrix("0.5 × σ⁺₀ σᶻ₁ σʸ₂")
    terms = config.monomial_terms("0.5 × σ⁺₀ σᶻ₁ σʸ₂", [0, 1, 2])
    for b in range(8):
        got = apply_terms_python(terms, b)
        for a in range(8):
            assert abs(mat[a, b] - got.get(a, 0)) < 1e-14


def test_host_scalars():
    lib = _lib.load()
    assert D.hash64_01(0x1f0) == 0xc56a3fa16c3a7f04
    assert D.localeIdxOf(0x155, 8) == 2 and D.localeIdxOf(0x155, 3) == 2
    for i in range(300):
        s = int(lib.ls_hs_fixed_hamming_index_to_state(i, 6))
        assert bin(s).count("1") == 6 and int(lib.ls_hs_fixed_hamming_state_to_index(C.c_uint64(s))) == i
    # ... and over the whole range the path uses: up to 64 sites, weights up to 33 (ls_hs_fixed_hamming_state_to_index, FFI.chpl:165)
    import math
    import random

    rnd = random.Random(5)
    for _ in range(3000):
        n = rnd.randint(1, 64)
        k = rnd.randint(1, min(n, 33))
        s = sum(1 << b for b in rnd.sample(range(n), k))
        want, t, j = 0, s, 1
        while t:
            want += math.comb((t & -t).bit_length() - 1, j)
            j += 1
            t &= t - 1
        if want >= 2 ** 63:
            continue
        assert int(lib.ls_hs_fixed_hamming_state_to_index(C.c_uint64(s))) == want
        assert int(lib.ls_hs_fixed_hamming_index_to_state(want, k)) & (2 ** 64 - 1) == s
    basis = D.loadConfigFromDict(model_config("heisenberg_chain_10"))
    assert basis.minStateEstimate() == 31 and basis.maxStateEstimate() == 496  # Appendix B


def test_kernel_table_registration():
    """ls_chpl_init_kernels fills the ls_chpl_kernels table (LatticeSymmetries.chpl:16-29)."""
    lib = _lib.load()
    lib.ls_chpl_init_kernels()
    table = (C.c_void_p * 4).from_address(lib.ls_hs_internal_get_chpl_kernels())
    want = [lib.ls_chpl_enumerate_representatives, lib.ls_chpl_operator_apply_off_diag,
            lib.ls_chpl_operator_apply_diag, lib.ls_chpl_matrix_vector_product]
    for got, fn in zip(table, want):
        assert got == C.cast(fn, C.c_void_p).value


def test_set_representatives_and_halts():
    basis, h = D.loadConfigFromDict(model_config("heisenberg_chain_10"), hamiltonian=True)
    with pytest.raises(D.LsAmdError, match="basis is not built"):
        h.basis.representatives()
    reps = oracle_for("heisenberg_chain_10").enumerate()
    h.basis.uncheckedSetRepresentatives(reps)
    assert np.array_equal(h.basis.representatives(), reps)
    lib = _lib.load()
    # numVectors != 1 halts (DMV:1101-1102); the handler turns the halt into an exception
    x = np.zeros(126)
    y = np.zeros(126)
    lib.ls_chpl_matrix_vector_product(h.payload, 2, x.ctypes.data_as(_lib.c_f64p), y.ctypes.data_as(_lib.c_f64p))
    with pytest.raises(D.LsAmdError, match="more than 1 vector"):
        _lib.raise_pending_halt()


def test_primme_view_matches_primme_header(have_reference, tmp_path):
    """ls_primme_params_view (include/ls_chpl.h) must place nLocal and matrix where PRIMME 3.1 does."""
    import subprocess

    if not have_reference:
        pytest.skip("reference (primme_headers/) not mounted")
    src = tmp_path / "off.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "ls_chpl.h"\n#include "primme.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(ls_primme_params_view,n), offsetof(primme_params,n),'
        ' offsetof(ls_primme_params_view,nLocal), offsetof(primme_params,nLocal),'
        ' offsetof(ls_primme_params_view,matrix), offsetof(primme_params,matrix));return 0;}\n')
    exe = tmp_path / "off"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", "/root/reference/primme_headers", str(src), "-o", str(exe)])
    v = [int(t) for t in subprocess.check_output([str(exe)]).split()]
    assert v[0] == v[1] and v[2] == v[3] and v[4] == v[5] == 264


def test_hot_kernel_register_budget():
    """Register hygiene of the row kernels, read from the compiler's resource report (scripts/kernel_resources.py):
    * NO hot kernel spills: ScratchSize == 0 for every instantiation of k_chain_t, k_pairs_t, k_pull_t, k_pull_gather, k_tile,
      k_tile_pull, k_direct and k_scatter (round 3 shipped 20-40 bytes per lane in three of them);
    * one block per tile is launched, so what matters is how many 256-thread blocks a CU ADMITS: the SGPR file admits
      floor(800 / (ceil(sgpr / 16) * 16 + 16)) of them (MI355X_MICROARCH.md; at 98 SGPRs the seventh block does not fit while the
      occupancy API still answers 7 -- measured r2 as 10.7 -> 13.1 ms on chain_32), and that must not be fewer than LDS and
      the VGPR file allow;
    * the library stays small: <= 300 device functions (round 3: 440, of which 225 were hipCUB trampolines and 100+
      instantiations that no plan can reach)."""
    import shutil
    import sys

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kernel_resources

    stats = kernel_resources.resources()
    assert 40 <= len(stats) <= 300, len(stats)
    hot_prefixes = ("_Z9k_chain_tI", "_Z9k_pairs_tI", "_Z8k_pull_tI", "_Z13k_pull_gatherI", "_Z6k_tileI", "_Z11k_tile_pullI", "_Z8k_directI",
                    "_Z9k_scatterI")
    hot = {k: v for k, v in stats.items() if k.startswith(hot_prefixes)}
    assert len(hot) >= 100, len(hot)
    spilling = {k: v["scratch"] for k, v in hot.items() if v["scratch"]}
    assert not spilling, spilling
    for name, v in stats.items():  # the generic row kernels run as persistent grids sized by the occupancy API: keep them where it is right
        if name.startswith("_Z8k_directIjLb0ELi1E") or name.startswith("_Z8k_directIjLb1ELi1E"):
            assert v["sgpr"] <= 80 and v["occ"] == 8, (name, v)
        # ... and so do the one-row-per-lane exchange-pair kernels of round 6 (k_pairs_row, k_pairs_site): no spills, <= 80 SGPRs
        if name.startswith(("_Z11k_pairs_rowI", "_Z12k_pairs_siteI")):
            assert v["sgpr"] <= 80 and v["scratch"] == 0, (name, v)
    assert sum(1 for k in stats if k.startswith("_Z11k_pairs_rowI")) == 4 and sum(1 for k in stats if k.startswith("_Z12k_pairs_siteI")) == 16
    chain = {k: v for k, v in stats.items() if k.startswith("_Z9k_chain_tI")}
    # six instantiations: (u32, u32) f64 on fused records [headline] and c128; (u64, u32) and (u64, u64) f64 / c128
    assert len(chain) == 6, sorted(chain)
    for name, v in chain.items():
        by_sgpr = 800 // (-(-v["sgpr"] // 16) * 16 + 16)
        # static window + the launch-time image (binomials [rows][weight + 2] in the rank type + the 3840-byte near-pair
        # table), priced at half filling of the widest basis the instantiation serves
        wide_state, wide_rank = name.startswith("_Z9k_chain_tIm"), name.startswith("_Z9k_chain_tImm")
        rows = 64 if wide_state else 32
        image = -(-(rows * (rows // 2 + 2) * (8 if wide_rank else 4)) // 16) * 16 + 3840
        by_lds = (160 * 1024) // (v["lds"] + image)
        assert by_sgpr >= min(by_lds, v["occ"], 8), (name, v, image)
    head = chain[[k for k in chain if k.startswith("_Z9k_chain_tIjjLb0ELi1024ELb1E")][0]]
    assert head["sgpr"] <= 96 and head["vgpr"] <= 72 and head["occ"] >= 7, head
    assert (160 * 1024) // (head["lds"] + 32 * 18 * 4 + 3840) >= 7
    c128 = chain[[k for k in chain if k.startswith("_Z9k_chain_tIjjLb1ELi512ELb0E")][0]]
    assert c128["sgpr"] <= 112 and c128["vgpr"] <= 96 and c128["occ"] >= 5, c128
    # the projected-basis kernels of the BASELINE configs (trivial sector, one amplitude, f64): fused and resolve
    for inst in ("_Z8k_pull_tImLi0ELi0ELb0ELi0E", "_Z8k_pull_tImLi0ELi0ELb0ELi1E", "_Z8k_pull_tIjLi0ELi0ELb0ELi0E"):
        v = stats[[k for k in stats if k.startswith(inst)][0]]
        assert v["vgpr"] <= 80 and v["occ"] >= 6 and v["lds"] <= 24 * 1024, (inst, v)  # >= 6 blocks per CU by registers (granule 8) and by LDS
    pairs = stats[[k for k in stats if k.startswith("_Z9k_pairs_tIjLb0ELi1024E")][0]]  # 32-bit states (<= 32 sites)
    assert pairs["vgpr"] <= 84 and pairs["sgpr"] <= 96 and pairs["lds"] <= 27 * 1024, pairs
    wide = stats[[k for k in stats if k.startswith("_Z9k_pairs_tImLb0ELi1024E")][0]]   # 64-bit states (33..64 sites, round 6)
    assert wide["vgpr"] <= 84 and wide["sgpr"] <= 96 and wide["scratch"] == 0 and wide["lds"] <= 30 * 1024, wide
    # the eigensolver's fused Gram-Schmidt sweep (csrc/orth.hip): 33 accumulators per thread that must stay in registers (one
    # run-time index into them sent the array to scratch: 3.9 instead of 5.8 TB/s)
    orth = kernel_resources.resources(source="orth.hip")
    assert len(orth) == 3 and all(v["scratch"] == 0 and v["occ"] >= 4 for v in orth.values()), orth  # two sweeps + the restart rotation


def _fixed_weight_states(L, hw):
    """all states of L bits with hw set, ascending (Gosper)"""
    import math

    n = math.comb(L, hw)
    out = np.empty(n, dtype=np.uint64)
    s = (1 << hw) - 1
    for i in range(n):
        out[i] = s
        c = s & -s
        r = s + c
        s = (((r ^ s) >> 2) // c) | r if c else 0
    return out


@pytest.mark.parametrize("n", [924, 12870, 31824, 184756, 1, 255, 1025])
@pytest.mark.parametrize("tile,chunk", [(256, 0), (1024, 0), (512, 0), (1024, 3), (256, 16), (512, 32), (1024, 32)])
def test_tile_map_is_a_partition_of_the_rows(n, tile, chunk):
    """lsk_tilemap: every row in exactly one tile, tiles <= tile_rows, balanced over the 8 XCD lists; chunked dealing puts
    tile q into list (q // chunk) % 8 in ascending order."""
    lib = _lib.load()
    ptr = C.POINTER(C.c_uint64)()
    slots = lib.ls_amd_test_tilemap(n, tile, chunk, C.byref(ptr))
    assert slots >= 0
    e = np.ctypeslib.as_array(ptr, shape=(8 * max(slots, 1),)).copy()
    lib.ls_amd_test_free(ptr)
    rows = e & np.uint64((1 << 48) - 1)
    cnt = (e >> np.uint64(48)).astype(np.int64)
    live = cnt > 0
    assert cnt.max() <= tile and cnt[live].min() >= 1
    cover = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cover, rows[live].astype(np.int64), 1)
    np.add.at(cover, rows[live].astype(np.int64) + cnt[live], -1)
    assert np.array_equal(np.cumsum(cover)[:n], np.ones(n, dtype=np.int64))
    per_xcd = cnt.reshape(8, -1).sum(axis=1)
    assert per_xcd.max() - per_xcd.min() <= 2 * tile * max(chunk, 1)
    if chunk:
        lists = rows.reshape(8, -1)
        for k in range(8):
            q = (lists[k][cnt.reshape(8, -1)[k] > 0] // np.uint64(tile)).astype(np.int64)
            assert np.all((q // chunk) % 8 == k) and np.all(np.diff(q) > 0)


class _ForeignBasis(C.Structure):
    """exactly the reference's struct prefix (/root/reference/src/FFI.chpl:94-105) followed by bytes that are NOT
    this library's: what an ls_hs_basis built by lattice-symmetries-haskell looks like from here"""
    _fields_ = _lib.LsHsBasis._fields_ + [("other_stuff", C.c_uint8 * 64)]


class _ForeignOperator(C.Structure):
    _fields_ = [("basis", C.POINTER(_ForeignBasis)), ("off_diag_terms", C.POINTER(_lib.LsHsNonbranchingTerms)),
                ("diag_terms", C.POINTER(_lib.LsHsNonbranchingTerms)), ("other_stuff", C.c_uint8 * 64)]


def _foreign_copy(name):
    """(foreign basis, foreign operator, keep-alive) built from one of this library's operators by copying ONLY the
    prefix fields and the public term arrays; the trailing bytes are poisoned"""
    import distributed_matvec_amd as D

    basis, h = D.loadConfigFromDict(model_config(name), hamiltonian=True)
    src = basis.payload.contents
    fb = _ForeignBasis()
    for f, _ in _lib.LsHsBasis._fields_:
        setattr(fb, f, getattr(src, f))
    fb.kernels = None
    fb.representatives = _lib.ChplExternalArray(None, 0, None)
    C.memset(C.addressof(fb) + _ForeignBasis.other_stuff.offset, 0xAB, 64)
    fo = _ForeignOperator()
    fo.basis = C.pointer(fb)
    fo.off_diag_terms = h.payload.contents.off_diag_terms
    fo.diag_terms = h.payload.contents.diag_terms
    C.memset(C.addressof(fo) + _ForeignOperator.other_stuff.offset, 0xCD, 64)
    return fb, fo, (basis, h)


@pytest.mark.parametrize("name", ["heisenberg_chain_10", "heisenberg_chain_12", "heisenberg_kagome_12_symm", "heisenberg_chain_24_symm"])
def test_adopt_foreign_operator_rebuilds_the_tables(name):
    """ls_amd_adopt_basis / ls_amd_adopt_operator (include/ls_amd.h): a struct WITHOUT any private field of ours is
    registered in the side table, and the rebuilt tables answer like the original operator's"""
    lib = _lib.load()
    fb, fo, (basis, h) = _foreign_copy(name)
    spec = basis.spec
    ng = len(spec.permutations)
    perms = (C.c_int * max(1, ng * spec.number_sites))(*[v for p in spec.permutations for v in p])
    sectors = (C.c_int * max(1, ng))(*spec.sectors)
    bp = C.cast(C.pointer(fb), C.POINTER(_lib.LsHsBasis))
    op = C.cast(C.pointer(fo), C.POINTER(_lib.LsHsOperator))
    # unknown objects are refused loudly
    _lib._pending_halt.clear()
    lib.ls_hs_basis_has_permutation_symmetries(bp)
    with pytest.raises(_lib.LsAmdError, match="ls_amd_adopt_basis"):
        _lib.raise_pending_halt()
    assert lib.ls_amd_adopt_operator(op) != 0 and b"ls_amd_adopt_basis" in lib.ls_amd_last_error()
    assert lib.ls_amd_adopt_basis(bp, ng, perms, sectors) == 0, lib.ls_amd_last_error()
    assert lib.ls_amd_adopt_basis(bp, ng, perms, sectors) != 0  # twice is an error
    assert lib.ls_amd_adopt_operator(op) == 0, lib.ls_amd_last_error()
    assert lib.ls_amd_basis_group_order(bp) == lib.ls_amd_basis_group_order(basis.payload)
    assert lib.ls_hs_basis_has_permutation_symmetries(bp) == lib.ls_hs_basis_has_permutation_symmetries(basis.payload)
    assert lib.ls_hs_operator_max_number_off_diag(op) == h.numberOffDiagTerms()
    assert bool(lib.ls_hs_operator_is_hermitian(op)) == h.isHermitian and bool(lib.ls_hs_operator_is_real(op)) == h.isReal
    # the poisoned tail was never touched
    assert bytes(fb.other_stuff) == b"\xab" * 64 and bytes(fo.other_stuff) == b"\xcd" * 64
    lib.ls_amd_release(C.cast(op, C.c_void_p))
    lib.ls_amd_release(C.cast(bp, C.c_void_p))
    _lib._pending_halt.clear()
    lib.ls_hs_operator_is_real(op)
    with pytest.raises(_lib.LsAmdError, match="ls_amd_adopt_operator"):
        _lib.raise_pending_halt()


def test_operator_shares_its_basis():
    """the operator holds a reference to the basis it was built on (no clone): representatives set on the basis
    afterwards are the operator's, and the basis outlives its Python owner while an operator uses it"""
    import gc

    import distributed_matvec_amd as D

    basis, h = D.loadConfigFromDict(model_config("heisenberg_chain_10"), hamiltonian=True)
    assert C.addressof(h.payload.contents.basis.contents) == C.addressof(basis.payload.contents)
    reps = np.array([31, 47, 55], dtype=np.uint64)
    basis.uncheckedSetRepresentatives(reps)
    assert h.basis.representatives().tolist() == [31, 47, 55]
    addr = C.addressof(basis.payload.contents)
    keep = basis._host_reps
    del basis
    gc.collect()
    assert C.addressof(h.payload.contents.basis.contents) == addr and h.basis.numberSites() == 10
    assert keep is not None


def test_near_window_search_of_the_pull_kernel():
    """k_tile_pull resolves partners that lie in the tile's neighbourhood of the sorted representatives with a binary
    search over saturating 32-bit offsets in LDS (window_find / window_offset in csrc/k_pull.hip, compiled for the host too):
    every member is found at its position, every non-member is a miss, gaps >= 2^32 - 1 fall back to the hash table."""
    lib = _lib.load()
    rng = np.random.RandomState(7)

    def find(reps, key):
        a = np.ascontiguousarray(reps, dtype=np.uint64)
        return lib.ls_amd_test_window_find(a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a), C.c_uint64(int(key)))

    for n in (1, 2, 3, 255, 256, 1023, 1024, 1025, 1279, 1280):
        reps = np.unique(rng.randint(0, (1 << 32) - 2, size=4 * n, dtype=np.int64).astype(np.uint64))[:n] + np.uint64(1 << 36)
        if len(reps) < n:
            continue
        for pos in {0, n - 1, n // 2, *rng.randint(0, n, size=16).tolist()}:
            assert find(reps, reps[pos]) == pos
        present = set(reps.tolist())
        for key in [int(reps[0]) - 1, int(reps[-1]) + 1, 0, (1 << 64) - 1, *[int(k) + 1 for k in reps[rng.randint(0, n, size=16)]]]:
            if key not in present and 0 <= key < (1 << 64):
                assert find(reps, key) == -1
    # a window whose span exceeds 32 bits: entries past the 2^32 - 1 cut are "absent" (the kernel then asks the table)
    base = 5 << 33
    reps = np.array([base, base + 7, base + (1 << 32) - 2, base + (1 << 32) - 1, base + (1 << 32), base + (1 << 40)], dtype=np.uint64)
    assert [find(reps, k) for k in reps] == [0, 1, 2, -1, -1, -1]
    assert find(reps, base + 8) == -1 and find(reps, base + (1 << 32) + 5) == -1
    assert find(reps[:1], base) == 0 and find(np.arange(1281, dtype=np.uint64), 3) == -2


@pytest.mark.parametrize("name", ["heisenberg_chain_24_symm", "heisenberg_chain_16", "heisenberg_square_4x4"])
def test_near_window_hash_set_of_the_indexed_pull_kernels(name):
    """k_pull_t stages the tile's neighbourhood of the sorted representatives as a two-way hash set in LDS (nw_* in
    csrc/k_pull.hip, compiled for the host too).  It may DROP an entry (full set: the partner then goes through the static index
    table), it must never answer with a wrong position or answer for a state that is not in the window; on real windows of
    768 consecutive representatives it answers for >= 93 % of them."""
    from helpers import oracle_reps

    lib = _lib.load()
    reps = np.ascontiguousarray(oracle_reps(name), dtype=np.uint64)
    rng = np.random.RandomState(11)

    def find(win, key):
        return lib.ls_amd_test_nw_find(win.ctypes.data_as(C.POINTER(C.c_uint64)), len(win), C.c_uint64(int(key)))

    answered = total = 0
    for start in [0, max(0, len(reps) - 768), *rng.randint(0, max(1, len(reps) - 768), size=6).tolist()]:
        win = np.ascontiguousarray(reps[start:start + 768])
        n = len(win)
        for pos in range(n):
            got = find(win, win[pos])
            assert got in (pos, -1), (start, pos, got)
            answered += got == pos
            total += 1
        present = set(win.tolist())
        for key in [int(win[0]) - 1, int(win[-1]) + 1, *[int(k) + 1 for k in win[rng.randint(0, n, size=64)]],
                    *[int(k) ^ 6 for k in win[rng.randint(0, n, size=64)]]]:
            if key not in present and 0 <= key < (1 << 64):
                assert find(win, key) == -1, (start, key)
    assert answered >= 0.93 * total, (answered, total)
    assert find(np.arange(1025, dtype=np.uint64), 3) == -2
    # offsets beyond 32 bits are never staged
    base = 5 << 33
    win = np.array([base, base + 7, base + (1 << 32) - 2, base + (1 << 32) - 1, base + (1 << 40)], dtype=np.uint64)
    assert [find(win, k) for k in win] == [0, 1, 2, -1, -1]


@pytest.mark.parametrize("name", ["heisenberg_chain_24_symm", "heisenberg_chain_16", "heisenberg_kagome_12_symm"])
def test_static_index_table_finds_every_representative(name):
    """lsk_gtab (indexed pull mode): the key is not stored -- bucket and tag come from an L-bit bijection -- so the test is
    that every representative is found with ITS payload, that states outside the basis are reported absent, and that the
    table has the advertised shape (two 8-byte entries per 16-byte bucket, load <= 0.5, tag + displacement + payload in 64
    bits).  Host mirror of the device build / lookup (same placement rule, same probe sequence)."""
    from helpers import oracle_reps

    lib = _lib.load()
    reps = np.ascontiguousarray(oracle_reps(name), dtype=np.uint64)
    n = len(reps)
    L = int(model_config(name)["basis"]["number_spins"])
    bb = lib.ls_amd_test_gtab_bits(L, n)
    assert bb >= 2 and (2 << bb) >= 2 * n and L - bb <= 24 and (bb <= 3 or (2 << (bb - 1)) < 2 * n or bb == L - 24)
    rs = np.random.RandomState(7)
    payload = rs.permutation(n).astype(np.uint32)  # a slot permutation, as the replicated-x exchange has it
    ent = C.POINTER(C.c_uint64)()
    assert lib.ls_amd_test_gtab_build(L, bb, n, reps.ctypes.data_as(C.POINTER(C.c_uint64)), payload.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      C.byref(ent)) == 0
    try:
        got = np.array([lib.ls_amd_test_gtab_find(L, bb, ent, C.c_uint64(int(k))) for k in reps])
        assert np.array_equal(got, payload.astype(np.int64))
        present = set(int(k) for k in reps)
        absent = [int(k) for k in rs.randint(0, 1 << L, size=4000, dtype=np.int64) if int(k) not in present]
        assert len(absent) > 100
        assert all(lib.ls_amd_test_gtab_find(L, bb, ent, C.c_uint64(k)) == -1 for k in absent)
        table = np.ctypeslib.as_array(ent, shape=(2 << bb,))
        used = table != np.uint64(0xFFFFFFFFFFFFFFFF)
        assert int(used.sum()) == n
        disp = (table[used] >> np.uint64(32)) & np.uint64(0xFF)
        assert float((disp > 0).mean()) < 0.2 and int(disp.max()) < 32  # the hash spreads the orbit minima
    finally:
        lib.ls_amd_test_free(ent)


def test_static_index_table_shapes():
    lib = _lib.load()
    assert lib.ls_amd_test_gtab_bits(40, 861725794) == 30  # chain_40_symm: 2^31 entries of 8 bytes = 17 GB, tag 10 bits
    assert lib.ls_amd_test_gtab_bits(36, 63068876) == 26
    assert lib.ls_amd_test_gtab_bits(48, 1000) == 24  # few keys of many bits: the tag must still fit 24 bits (268 MB)
    assert lib.ls_amd_test_gtab_bits(64, 1000) == -1  # ... and does not beyond 1 TiB: those plans keep the value table
    assert lib.ls_amd_test_gtab_bits(64, 1 << 41) == -1
    assert lib.ls_amd_test_gtab_bits(10, 13) >= 3


@pytest.mark.parametrize("elem,ldsp", [(8, 12), (16, 11)])
def test_chain_near_pair_table(elem, ldsp):
    """The near-pair table of the staged row kernel against the rule it replaces (k_chain_t, per-pair form): pair lo of state a
    is anti-aligned when bits lo, lo + 1 differ; its partner sits C(lo, k) rows above (bit lo set) or below (bit lo + 1 set),
    k = set bits of a below lo.  Aligned pairs and pairs >= ldsp hold 0x7000, which clamps to the zero slot of the window."""
    import math

    L = _lib.load()
    tab = np.zeros(480 * 4, dtype=np.int16)
    assert L.ls_amd_test_chain_near_table(elem, ldsp, tab.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    base = (0, 32, 192)
    rng = np.random.default_rng(5)
    states = np.concatenate([np.arange(1 << 13, dtype=np.uint64), rng.integers(0, 1 << 32, 4000, dtype=np.uint64)])
    for a in states.tolist():
        for g in range(3):
            kidx = bin(a & ((1 << (4 * g)) - 1)).count("1")
            entry = base[g] + 32 * kidx + ((a >> (4 * g)) & 31)
            for p in range(4):
                lo = 4 * g + p
                bit, nxt = (a >> lo) & 1, (a >> (lo + 1)) & 1
                k = bin(a & ((1 << lo) - 1)).count("1")
                want = 0x7000
                if lo < ldsp and bit != nxt:
                    want = elem * math.comb(lo, k) * (1 if bit else -1)
                assert int(tab[4 * entry + p]) == want, (hex(a), lo)
    # every real displacement stays inside the halo of the window (512 rows for f64, 256 for c128)
    real = tab[tab != 0x7000]
    assert np.abs(real).max() <= elem * (512 if elem == 8 else 256)


def _orbit_min(a, L, inv, reflect):
    mask = (1 << L) - 1
    words = [a] + ([a ^ mask] if inv else [])
    if reflect:
        words += [int(format(w, f"0{L}b")[::-1], 2) for w in list(words)]
    best = mask
    for w in words:
        for s in range(L):
            best = min(best, ((w << s) | (w >> (L - s))) & mask)
    return best


@pytest.mark.parametrize("inv,reflect", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_k4_trivial_sector_orbit_minimum(inv, reflect):
    """K4 of the trivial sector on rings (mode 3: longest-run candidates; the run length by doubling and refinement) against
    the brute-force minimum over all translations / reflections / spin flips: every state of the small rings (including the
    < 9-site ones that take the step-by-step path), random states of every weight on 32-, 33-, 36-, 40-, 63- and 64-site
    rings, and states with long runs (>= 8: the step-by-step tail)."""
    lib = _lib.load()
    f = lib.ls_amd_test_rep_trivial_dihedral
    for L in (2, 3, 5, 8, 9, 10, 13):
        for a in range(1 << L):
            assert f(a, L, inv, reflect) == _orbit_min(a, L, inv, reflect), (L, a)
    rng = np.random.default_rng(11)
    for L in (16, 31, 32, 33, 36, 40, 63, 64):
        samples = []
        for w in (1, 2, L // 4, L // 2, L - 3, L - 1):
            for _ in range(25):
                bits = rng.choice(L, size=w, replace=False)
                samples.append(sum(1 << int(b) for b in bits))
        for run in (8, 9, 15, L - 2):  # long runs of zeros and of ones, at the seam and inside
            for shift in (0, 1, L - 3):
                block = ((1 << run) - 1)
                z = ((block << shift) | (block >> (L - shift))) & ((1 << L) - 1) if shift else block
                samples += [z, z ^ ((1 << L) - 1), z | (1 << ((shift + run + 2) % L))]
        for a in samples:
            assert f(a, L, inv, reflect) == _orbit_min(a, L, inv, reflect), (L, hex(a))


def test_xcd_chunked_tile_map_is_a_permutation():
    """pull_tile_of_block (k_pull_t, k_pull_gather, k_scatter): blocks b = x (mod 8) walk C consecutive tiles of XCD x before
    they jump by 8 C; the last incomplete round keeps the identity.  Every tile must be visited exactly once, and inside a
    full round the blocks of one XCD must see consecutive tiles."""
    lib = _lib.load()
    lib.lsk_test_pull_tile_of_block.restype = C.c_int64
    lib.lsk_test_pull_tile_of_block.argtypes = [C.c_int64, C.c_int64, C.c_int]
    for n in (1, 7, 8, 9, 63, 64, 65, 2047, 2048, 2049, 5000, 8 * 256 * 3, 8 * 256 * 3 + 777):
        for c in (0, 1, 4, 64, 256):
            seen = sorted(int(lib.lsk_test_pull_tile_of_block(b, n, c)) for b in range(n))
            assert seen == list(range(n)), (n, c)
    n, c = 8 * 64 * 2 + 100, 64
    for x in range(8):
        tiles = [int(lib.lsk_test_pull_tile_of_block(b, n, c)) for b in range(x, 8 * 64, 8)]  # the first 64 blocks of XCD x
        assert tiles == list(range(x * 64, x * 64 + 64)), (x, tiles[:4])


@pytest.mark.parametrize("inv", [0, 1])
def test_torus_minimum_by_row_table(inv):
    """torus_min (K4 mode 4 since mid round 4): the minimum over the tw x th translations (and the complement) from one table
    look-up per row + the few (row, rotation) candidates that put the smallest row on top == the minimum over all tw * th
    translations done one by one, on random words, periodic rows, identical rows, the empty and the full word; and a `best`
    handed in (an earlier coset's result) is only ever lowered.  Host mirror of the device routine (same code)."""
    lib = _lib.load()
    lib.lsk_torus_rowtab.restype = C.c_int
    lib.lsk_torus_rowtab.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    lib.lsk_test_torus_min.restype = C.c_uint64
    lib.lsk_test_torus_min.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_uint64]
    rng = np.random.RandomState(5 + inv)

    def brute(v, tw, th):
        L = tw * th
        mask = (1 << L) - 1
        rmask = (1 << tw) - 1
        rows = [(v >> (k * tw)) & rmask for k in range(th)]
        best = mask
        for i in range(tw):
            rr = [((r << i) | (r >> (tw - i))) & rmask for r in rows]
            for j in range(th):
                c = sum(rr[k] << (((k + j) % th) * tw) for k in range(th))
                best = min(best, c, (c ^ mask) if inv else c)
        return best

    for tw, th in ((2, 2), (2, 8), (3, 4), (4, 4), (4, 6), (5, 5), (6, 6), (6, 4), (7, 9), (8, 8), (8, 3), (1, 7)):
        L = tw * th
        tab = (C.c_uint32 * (1 << tw))()
        assert lib.lsk_torus_rowtab(tw, tab) == 0
        mask = (1 << L) - 1
        samples = [0, mask, 1, 1 << (L - 1), int("01" * 64, 2) & mask, int("0011" * 32, 2) & mask]
        row = int(rng.randint(0, 1 << tw))
        samples.append(sum(row << (k * tw) for k in range(th)))  # identical rows
        for _ in range(150):
            w = int(rng.randint(0, L + 1))
            samples.append(int(sum(1 << int(b) for b in rng.permutation(L)[:w])))
        for v in samples:
            want = brute(v, tw, th)
            got = int(lib.lsk_test_torus_min(C.c_uint64(v), L, tw, inv, tab, C.c_uint64((1 << 64) - 1)))
            assert got == want, (tw, th, hex(v), hex(got), hex(want))
            for prev in (want, max(want - 1, 0), min(want + 1, mask), int(rng.randint(0, 1 << 30)) & mask):
                got = int(lib.lsk_test_torus_min(C.c_uint64(v), L, tw, inv, tab, C.c_uint64(prev)))
                assert got == min(prev, want), (tw, th, hex(v), hex(prev))
    assert lib.lsk_torus_rowtab(9, (C.c_uint32 * 512)()) == -1


@pytest.mark.parametrize("name,tw,cosets", [("heisenberg_square_4x4", 4, 8), ("heisenberg_square_6x6", 6, 8),
                                             ("heisenberg_kagome_12_symm", None, None), ("heisenberg_chain_24_symm", None, None)])
def test_lattice_group_orbit_minimum_by_translation_cosets(name, tw, cosets):
    """K4 mode 4 (trivial sectors of lattice groups, e.g. the reference's benchmark model heisenberg_square_6x6: 36
    translations x the 8 elements of D4, x the global spin flip): the orbit minimum from 8 compiled networks + 288 cheap
    translation steps equals the minimum over all 288 (576) images, on random states of the right weight.  Host mirror of
    the device routine (same bit operations)."""
    lib = _lib.load()
    cfg = model_config(name)
    basis = D.loadConfigFromDict(cfg)
    nc = C.c_int(0)
    w = lib.ls_amd_test_translation_cosets(basis.payload, C.byref(nc))
    if tw is not None:
        assert (w, nc.value) == (tw, cosets), (w, nc.value)
    if w <= 0:
        assert int(lib.ls_amd_test_rep_by_cosets(basis.payload, C.c_uint64(5))) == (1 << 64) - 1
        return
    L = basis.numberSites()
    order = basis.groupOrder()
    assert order == nc.value * L
    inv = basis.spinInversion()
    mask = (1 << L) - 1
    hw = cfg["basis"].get("hamming_weight")
    rng = np.random.RandomState(3)
    for _ in range(60):
        bits = rng.permutation(L)[: (hw if hw is not None else rng.randint(1, L))]
        a = int(sum(1 << int(b) for b in bits))
        imgs = [int(lib.ls_amd_basis_apply_group_element(basis.payload, g, C.c_uint64(a))) for g in range(order)]
        if inv != 0:
            imgs = [min(v, v ^ mask) for v in imgs]
        assert int(lib.ls_amd_test_rep_by_cosets(basis.payload, C.c_uint64(a))) == min(imgs), (name, hex(a))


def _torus_config(tw, th, weight, gens, inv):
    """spin basis on a tw x th torus (site = y tw + x) whose symmetry group is generated by the two translations and `gens`
    (subset of "r" rows reversed, "o" row order reversed, "t" transpose), all in the trivial sector"""
    L = tw * th
    site = lambda y, x: (y % th) * tw + x % tw  # noqa: E731
    perms = {"tx": [site(s // tw, s % tw + 1) for s in range(L)], "ty": [site(s // tw + 1, s % tw) for s in range(L)],
             "r": [site(s // tw, tw - 1 - s % tw) for s in range(L)], "o": [site(th - 1 - s // tw, s % tw) for s in range(L)],
             "ro": [site(th - 1 - s // tw, tw - 1 - s % tw) for s in range(L)], "t": [site(s % tw, s // tw) for s in range(L)]}
    basis = {"number_spins": L, "hamming_weight": weight,
             "symmetries": [{"permutation": perms[g], "sector": 0} for g in ["tx", "ty"] + list(gens)]}
    if inv:
        basis["spin_inversion"] = 1
    return {"basis": basis}


@pytest.mark.parametrize("case", ["6x6/r,o,t/inv/0xff", "4x4/r,o,t/inv/0xff", "6x4/r,o/inv/0x0f", "5x3/r,o//0x0f", "4x4/t//0x11", "6x3/ro//0x09",
                                  "4x4/r//0x03", "5x5/r,t//0xff", "8x4/o/inv/0x05", "3x3/ro,t//0x99", "7x2/r,o/inv/0x03", "4x6/o//0x05"])
def test_lattice_group_orbit_minimum_factorised(case, monkeypatch):
    """K4 mode 5: when the cosets of the translation subgroup are the point group of the torus -- D2 = {1, r, o, r o}, on a square
    torus D4 = D2 x {1, transpose}, or any subgroup -- only the transpose is a compiled network: r is delta swaps, r o a bit
    reversal, and ONE pass over the rows prices all four images of a word through a row table that carries the reversed row's
    fields as well (torus_min_d2).  The mask of images found is the expected one, and the routine (host mirror: same bit
    operations as the device code) equals the brute-force minimum over the whole group on random states -- and equals mode 4."""
    shape, gens, inv, want_mask = case.split("/")
    tw, th = (int(v) for v in shape.split("x"))
    L = tw * th
    cfg = _torus_config(tw, th, None if L > 30 and False else L // 2, gens.split(","), bool(inv))
    lib = _lib.load()
    basis = D.loadConfigFromDict(cfg)
    nc = C.c_int(0)
    w = lib.ls_amd_test_translation_cosets(basis.payload, C.byref(nc))
    order = basis.groupOrder()
    assert w > 0 and order == nc.value * L, (w, nc.value, order)
    mask_found = lib.ls_amd_test_d4_mask(basis.payload)
    if w == tw:  # (a 4 x 6 torus is found as the 2-wide one first when that qualifies: any factorisation that is found is fine)
        assert mask_found == int(want_mask, 16), hex(mask_found)
        assert bin(mask_found).count("1") == nc.value
    full = (1 << L) - 1
    rng = np.random.RandomState(11)
    states = [int(sum(1 << int(b) for b in rng.permutation(L)[: L // 2])) for _ in range(80)]
    states += [full >> (L // 2), 0x5555555555555555 & full, 1, full ^ 1]
    for a in states:
        imgs = [int(lib.ls_amd_basis_apply_group_element(basis.payload, g, C.c_uint64(a))) for g in range(order)]
        if inv:
            imgs = [min(v, v ^ full) for v in imgs]
        got = int(lib.ls_amd_test_rep_by_cosets(basis.payload, C.c_uint64(a)))
        assert got == min(imgs), (case, hex(a), hex(got), hex(min(imgs)))
    if mask_found:
        monkeypatch.setenv("LS_AMD_K4", "cosets")  # mode 4 on the same states
        for a in states[:20]:
            imgs = [int(lib.ls_amd_basis_apply_group_element(basis.payload, g, C.c_uint64(a))) for g in range(order)]
            if inv:
                imgs = [min(v, v ^ full) for v in imgs]
            assert int(lib.ls_amd_test_rep_by_cosets(basis.payload, C.c_uint64(a))) == min(imgs)


def test_reference_lattice_models_take_the_factorised_form():
    """heisenberg_square_4x4 / _6x6 (the reference's benchmark model, Makefile:86,109): the 8 cosets are D4 -> mode 5"""
    lib = _lib.load()
    for name in ("heisenberg_square_4x4", "heisenberg_square_6x6"):
        basis = D.loadConfigFromDict(model_config(name))
        assert lib.ls_amd_test_d4_mask(basis.payload) == 0xff, name
    basis = D.loadConfigFromDict(model_config("heisenberg_chain_24_symm"))
    assert lib.ls_amd_test_d4_mask(basis.payload) == 0


def test_exchange_choice_by_per_rank_memory(monkeypatch):
    """the replicated-x exchange keeps O(N) tables on EVERY rank, the packets O(N / P); `auto` takes the former
    while it fits and falls back to the latter -- the strategy that scales in capacity -- when it does not."""
    from distributed_matvec_amd.distributed import choose_exchange, exchange_memory_estimate

    monkeypatch.delenv("LS_AMD_EXCHANGE_HBM_CEILING", raising=False)
    n40, n32 = 861725794, 601080390
    e = exchange_memory_estimate(n40, n40 // 8, 8, 8, True, 40, krylov_vectors=16)
    assert 50e9 < e["replicated"] < 80e9 and e["packets"] - e["vectors"] < 0.5 * (e["replicated"] - e["vectors"])
    assert choose_exchange(True, e, 288 << 30) == "replicated"
    assert choose_exchange(True, e, 64 << 30) == "packets"          # a smaller part: the tables do not fit, the packets do
    assert choose_exchange(False, e, 288 << 30) == "packets"         # not Hermitian: no pull form at all
    u = exchange_memory_estimate(n32, n32 // 8, 8, 8, False, 32)
    assert u["replicated"] > n32 * 29
    # memory per rank: replicated does not shrink with P; of the packets only the round buffers (LS_AMD_ROWS_PER_ROUND) stay
    e8, e64 = exchange_memory_estimate(n40, n40 // 8, 8, 8, True, 40), exchange_memory_estimate(n40, n40 // 64, 64, 8, True, 40)
    assert e64["replicated"] > 0.8 * e8["replicated"]
    small = exchange_memory_estimate(n40, n40 // 64, 64, 8, True, 40, rows_per_round=1 << 20)
    assert small["packets"] < 0.05 * small["replicated"]
    monkeypatch.setenv("LS_AMD_EXCHANGE_HBM_CEILING", str(10 << 30))
    assert choose_exchange(True, e, 288 << 30) == "packets"

"""The oracle against everything in-tree that pins it."""
import os

import numpy as np
import pytest

from helpers import SMALL_MODELS, golden, golden_vectors, model_config, oracle_for, oracle_reps
from oracle import c_oracle as CO
from oracle import model as M


def test_old_schema_matrices_pin_the_expressions():
    """/root/reference/data/old/heisenberg_chain_10.yaml:9-12 spells out the two-site matrix the new
    schema's three expressions must sum to."""
    checked = 0
    for name, entry in golden()["models"].items():
        if "old_config" not in entry:
            continue
        new_terms, old_terms = entry["config"]["hamiltonian"]["terms"], entry["old_config"]["hamiltonian"]["terms"]
        # group the new-schema expressions by their site list
        by_sites = {}
        for t in new_terms:
            k, mat = M.local_matrix(t["expression"])
            key = tuple(map(tuple, t["sites"]))
            by_sites[key] = by_sites.get(key, 0) + mat
        old_by_sites = {}
        for t in old_terms:
            mat = np.array(t["matrix"], dtype=complex)
            key = tuple(map(tuple, t["sites"]))
            old_by_sites[key] = old_by_sites.get(key, 0) + M._reverse_site_order(mat, 2)
        assert set(by_sites) == set(old_by_sites), name
        for key in by_sites:
            assert np.allclose(by_sites[key], old_by_sites[key], atol=1e-14), name
        checked += 1
    assert checked >= 10


def test_v1_representative_fixture():
    """/root/reference/v1/error.chpl:21"""
    want = golden()["fixtures"]["v1_error_chpl_21_representatives"]
    cfg = M.heisenberg_chain_config(10, symm=True)
    m = M.model_from_config(cfg)
    assert list(M.enumerate_representatives(m)) == want
    assert list(CO.COracle(m).enumerate()) == want


def test_generated_chain_configs_equal_reference_yaml(have_reference):
    if not have_reference:
        pytest.skip("reference not mounted")
    import yaml

    for L, symm in [(10, False), (24, False), (32, False), (24, True), (36, True), (40, True)]:
        name = f"heisenberg_chain_{L}" + ("_symm" if symm else "")
        ref = yaml.safe_load(open(f"/root/reference/data/{name}.yaml", encoding="utf-8"))
        mine = M.heisenberg_chain_config(L, symm=symm, spin_inversion=-1 if L == 10 else None)
        assert ref["basis"] == {**mine["basis"]}, name
        assert [(t["expression"], t["sites"]) for t in ref["hamiltonian"]["terms"]] == \
               [(t["expression"], t["sites"]) for t in mine["hamiltonian"]["terms"]], name


def test_known_answers_chain_10():
    """SURVEY.md Appendix B (independent dense construction by the surveyor)."""
    o = oracle_for("heisenberg_chain_10")
    reps = oracle_reps("heisenberg_chain_10")
    assert len(reps) == 126 and list(reps[:5]) == [31, 47, 55, 59, 61] and reps[-1] == 496
    x = np.random.RandomState(42).rand(126) - 0.5
    y = o.local_matvec(reps, x)
    assert np.allclose(y[:3], [-0.6435132739429792, 0.8575238407016947, 2.0282782433657394], rtol=1e-13)
    assert abs(np.linalg.norm(y) - 19.418360955956267) < 1e-11
    assert abs(x @ y - (-1.8478819708706284)) < 1e-12
    _, H = M.dense_sector_matrix(model_config("heisenberg_chain_10"))
    assert abs(np.linalg.eigvalsh(H)[0] - (-18.061785417968)) < 1e-9


def test_known_answers_issue_01():
    reps = oracle_reps("issue_01")
    assert len(reps) == 452
    _, H = M.dense_sector_matrix(model_config("issue_01"))
    assert np.abs(H - H.conj().T).max() < 1e-12
    assert abs(np.linalg.eigvalsh(H)[0] - (-19.953385280506)) < 1e-9


def test_hash_vectors():
    """splitmix64 finaliser, /root/reference/src/StatesEnumeration.chpl:122-127 (Appendix B)."""
    table = [(0x1, 0x5692161d100b05e5), (0x1f, 0x540f172e046ef165), (0x1f0, 0xc56a3fa16c3a7f04),
             (0x155, 0x26f4658275ee4b52), (0xffff, 0xb2647e0ec6567475), (0xffff0000ffff, 0x109307d71522a337)]
    for x, h in table:
        assert M.hash64_01(x) == h
        assert int(CO.lib().lso_hash64_01(x)) == h
    assert M.hash64_01(0) == 0
    states = np.array([t[0] for t in table], dtype=np.uint64)
    assert list(CO.locale_idx_of(states, 8)) == [5, 5, 4, 2, 5, 7]
    assert list(CO.locale_idx_of(states, 3)) == [1, 0, 2, 2, 2, 1]


def test_chain_24_partition_sizes():
    reps = oracle_for("heisenberg_chain_24").enumerate()
    assert len(reps) == 2704156
    counts = np.bincount(CO.locale_idx_of(reps, 8), minlength=8)
    assert list(counts) == [338991, 338013, 338427, 337639, 338337, 337518, 337261, 337970]


@pytest.mark.parametrize("name", SMALL_MODELS)
def test_c_oracle_matches_dense_projector(name):
    """term tables + state_info scaling (C restatement) == explicit projector on the dense matrix."""
    o = oracle_for(name)
    reps = oracle_reps(name)
    cfg = model_config(name)
    if M.model_from_config(cfg).number_sites > 16:
        pytest.skip("dense oracle limited to 16 sites")
    r2, H = M.dense_sector_matrix(cfg)
    assert np.array_equal(r2, reps)
    assert np.abs(H.imag).max() < 1e-12
    x = np.random.RandomState(7).rand(len(reps)) - 0.5
    y = o.local_matvec(reps, x)
    want = H.real @ x
    assert np.abs(y - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    xc = x + 1j * (np.random.RandomState(8).rand(len(reps)) - 0.5)
    yc = o.local_matvec(reps, xc)
    assert np.abs(yc - H @ xc).max() <= 1e-12 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("L,sector", [(8, 1), (8, 3), (10, 3), (12, 5)])
def test_complex_characters_match_dense_projector(L, sector):
    from helpers import complex_translation_config

    cfg = complex_translation_config(L, sector)
    m = M.model_from_config(cfg)
    o = CO.COracle(m)
    reps = o.enumerate()
    r2, H = M.dense_sector_matrix(cfg)
    assert np.array_equal(r2, reps)
    assert np.abs(H - H.conj().T).max() < 1e-12
    rs = np.random.RandomState(3)
    x = (rs.rand(len(reps)) - 0.5) + 1j * (rs.rand(len(reps)) - 0.5)
    assert np.abs(o.local_matvec(reps, x) - H @ x).max() < 1e-12


def test_golden_vectors_against_c_oracle():
    """x regenerated by the recipe of input_for_matvec.py; y from the dense oracle (vectors.npz)."""
    v = golden_vectors()
    dims = dict(zip(v["dims_names"], v["dims_values"]))
    assert dims["heisenberg_chain_24"] == 2704156 and dims["heisenberg_chain_24_symm"] == 28968
    n_checked = 0
    for name in SMALL_MODELS:
        if name + "/y" not in v:
            continue
        reps = v[name + "/representatives"]
        assert np.array_equal(reps, oracle_reps(name))
        y = oracle_for(name).local_matvec(reps, v[name + "/x"])
        want = v[name + "/y"]
        assert np.abs(y - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), name
        n_checked += 1
    assert n_checked >= 8
    # Appendix B, golden-recipe row
    y10 = v["heisenberg_chain_10/y"]
    assert np.allclose(y10[:3], [-0.5015037972269847, -0.8684407077991885, -1.9586230522118682], rtol=1e-12)
    assert abs(y10.sum() - (-15.391509897327113)) < 1e-10


@pytest.mark.parametrize("name", ["heisenberg_chain_10", "heisenberg_chain_12", "issue_01", "heisenberg_square_4x4", "heisenberg_kagome_16"])
@pytest.mark.parametrize("P", [2, 3, 4, 8])
def test_partitioned_oracle_equals_single_locale(name, P):
    o = oracle_for(name)
    reps = oracle_reps(name)
    x = np.random.RandomState(11).rand(len(reps)) - 0.5
    y = o.local_matvec(reps, x)
    keys = CO.locale_idx_of(reps, P)
    rp, xp = CO.block_to_hashed(reps, keys, P), CO.block_to_hashed(x, keys, P)
    for r in rp:
        assert np.all(np.diff(r.astype(np.int64)) > 0) or len(r) <= 1  # every part ascending
    yb = CO.hashed_to_block(o.matvec_partitioned(rp, xp), keys)
    assert np.abs(yb - y).max() <= 1e-13 * max(1.0, np.abs(y).max())


def test_extern_restatements_small():
    """apply_off_diag_x1 output format (BatchedOperator.chpl:11-36) and state_index semantics."""
    o = oracle_for("heisenberg_chain_10")
    reps = oracle_reps("heisenberg_chain_10")
    betas, cs, offs = o.apply_off_diag(reps[:4])
    assert offs[0] == 0 and offs[-1] == len(betas) and np.all(np.diff(offs) >= 0)
    assert np.all(cs == 2.0)
    for i in range(4):
        for b in betas[offs[i]:offs[i + 1]]:
            assert bin(int(b) ^ int(reps[i])).count("1") == 2
    idx = CO.state_index(reps, np.array([31, 47, 496, 32, 0], dtype=np.uint64))
    assert list(idx) == [0, 1, 125, -1, -1]
    assert int(CO.lib().lso_fixed_hamming_state_to_index(0b10110)) == 6
    for i in range(200):
        s = int(CO.lib().lso_fixed_hamming_index_to_state(i, 5))
        assert bin(s).count("1") == 5 and int(CO.lib().lso_fixed_hamming_state_to_index(s)) == i

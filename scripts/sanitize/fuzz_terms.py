"""Random arguments to ls_hs_create_spin_basis / ls_hs_create_operator_from_terms (bits outside the lattice, non-permutations,
   NaN amplitudes, negative counts) + clone / query / destroy, under ASAN + UBSAN (scripts/sanitize/run.sh)."""
import os
import sys, time, random, ctypes as C
sys.path.insert(0, "/root/repo")
import importlib
import numpy as np
_lib = importlib.import_module("distributed_matvec_amd._lib")
L = _lib.load()
rnd = random.Random(int(sys.argv[1])); budget = float(sys.argv[2])
t0 = time.time(); n = okb = oko = 0
def rbits(Ls, wild):
    top = 64 if wild else Ls
    k = rnd.choice([0, 1, 2, 2, 2, 3, 4])
    v = 0
    for _ in range(k): v |= 1 << rnd.randrange(top)
    return v
create_b = L.ls_hs_create_spin_basis; 
while time.time() - t0 < budget:
    n += 1
    Ls = rnd.choice([1, 2, 3, 4, 8, 12, 16, 31, 32, 33, 63, 64, 0, 65, -1]) if rnd.random() < 0.3 else rnd.randint(2, 24)
    hw = rnd.choice([-1, Ls // 2 if Ls > 0 else 0, rnd.randint(-3, 70)])
    inv = rnd.choice([0, 0, 1, -1, 2])
    ng = rnd.choice([0, 0, 1, 2, 3]) if Ls > 0 else 0
    perms = []
    for g in range(ng):
        kind = rnd.random()
        if kind < 0.4: p = [(i + 1) % Ls for i in range(Ls)]
        elif kind < 0.7: p = [Ls - 1 - i for i in range(Ls)]
        elif kind < 0.9:
            p = list(range(Ls)); rnd.shuffle(p)
        else: p = [rnd.randint(-2, Ls + 1) for _ in range(Ls)]
        perms += p
    sectors = [rnd.randint(-3, 70) for _ in range(ng)]
    pa = (C.c_int * max(1, len(perms)))(*perms); sa = (C.c_int * max(1, ng))(*sectors)
    with open(os.path.join(os.environ.get("FUZZ_DIR", "/tmp"), f"terms_current_{sys.argv[1]}.txt"), "w") as f: f.write(repr((Ls, hw, inv, ng, perms, sectors)))
    b = create_b(Ls, hw, inv, ng, pa, sa)
    if not b: continue
    okb += 1
    for _ in range(rnd.randint(1, 4)):
        nt = rnd.choice([0, 1, 2, 5, 40, 300, -1]) if rnd.random() < 0.5 else rnd.randint(1, 64)
        wild = rnd.random() < 0.2
        k = max(1, nt)
        v = np.array([rnd.choice([0.0, 1.0, -1.0, 0.5, rnd.uniform(-2, 2), float("nan"), float("inf")]) if rnd.random() < 0.9 else 0.0 for _ in range(2 * k)])
        if rnd.random() < 0.7: v[1::2] = 0.0
        m = np.array([rbits(Ls, wild) for _ in range(k)], dtype=np.uint64)
        r = np.array([int(mm) & rbits(Ls, wild) if rnd.random() < 0.8 else rbits(Ls, wild) for mm in m], dtype=np.uint64)
        x = np.array([rbits(Ls, wild) for _ in range(k)], dtype=np.uint64)
        s = np.array([rbits(Ls, wild) for _ in range(k)], dtype=np.uint64)
        with open(os.path.join(os.environ.get("FUZZ_DIR", "/tmp"), f"terms_current_{sys.argv[1]}.txt"), "a") as f: f.write("\n" + repr((nt, v.tolist(), m.tolist(), r.tolist(), x.tolist(), s.tolist())))
        op = L.ls_hs_create_operator_from_terms(b, nt, v.ctypes.data_as(C.POINTER(C.c_double)), m.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               r.ctypes.data_as(C.POINTER(C.c_uint64)), x.ctypes.data_as(C.POINTER(C.c_uint64)), s.ctypes.data_as(C.POINTER(C.c_uint64)))
        if op:
            oko += 1
            L.ls_hs_operator_max_number_off_diag(op); L.ls_hs_operator_is_hermitian(op); L.ls_hs_operator_is_real(op)
            c = L.ls_hs_clone_operator(op)
            if c: L.ls_hs_destroy_operator(c)
            L.ls_hs_destroy_operator(op)
    L.ls_hs_destroy_basis(b)
print(n, okb, oko, flush=True)

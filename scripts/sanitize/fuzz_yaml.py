"""Mutation fuzz of the C YAML loader (csrc/yaml.c + the basis / operator constructors of csrc/host.c) under ASAN + UBSAN:
   scripts/sanitize/run.sh builds the sanitizer library and calls this with a seed and a time budget.  Seeds are the reference's own
   inputs (/root/reference/data/*.yaml: this runs in the build container only); the input under test is written to
   <workdir>/current.yaml before every call, so a crash or a hang leaves its reproducer behind."""
import glob, os, random, sys, time
sys.path.insert(0, "/root/repo")
import importlib
_lib = importlib.import_module("distributed_matvec_amd._lib")
L = _lib.load()
seeds = [open(p, "rb").read() for p in sorted(glob.glob("/root/reference/data/*.yaml")) if os.path.getsize(p) < 20000]
seeds += [open(p, "rb").read() for p in sorted(glob.glob("/root/reference/data/old/*.yaml")) if os.path.getsize(p) < 20000][:10]
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
TOK = [b"&a", b"*a", b"[", b"]", b"{", b"}", b":", b"-", b"\n", b"  ", b"'", b'"', b"#", b"999999999999999999999", b"-1", b"1e400",
       b"\xcf\x83", b"\xe2\x82\x80", b"\xe2\x81\xba", b"\xc3\x97", b"sites", b"expression", b"permutation", b"sector", b"basis:", b"hamiltonian:",
       b"number_spins: 64", b"number_spins: 65", b"hamming_weight: 70", b"spin_inversion: 2", b"particle: spin-1/2", b"\t", b"\r\n", b"\x00", b"\xff"]
def mutate(b):
    b = bytearray(b)
    for _ in range(rnd.randint(1, 6)):
        k = rnd.randint(0, 8)
        if not b: b = bytearray(b"a")
        i = rnd.randrange(len(b))
        if k == 0: del b[i:i + rnd.randint(1, 20)]
        elif k == 1: b[i:i] = rnd.choice(TOK)
        elif k == 2: b[i] = rnd.randrange(256)
        elif k == 3: b = b[:i]
        elif k == 4:
            j = rnd.randrange(len(b)); b[i:i] = b[j:j + rnd.randint(1, 60)]
        elif k == 5:
            lines = bytes(b).split(b"\n"); j = rnd.randrange(len(lines)); lines.insert(rnd.randrange(len(lines) + 1), lines[j]); b = bytearray(b"\n".join(lines))
        elif k == 6:
            lines = bytes(b).split(b"\n"); j = rnd.randrange(len(lines)); del lines[j]; b = bytearray(b"\n".join(lines))
        elif k == 7:
            lines = bytes(b).split(b"\n"); j = rnd.randrange(len(lines)); lines[j] = b" " * rnd.randint(0, 6) + lines[j].lstrip(); b = bytearray(b"\n".join(lines))
        else:
            # digits -> other numbers
            for j in range(len(b)):
                if 48 <= b[j] <= 57 and rnd.random() < 0.05: b[j] = 48 + rnd.randrange(10)
    return bytes(b).replace(b"\x00", b"0")
t0 = time.time(); n = ok = 0
while time.time() - t0 < budget:
    s = mutate(rnd.choice(seeds))
    with open(os.path.join(os.environ.get("FUZZ_DIR", "/tmp"), f"current_{sys.argv[1] if len(sys.argv) > 1 else 1}.yaml"), "wb") as f: f.write(s)
    conf = L.ls_amd_load_yaml_config_from_string(s)
    n += 1
    if conf:
        ok += 1
        L.ls_hs_destroy_yaml_config(conf)
print(f"{n} inputs, {ok} accepted", flush=True)

#!/usr/bin/env python3
"""Per-kernel fingerprints of the gfx950 machine code of the kernel translation units (csrc/k_*.hip; kernels.hip until round 5).

A PMC measurement (profiles/pmc_traffic.json) belongs to the machine code of ONE kernel family, not to the whole
source file: editing k_tile_pull must not orphan the traffic measured for k_chain_t, and editing k_chain_t must.  So the
key is a hash of the kernel's own ISA: every k_*.hip is compiled device-only to assembly with the flags of the product
build, the body of every kernel (between its label and its .Lfunc_end) is cut out, function-local label numbers and
comments are normalised away (they depend on the position of the function in the file), and the bodies of all
instantiations of a family (k_chain_t, k_tile_pull, ...) are hashed together in sorted order.

usage: kernel_isa_sha.py [--source DIR-with-the-k_*.hip-files] [--out FILE]   (default: the tree, print to stdout)
Written by __graft_entry__.build() to distributed-matvec_amd/kernel_isa.json; read by bench.py."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the device-side flags of csrc/Makefile (HIPFLAGS)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-function",
         "-Wno-pass-failed", "-Wno-unused-command-line-argument"]


KERNEL_SOURCES = ("k_runtime.hip", "k_rows.hip", "k_packets.hip", "k_pull.hip", "k_plan.hip")
SHARED_SOURCES = ("lsk_dev.hpp", "lsk.h")


def kernel_sources(src_dir):
    """the kernel translation units of the tree (a pre-split tree: kernels.hip alone)"""
    have = [f for f in KERNEL_SOURCES if os.path.exists(os.path.join(src_dir, f))]
    return have or ["kernels.hip"]


def device_asm(src_dir):
    """the device assembly of every kernel translation unit, concatenated (compiled side by side)"""
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory() as tmp:
        def one(f):
            out = os.path.join(tmp, f + ".s")
            subprocess.check_call([HIPCC, *FLAGS, "--cuda-device-only", "-S", f"-I{src_dir}", os.path.join(src_dir, f), "-o", out],
                                  stderr=subprocess.DEVNULL)
            with open(out) as fh:
                return fh.read()

        with ThreadPoolExecutor(max_workers=5) as ex:
            return "\n".join(ex.map(one, kernel_sources(src_dir)))


def source_sha(src_dir):
    """= bench.source_sha(): the source the fingerprints were computed from"""
    hs = hashlib.sha256()
    for f in (*kernel_sources(src_dir), *SHARED_SOURCES):
        path = os.path.join(src_dir, f)
        if os.path.exists(path):
            with open(path, "rb") as fh:
                hs.update(fh.read())
    return hs.hexdigest()[:16]


_LOCAL_LABEL = re.compile(r"\.L(BB|tmp|func_begin|func_end|JTI)(\d+)(_\d+)?")


def kernel_bodies(asm):
    """{mangled kernel name: normalised instruction text}"""
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, flags=re.M))
    bodies = {}
    lines = asm.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(\S+):\s*(;.*)?$", lines[i])
        if m and m.group(1) in kernels and m.group(1) not in bodies:
            name, body = m.group(1), []
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
                ln = lines[i].split(";", 1)[0].rstrip()  # comments carry source line numbers
                if ln.strip() and not ln.lstrip().startswith((".loc", ".file", ".cfi", ".p2align")):
                    body.append(_LOCAL_LABEL.sub(lambda mm: ".L" + mm.group(1) + (mm.group(3) or ""), ln))
                i += 1
            bodies[name] = "\n".join(body)
        i += 1
    return bodies


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        if out.returncode == 0 and len(out.stdout.splitlines()) == len(names):
            return dict(zip(names, out.stdout.splitlines()))
    except OSError:
        pass
    return {n: n for n in names}


def family_of(mangled):
    """_Z9k_chain_tImjLb0E... -> k_chain_t (Itanium: length-prefixed source name); plain C names stay"""
    m = re.match(r"_Z(\d+)", mangled)
    return mangled[m.end():m.end() + int(m.group(1))] if m else mangled


def fingerprints(src_dir):
    bodies = kernel_bodies(device_asm(src_dir))
    dem = demangle(sorted(bodies))
    fam = {}
    for mangled in sorted(bodies):
        if family_of(mangled).startswith("k_"):  # this library's kernels; the hipcub/rocprim scans are not fingerprinted
            fam.setdefault(family_of(mangled), []).append(mangled)
    out = {"_comment": "sha256[:16] of the normalised gfx950 ISA of every kernel family of csrc/k_*.hip "
                       "(scripts/kernel_isa_sha.py); bench.py attaches a PMC entry only to the machine code it measured",
           "source_sha": source_sha(src_dir), "families": {}, "kernels": {}}
    for f, members in sorted(fam.items()):
        h = hashlib.sha256()
        for mangled in members:
            h.update(mangled.encode() + b"\n" + bodies[mangled].encode() + b"\n")
            out["kernels"][re.sub(r"^void ", "", dem[mangled].split("(")[0])] = hashlib.sha256(bodies[mangled].encode()).hexdigest()[:16]
        out["families"][f] = {"isa_sha": h.hexdigest()[:16], "instantiations": len(members)}
    return out


def main():
    src = os.path.join(ROOT, "distributed-matvec_amd", "csrc")
    dst = None
    a = sys.argv[1:]
    while a:
        if a[0] == "--source":
            src = a[1]
        elif a[0] == "--out":
            dst = a[1]
        else:
            sys.exit(__doc__)
        a = a[2:]
    txt = json.dumps(fingerprints(src), indent=1, sort_keys=True) + "\n"
    if dst:
        with open(dst, "w") as f:
            f.write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()

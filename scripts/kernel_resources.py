#!/usr/bin/env python3
"""Register / LDS / scratch use of every device function of the kernel translation units csrc/k_*.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [substring ...]   -- prints name, SGPRs, VGPRs, occupancy, scratch bytes per lane, LDS bytes"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


KERNEL_SOURCES = ("k_rows.hip", "k_packets.hip", "k_pull.hip", "k_plan.hip")


def resources(extra_flags=(), source=None):
    """{mangled device function: registers, occupancy, scratch, LDS} over every kernel translation unit (or the one named)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    csrc = os.path.join(ROOT, "distributed-matvec_amd", "csrc")
    stats = {}

    def one(name):
        src = os.path.join(csrc, name)
        return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", src,
                               "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", *extra_flags], capture_output=True, text=True,
                              cwd=csrc).stderr

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(one, [source] if source else KERNEL_SOURCES))
    for err in outs:
        for b in re.split(r"Function Name: ", err)[1:]:
            name = b.split()[0]
            g = lambda pat: (lambda m: int(m.group(1)) if m else None)(re.search(pat, b))  # noqa: E731
            stats[name] = {"sgpr": g(r"TotalSGPRs: (\d+)"), "vgpr": g(r"VGPRs: (\d+)"), "occ": g(r"Occupancy \[waves/SIMD\]: (\d+)"),
                           "scratch": g(r"ScratchSize \[bytes/lane\]: (\d+)"), "lds": g(r"LDS Size \[bytes/block\]: (\d+)")}
    return stats


if __name__ == "__main__":
    st = resources()
    demangle = subprocess.run(["c++filt"], input="\n".join(st), capture_output=True, text=True).stdout.split("\n")
    print(len(st), "device functions")
    for (name, v), dn in zip(st.items(), demangle):
        short = dn.split("(")[0].replace("void ", "")
        if sys.argv[1:] and not any(a in short for a in sys.argv[1:]):
            continue
        print(f"{short:70s} sgpr {v['sgpr']:3d} vgpr {v['vgpr']:3d} occ {v['occ']} scratch {v['scratch']:3d} lds {v['lds']}")

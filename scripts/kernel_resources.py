#!/usr/bin/env python3
"""Register / LDS / scratch use of every device function of csrc/kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [substring ...]   -- prints name, SGPRs, VGPRs, occupancy, scratch bytes per lane, LDS bytes"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(extra_flags=(), source="kernels.hip"):
    src = os.path.join(ROOT, "distributed-matvec_amd", "csrc", source)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", src,
                          "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", *extra_flags], capture_output=True, text=True,
                         cwd=os.path.dirname(src))
    stats = {}
    for b in re.split(r"Function Name: ", out.stderr)[1:]:
        name = b.split()[0]
        g = lambda pat: (lambda m: int(m.group(1)) if m else None)(re.search(pat, b))
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

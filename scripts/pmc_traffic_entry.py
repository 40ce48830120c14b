#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of one bench command into an entry of profiles/pmc_traffic.json.

usage: pmc_traffic_entry.py <dir with pmc_fetch/ pmc_write/ [pmc_valu/] rocpd dbs> <model> <dtype> <plan kernel name>
Prints {key: entry} as JSON.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (gfx950: the counter
tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md; calibrated on k_diag in round 1)."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEVICE_KERNEL = {"push+staged": "k_push_t", "staged": "k_chain_t", "indexed": "k_pull_t", "tile-pull": "k_tile_pull<", "direct-push": "k_direct",
                 "direct-pull": "k_direct", "tile": "k_tile<"}
SEEN = []  # kernel names the counters were read from


def mean_counter(root, sub, counter, needle):
    """mean of `counter` over the dispatches of THE instantiation of `needle` that ran most often (the timed kernel: since round
    6 the bench line also runs the generic kernels once each for its parity object, and they share a family name)"""
    by_name = {}
    for db_path in glob.glob(os.path.join(root, sub, "**", "*_results.db"), recursive=True):
        cur = sqlite3.connect(db_path).cursor()
        cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
        name_col = "kernel_name" if "kernel_name" in cols else cols[0]
        for kn, v in cur.execute(f"select {name_col}, value from counters_collection where counter_name = ?", (counter,)):
            if needle in kn:
                by_name.setdefault(kn, []).append(v)
    if not by_name:
        return None, 0
    kn = max(by_name, key=lambda k: len(by_name[k]))
    if kn not in SEEN:
        SEEN.append(kn)
    vals = by_name[kn]
    return sum(vals) / len(vals), len(vals)


def main():
    root, model, dtype, kname = sys.argv[1:5]
    from bench import kernel_instance_sha, kernel_isa_sha, source_sha

    needle = next(v for k, v in DEVICE_KERNEL.items() if k in kname)
    fetch, nf = mean_counter(root, "pmc_fetch", "FETCH_SIZE", needle)
    write, nw = mean_counter(root, "pmc_write", "WRITE_SIZE", needle)
    valu, nv = mean_counter(root, "pmc_valu", "SQ_INSTS_VALU", needle)
    # FETCH_SIZE tallies requests at 64 bytes: a wide streaming read issues 128-byte requests (x 2, calibrated on k_diag in
    # round 1); the staged kernel of the projected bases reads one 16-byte hash entry per probe = one 64-byte request (x 1)
    corr = 1.0 if "tile" in kname else 2.0
    entry = {
        "fetch_size_kib_raw": fetch, "fetch_correction": corr, "write_size_kib": write,
        "traffic_bytes": (corr * fetch + write) * 1024.0 if fetch is not None and write is not None else None,
        "dispatches": [nf, nw], "device_kernel": needle, "source_sha": source_sha(),
        "isa_sha": kernel_isa_sha(needle.rstrip("<")),
        "source": os.environ.get("PMC_SOURCE", ""),
    }
    if valu is not None:
        entry["valu_insts"] = valu
    if len(SEEN) == 1:  # the instantiation that ran: the entry is tied to ITS machine code (bench.py prefers this)
        inst = SEEN[0].split("(")[0].replace("void ", "").strip()
        sha = kernel_instance_sha(inst)
        if sha:
            entry["instance"], entry["instance_isa_sha"] = inst, sha
    print(json.dumps({f"{model}/{dtype}/{kname}": entry}, indent=1))


if __name__ == "__main__":
    main()

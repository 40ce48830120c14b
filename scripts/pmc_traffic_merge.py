#!/usr/bin/env python3
"""merge gpurun_out/<tag>/pmc_traffic_entry.json files into profiles/pmc_traffic.json and copy the summaries"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
with open(dst) as f:
    table = json.load(f)
for tag in sys.argv[1:]:
    d = os.path.join(ROOT, "gpurun_out", tag)
    with open(os.path.join(d, "pmc_traffic_entry.json")) as f:
        table.update(json.load(f))
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.txt"))
    if os.path.exists(os.path.join(d, "bench_line.json")):
        shutil.copy(os.path.join(d, "bench_line.json"), os.path.join(ROOT, "profiles", f"{tag}_bench_line.json"))
with open(dst, "w") as f:
    json.dump(table, f, indent=1)
print("merged", sys.argv[1:])

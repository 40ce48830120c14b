#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel stats and mean PMC counter per dispatch.
usage: rocpd_summary.py <dir with *_results.db files (searched recursively)>"""
import glob
import os
import sqlite3
import sys


def main(root):
    for db_path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        db = sqlite3.connect(db_path)
        cur = db.cursor()
        print(f"== {os.path.relpath(db_path, root)}")
        try:
            rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            print(f"{'kernel':80s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'pct':>7s}")
            for name, calls, total, avg, pct in rows[:12]:
                print(f"{name[:80]:80s} {calls:6d} {total/1e3:12.4f} {avg/1e3:12.4f} {pct:7.2f}")
        except sqlite3.Error as e:
            print("  (no kernel stats:", e, ")")
        try:
            cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
            if cols:
                name_col = "kernel_name" if "kernel_name" in cols else cols[0]
                q = f"select {name_col}, counter_name, avg(value), count(*) from counters_collection group by {name_col}, counter_name"
                rows = cur.execute(q).fetchall()
                if rows:
                    print(f"{'kernel':60s} {'counter':24s} {'mean/dispatch':>16s} {'n':>5s}")
                for kn, cn, v, n in rows:
                    if any(t in kn for t in ("k_direct", "k_diag", "k_tile", "k_scatter", "k_window", "k_lin", "k_pull", "k_chain", "k_pairs")):
                        print(f"{kn[:60]:60s} {cn:24s} {v:16.6g} {n:5d}")
        except sqlite3.Error as e:
            print("  (no counters:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")

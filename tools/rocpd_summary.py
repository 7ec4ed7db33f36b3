#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) into the
per-kernel table rocprofv3 prints as kernel_stats.csv: calls, total / average / min / max duration, share.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "")
        short = re.sub(r"^void ", "", short)
        short = re.sub(r"\(.*$", "", short)
        d = (e - s) / 1e3          # us
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# source: {path}   kernels: {len(rows)} dispatches, {tot / 1e3:.2f} ms total GPU kernel time")
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:100]:100s} {a[0]:7d} {a[1] / 1e3:10.3f} {a[1] / a[0]:10.1f} {a[2]:10.1f} {a[3]:10.1f} {100 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python
"""Durations of the dispatches of one kernel family in issue order, averaged over runs of `n` consecutive launches (a
sweep script launches each configuration n times): GPU-side times where an event-timed loop would be host-bound.
    python tools/rocpd_sequence.py results.db conv_gemm 51"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub, n = sys.argv[2], int(sys.argv[3])
rows = [(r[1], r[2] - r[1], re.sub(r"\(.*$", "", r[0])) for r in cur.execute("select name, start, end from kernels") if sub in r[0]]
rows.sort()
for i in range(0, len(rows) - n + 1, n):
    blk = rows[i:i + n]
    d = sorted(b[1] for b in blk)
    gaps = [blk[j + 1][0] - (blk[j][0] + blk[j][1]) for j in range(len(blk) - 1)]
    print(f"run {i // n:3d}: median {d[len(d) // 2] / 1e3:7.1f} us  min {d[0] / 1e3:7.1f}  start-to-start {((blk[-1][0] - blk[0][0]) / (n - 1)) / 1e3:7.1f} us  "
          f"{blk[0][2][-60:]}")

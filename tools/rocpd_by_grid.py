#!/usr/bin/env python
"""Per-(kernel, grid size) duration summary of a rocprofv3 rocpd database: tells the shapes of one kernel apart.
    python tools/rocpd_by_grid.py results.db [kernel-substring]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gcol = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid")] or [c for c in cols if "grid" in c.lower()]
wcol = [c for c in cols if "workgroup" in c.lower() or "block" in c.lower()]
# (r6: all three grid dimensions -- the GroupNorm kernels launch 2-D grids)
full3 = all(c in cols for c in ("grid_x", "grid_y", "grid_z", "workgroup_x", "workgroup_y", "workgroup_z"))
if full3:
    q = f"select {name_col}, start, end, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * workgroup_z from kernels"
else:
    q = f"select {name_col}, start, end, {gcol[0]}" + (f", {wcol[0]}" if wcol else "") + " from kernels"
agg = {}
for row in cur.execute(q):
    name = re.sub(r"\(.*$", "", row[0].replace("(anonymous namespace)::", "").replace("void ", ""))
    if sub not in name:
        continue
    g = row[3] // (row[4] if (full3 or wcol) and row[4] else 1)
    a = agg.setdefault((name, g), [0, 0.0])
    a[0] += 1; a[1] += (row[2] - row[1]) / 1e3
print("columns:", cols)
for (name, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name[:60]:60s} wgs={g:6d} calls={a[0]:5d} total_ms={a[1] / 1e3:8.3f} avg_us={a[1] / a[0]:8.1f}")

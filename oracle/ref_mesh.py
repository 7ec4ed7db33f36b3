"""TEST INFRASTRUCTURE: numpy restatement of table-driven marching cubes as the reference uses it --
`mcubes.marching_cubes(sdf_i, level)` then `verts / n_cell - .5` (model/diff_utils/util_3d.py:215-218).

PARITY UNPINNED: PyMCubes (pmneila/PyMCubes, the `mcubes` import of util_3d.py:9, unpinned in requirements.txt) is not in
the image and cannot be fetched, so neither its tables nor its output order can be compared.  What is restated is the
published algorithm: a vertex on every grid edge whose endpoints straddle the level ((v < level) differs), linearly
interpolated in float64 (PyMCubes converts the volume to double), one shared vertex per edge, triangles from a 256-case
table.  The table is derived here from the cube geometry with the same rules as the product's generator
(commonscenes_amd/mc_tables.py) but by its own code path; tests compare both and check invariants that hold for ANY
correct marching cubes: watertightness, vertices on the isosurface, outward orientation, Euler characteristic."""
from __future__ import annotations

import itertools
from typing import List, Tuple

import numpy as np

CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def _faces():
    """the six faces as corner 4-cycles, found from the coordinates (not copied from the product)."""
    out = []
    for axis, val in itertools.product(range(3), (0, 1)):
        cs = [c for c in range(8) if CORNER[c][axis] == val]
        # order the four corners into a cycle: neighbours differ in exactly one coordinate
        cyc = [cs[0]]
        while len(cyc) < 4:
            for c in cs:
                if c not in cyc and sum(a != b for a, b in zip(CORNER[c], CORNER[cyc[-1]])) == 1:
                    cyc.append(c)
                    break
        out.append(cyc)
    return out


def case_triangles(case: int) -> List[Tuple[int, int, int]]:
    eid = {frozenset(e): i for i, e in enumerate(EDGE)}
    inside = [(case >> c) & 1 for c in range(8)]
    adj = {i: [] for i in range(12)}
    for cyc in _faces():
        edges = [eid[frozenset((cyc[i], cyc[(i + 1) % 4]))] for i in range(4)]
        cross = [i for i in range(4) if inside[cyc[i]] != inside[cyc[(i + 1) % 4]]]
        pairs = []
        if len(cross) == 2:
            pairs = [(edges[cross[0]], edges[cross[1]])]
        elif len(cross) == 4:        # ambiguous face: isolate each inside corner
            pairs = [(edges[(i - 1) % 4], edges[i]) for i in range(4) if inside[cyc[i]]]
        for a, b in pairs:
            adj[a].append(b)
            adj[b].append(a)
    tris, used = [], set()
    for start in range(12):
        if start in used or not adj[start]:
            continue
        loop, prev, cur = [start], -1, start
        used.add(start)
        while True:
            cand = [x for x in adj[cur] if x != prev]
            nxt = cand[0] if cand else adj[cur][0]
            if nxt == start:
                break
            loop.append(nxt)
            used.add(nxt)
            prev, cur = cur, nxt
        mid = np.array([(np.array(CORNER[EDGE[e][0]]) + np.array(CORNER[EDGE[e][1]])) / 2.0 for e in loop])
        nrm = sum(np.cross(mid[i], mid[(i + 1) % len(loop)]) for i in range(len(loop)))
        out_dir = sum((np.array(CORNER[b]) - np.array(CORNER[a])) * (1 if inside[a] else -1) for a, b in (EDGE[e] for e in loop))
        if float(np.dot(nrm, out_dir)) < 0:
            loop = loop[::-1]
        k = loop.index(min(loop))
        loop = loop[k:] + loop[:k]
        # fan apex: the first rotation without a diagonal inside a cube face (two crossed edges of one face joined by a
        # diagonal could coincide with the neighbouring cube's diagonal: a non-manifold mesh edge)
        on_face = lambda a, b: any(set(EDGE[a]) <= set(f) and set(EDGE[b]) <= set(f) for f in _faces())
        for r in range(len(loop)):
            cand = loop[r:] + loop[:r]
            if not any(on_face(cand[0], cand[i]) for i in range(2, len(cand) - 1)):
                loop = cand
                break
        tris.append(loop)
    res = []
    for loop in sorted(tris, key=min):
        res += [(loop[0], loop[i], loop[i + 1]) for i in range(1, len(loop) - 1)]
    return res


_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        _TABLE = [case_triangles(c) for c in range(256)]
    return _TABLE


def marching_cubes(vol: np.ndarray, level: float, tab=None):
    """vol: (n,n,n) float32 -> (verts float64 [V,3] in index coordinates, faces int64 [F,3]); vertex order: voxel raster,
    +x, +y, +z edge of each voxel; face order: cube raster, table order.  tab: 256 lists of (edge, edge, edge) triangles;
    None = the table derived above (the product's 'watertight' table).  The classic Lorensen-Cline table is published
    DATA -- it cannot be re-derived, only validated (mc_tables.validate_table) -- so tests hand the product's copy in."""
    n = vol.shape[0]
    v64 = vol.astype(np.float64)
    ins = vol < np.float32(level)
    vid = -np.ones((n, n, n, 3), dtype=np.int64)
    verts = []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                for a, (di, dj, dk) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    i2, j2, k2 = i + di, j + dj, k + dk
                    if i2 < n and j2 < n and k2 < n and ins[i, j, k] != ins[i2, j2, k2]:
                        f0, f1 = v64[i, j, k], v64[i2, j2, k2]
                        t = (np.float64(np.float32(level)) - f0) / (f1 - f0)
                        p = [float(i), float(j), float(k)]
                        p[a] += t
                        vid[i, j, k, a] = len(verts)
                        verts.append(p)
    tab = table() if tab is None else tab
    faces = []
    for i in range(n - 1):
        for j in range(n - 1):
            for k in range(n - 1):
                case = 0
                for c, (dx, dy, dz) in enumerate(CORNER):
                    case |= int(ins[i + dx, j + dy, k + dz]) << c
                for tri in tab[case]:
                    f = []
                    for e in tri:
                        a, b = EDGE[e]
                        ca, cb = CORNER[a], CORNER[b]
                        axis = [x != y for x, y in zip(ca, cb)].index(True)
                        o = [min(x, y) for x, y in zip(ca, cb)]
                        f.append(vid[i + o[0], j + o[1], k + o[2], axis])
                    faces.append(f)
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def mesh_invariants(verts: np.ndarray, faces: np.ndarray):
    """(open_edges, nonmanifold_edges, inconsistently_oriented_edges, euler_characteristic)."""
    from collections import Counter
    und, dire = Counter(), Counter()
    for f in faces:
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            und[(min(a, b), max(a, b))] += 1
            dire[(a, b)] += 1
    open_e = sum(1 for c in und.values() if c == 1)
    nonman = sum(1 for c in und.values() if c > 2)
    bad_orient = sum(1 for (a, b), c in dire.items() if c > 1)
    chi = len(np.unique(faces)) - len(und) + len(faces)
    return open_e, nonman, bad_orient, chi

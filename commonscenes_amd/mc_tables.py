"""Marching-cubes case tables: the classic one (default) and a watertight one derived from the cube's geometry.

The reference extracts meshes with PyMCubes (`mcubes.marching_cubes(sdf_i, level)`, model/diff_utils/util_3d.py:217),
a table-driven marching cubes over the 256 corner-sign cases.  PyMCubes is not in this image and cannot be fetched,
so its tables cannot be compared byte for byte.  Two tables are offered (`TABLES`, `cs_mc_count(..., table)`):

TABLE_CLASSIC (0, the default since r4) -- the classic Lorensen-Cline case table in its universally replicated 256-row
  form (P. Bourke, "Polygonising a scalar field", triTable; 820 triangles, at most 5 per cube) in the corner / edge
  numbering below, which is also the numbering PyMCubes documents for its `tri_table`: the triangle SET a user of
  `mcubes.marching_cubes` gets.  Its winding is kept as published: triangle normals point towards DECREASING values
  (into a negative-inside SDF's object).  The rows are data, not code; `validate_table` checks every one of them against
  what any marching-cubes table must satisfy -- exactly the crossed edges used, a manifold consistently oriented patch
  whose boundary runs along cube faces through every crossed edge once in and once out.  Measured (tests/test_mesh.py):
  on all 256 cases its patches have the SAME boundary loops as the derived table below -- the replicated table, too,
  cuts off the inside corners of an ambiguous face (only the original 15-case table with complement symmetry leaves
  holes) -- so both tables mesh the same watertight surface; they differ in how polygons are fanned (~a third of the
  cases) and in winding.  Vertex / face ORDER of PyMCubes stays unpinned.

TABLE_WATERTIGHT (1, r2-r3's only table) -- built from first principles with one rule per step:

  corners   c = x + 2y + 4z ... in the classic numbering  0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1)
            6:(1,1,1) 7:(0,1,1);  edges 0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7
  inside    bit c of the case index is set when value[c] < level (PyMCubes' test)
  faces     on every cube face the crossed edges are joined pairwise; a face with four crossings (diagonal corners
            alike) is AMBIGUOUS and is resolved by cutting off each INSIDE corner -- a rule that depends only on the
            face's own four corner signs, so the two cubes sharing a face always agree and the surface is watertight
            (the classic 15-case table with complement symmetry is not)
  loops     the face segments close into loops; each loop is oriented so its normal points from the inside corners to
            the outside ones (towards increasing value: outward for an SDF) and triangulated as a fan whose apex is
            chosen so that no diagonal lies in a cube face (such a diagonal could coincide with the neighbouring
            cube's and make the shared mesh edge non-manifold); an apex with that property exists for every loop
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
# the six faces as corner cycles
FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5)]
_EDGE_ID = {frozenset(e): i for i, e in enumerate(EDGES)}
# owner of each cube edge on the grid: (di, dj, dk, axis): the voxel whose +axis edge it is
EDGE_OWNER = [(int(CORNERS[a][0] if CORNERS[a][0] == CORNERS[b][0] else 0), int(CORNERS[a][1] if CORNERS[a][1] == CORNERS[b][1] else 0),
               int(CORNERS[a][2] if CORNERS[a][2] == CORNERS[b][2] else 0), int(np.argmax(np.abs(CORNERS[a] - CORNERS[b]))))
              for a, b in EDGES]


_FACES_OF = {e: set() for e in range(12)}
for _fi, _cyc in enumerate(FACES):
    for _i in range(4):
        _FACES_OF[_EDGE_ID[frozenset((_cyc[_i], _cyc[(_i + 1) % 4]))]].add(_fi)


def _case_loops(case: int) -> List[List[int]]:
    inside = [(case >> c) & 1 for c in range(8)]
    nbr = {e: [] for e in range(12)}
    for cyc in FACES:
        fe = [_EDGE_ID[frozenset((cyc[i], cyc[(i + 1) % 4]))] for i in range(4)]      # edge i joins corner i, i+1
        crossed = [i for i in range(4) if inside[cyc[i]] != inside[cyc[(i + 1) % 4]]]
        if len(crossed) == 2:
            a, b = fe[crossed[0]], fe[crossed[1]]
            nbr[a].append(b)
            nbr[b].append(a)
        elif len(crossed) == 4:
            for i in range(4):                      # cut off every inside corner: join its two face edges
                if inside[cyc[i]]:
                    a, b = fe[(i - 1) % 4], fe[i]
                    nbr[a].append(b)
                    nbr[b].append(a)
    loops, seen = [], set()
    for e0 in range(12):
        if e0 in seen or not nbr[e0]:
            continue
        assert len(nbr[e0]) == 2
        loop, prev, cur = [e0], None, e0
        seen.add(e0)
        while True:
            nxt = [n for n in nbr[cur] if n != prev] or nbr[cur]
            n = nxt[0]
            if n == e0:
                break
            if n in seen:            # two-edge degenerate cannot happen on a cube
                raise AssertionError("broken loop")
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        # orientation: Newell normal of the edge-midpoint polygon vs (outside endpoints - inside endpoints)
        mid = np.array([(CORNERS[EDGES[e][0]] + CORNERS[EDGES[e][1]]) / 2.0 for e in loop])
        nrm = np.zeros(3)
        for i in range(len(loop)):
            p, q = mid[i], mid[(i + 1) % len(loop)]
            nrm += np.cross(p, q)
        d = np.zeros(3)
        for e in loop:
            a, b = EDGES[e]
            d += (CORNERS[b] - CORNERS[a]) * (1 if inside[a] else -1)
        assert abs(float(nrm @ d)) > 1e-9
        if nrm @ d < 0:
            loop = loop[::-1]
        # canonical start: the smallest edge id first (keeps the table independent of traversal order) ...
        k = loop.index(min(loop))
        loop = loop[k:] + loop[:k]
        # ... then the first rotation whose fan has no diagonal inside a cube face
        for r in range(len(loop)):
            cand = loop[r:] + loop[:r]
            if all(not (_FACES_OF[cand[0]] & _FACES_OF[cand[i]]) for i in range(2, len(cand) - 1)):
                loop = cand
                break
        else:
            raise AssertionError("no face-diagonal-free fan")
        loops.append(loop)
    return sorted(loops, key=lambda l: min(l))


def build_tables() -> Tuple[np.ndarray, np.ndarray]:
    """(tri_table int8 [256][3 * MAX_TRIS] edge ids, -1 padded; n_tris uint8 [256])."""
    rows = []
    for case in range(256):
        tris: List[int] = []
        for loop in _case_loops(case):
            for i in range(1, len(loop) - 1):
                tris += [loop[0], loop[i], loop[i + 1]]
        rows.append(tris)
    mx = max(len(r) for r in rows)
    tab = -np.ones((256, mx), dtype=np.int8)
    for i, r in enumerate(rows):
        tab[i, :len(r)] = r
    return tab, np.array([len(r) // 3 for r in rows], dtype=np.uint8)


CLASSIC_ROWS = [
    [],
    [0,8,3],
    [0,1,9],
    [1,8,3,9,8,1],
    [1,2,10],
    [0,8,3,1,2,10],
    [9,2,10,0,2,9],
    [2,8,3,2,10,8,10,9,8],
    [3,11,2],
    [0,11,2,8,11,0],
    [1,9,0,2,3,11],
    [1,11,2,1,9,11,9,8,11],
    [3,10,1,11,10,3],
    [0,10,1,0,8,10,8,11,10],
    [3,9,0,3,11,9,11,10,9],
    [9,8,10,10,8,11],
    [4,7,8],
    [4,3,0,7,3,4],
    [0,1,9,8,4,7],
    [4,1,9,4,7,1,7,3,1],
    [1,2,10,8,4,7],
    [3,4,7,3,0,4,1,2,10],
    [9,2,10,9,0,2,8,4,7],
    [2,10,9,2,9,7,2,7,3,7,9,4],
    [8,4,7,3,11,2],
    [11,4,7,11,2,4,2,0,4],
    [9,0,1,8,4,7,2,3,11],
    [4,7,11,9,4,11,9,11,2,9,2,1],
    [3,10,1,3,11,10,7,8,4],
    [1,11,10,1,4,11,1,0,4,7,11,4],
    [4,7,8,9,0,11,9,11,10,11,0,3],
    [4,7,11,4,11,9,9,11,10],
    [9,5,4],
    [9,5,4,0,8,3],
    [0,5,4,1,5,0],
    [8,5,4,8,3,5,3,1,5],
    [1,2,10,9,5,4],
    [3,0,8,1,2,10,4,9,5],
    [5,2,10,5,4,2,4,0,2],
    [2,10,5,3,2,5,3,5,4,3,4,8],
    [9,5,4,2,3,11],
    [0,11,2,0,8,11,4,9,5],
    [0,5,4,0,1,5,2,3,11],
    [2,1,5,2,5,8,2,8,11,4,8,5],
    [10,3,11,10,1,3,9,5,4],
    [4,9,5,0,8,1,8,10,1,8,11,10],
    [5,4,0,5,0,11,5,11,10,11,0,3],
    [5,4,8,5,8,10,10,8,11],
    [9,7,8,5,7,9],
    [9,3,0,9,5,3,5,7,3],
    [0,7,8,0,1,7,1,5,7],
    [1,5,3,3,5,7],
    [9,7,8,9,5,7,10,1,2],
    [10,1,2,9,5,0,5,3,0,5,7,3],
    [8,0,2,8,2,5,8,5,7,10,5,2],
    [2,10,5,2,5,3,3,5,7],
    [7,9,5,7,8,9,3,11,2],
    [9,5,7,9,7,2,9,2,0,2,7,11],
    [2,3,11,0,1,8,1,7,8,1,5,7],
    [11,2,1,11,1,7,7,1,5],
    [9,5,8,8,5,7,10,1,3,10,3,11],
    [5,7,0,5,0,9,7,11,0,1,0,10,11,10,0],
    [11,10,0,11,0,3,10,5,0,8,0,7,5,7,0],
    [11,10,5,7,11,5],
    [10,6,5],
    [0,8,3,5,10,6],
    [9,0,1,5,10,6],
    [1,8,3,1,9,8,5,10,6],
    [1,6,5,2,6,1],
    [1,6,5,1,2,6,3,0,8],
    [9,6,5,9,0,6,0,2,6],
    [5,9,8,5,8,2,5,2,6,3,2,8],
    [2,3,11,10,6,5],
    [11,0,8,11,2,0,10,6,5],
    [0,1,9,2,3,11,5,10,6],
    [5,10,6,1,9,2,9,11,2,9,8,11],
    [6,3,11,6,5,3,5,1,3],
    [0,8,11,0,11,5,0,5,1,5,11,6],
    [3,11,6,0,3,6,0,6,5,0,5,9],
    [6,5,9,6,9,11,11,9,8],
    [5,10,6,4,7,8],
    [4,3,0,4,7,3,6,5,10],
    [1,9,0,5,10,6,8,4,7],
    [10,6,5,1,9,7,1,7,3,7,9,4],
    [6,1,2,6,5,1,4,7,8],
    [1,2,5,5,2,6,3,0,4,3,4,7],
    [8,4,7,9,0,5,0,6,5,0,2,6],
    [7,3,9,7,9,4,3,2,9,5,9,6,2,6,9],
    [3,11,2,7,8,4,10,6,5],
    [5,10,6,4,7,2,4,2,0,2,7,11],
    [0,1,9,4,7,8,2,3,11,5,10,6],
    [9,2,1,9,11,2,9,4,11,7,11,4,5,10,6],
    [8,4,7,3,11,5,3,5,1,5,11,6],
    [5,1,11,5,11,6,1,0,11,7,11,4,0,4,11],
    [0,5,9,0,6,5,0,3,6,11,6,3,8,4,7],
    [6,5,9,6,9,11,4,7,9,7,11,9],
    [10,4,9,6,4,10],
    [4,10,6,4,9,10,0,8,3],
    [10,0,1,10,6,0,6,4,0],
    [8,3,1,8,1,6,8,6,4,6,1,10],
    [1,4,9,1,2,4,2,6,4],
    [3,0,8,1,2,9,2,4,9,2,6,4],
    [0,2,4,4,2,6],
    [8,3,2,8,2,4,4,2,6],
    [10,4,9,10,6,4,11,2,3],
    [0,8,2,2,8,11,4,9,10,4,10,6],
    [3,11,2,0,1,6,0,6,4,6,1,10],
    [6,4,1,6,1,10,4,8,1,2,1,11,8,11,1],
    [9,6,4,9,3,6,9,1,3,11,6,3],
    [8,11,1,8,1,0,11,6,1,9,1,4,6,4,1],
    [3,11,6,3,6,0,0,6,4],
    [6,4,8,11,6,8],
    [7,10,6,7,8,10,8,9,10],
    [0,7,3,0,10,7,0,9,10,6,7,10],
    [10,6,7,1,10,7,1,7,8,1,8,0],
    [10,6,7,10,7,1,1,7,3],
    [1,2,6,1,6,8,1,8,9,8,6,7],
    [2,6,9,2,9,1,6,7,9,0,9,3,7,3,9],
    [7,8,0,7,0,6,6,0,2],
    [7,3,2,6,7,2],
    [2,3,11,10,6,8,10,8,9,8,6,7],
    [2,0,7,2,7,11,0,9,7,6,7,10,9,10,7],
    [1,8,0,1,7,8,1,10,7,6,7,10,2,3,11],
    [11,2,1,11,1,7,10,6,1,6,7,1],
    [8,9,6,8,6,7,9,1,6,11,6,3,1,3,6],
    [0,9,1,11,6,7],
    [7,8,0,7,0,6,3,11,0,11,6,0],
    [7,11,6],
    [7,6,11],
    [3,0,8,11,7,6],
    [0,1,9,11,7,6],
    [8,1,9,8,3,1,11,7,6],
    [10,1,2,6,11,7],
    [1,2,10,3,0,8,6,11,7],
    [2,9,0,2,10,9,6,11,7],
    [6,11,7,2,10,3,10,8,3,10,9,8],
    [7,2,3,6,2,7],
    [7,0,8,7,6,0,6,2,0],
    [2,7,6,2,3,7,0,1,9],
    [1,6,2,1,8,6,1,9,8,8,7,6],
    [10,7,6,10,1,7,1,3,7],
    [10,7,6,1,7,10,1,8,7,1,0,8],
    [0,3,7,0,7,10,0,10,9,6,10,7],
    [7,6,10,7,10,8,8,10,9],
    [6,8,4,11,8,6],
    [3,6,11,3,0,6,0,4,6],
    [8,6,11,8,4,6,9,0,1],
    [9,4,6,9,6,3,9,3,1,11,3,6],
    [6,8,4,6,11,8,2,10,1],
    [1,2,10,3,0,11,0,6,11,0,4,6],
    [4,11,8,4,6,11,0,2,9,2,10,9],
    [10,9,3,10,3,2,9,4,3,11,3,6,4,6,3],
    [8,2,3,8,4,2,4,6,2],
    [0,4,2,4,6,2],
    [1,9,0,2,3,4,2,4,6,4,3,8],
    [1,9,4,1,4,2,2,4,6],
    [8,1,3,8,6,1,8,4,6,6,10,1],
    [10,1,0,10,0,6,6,0,4],
    [4,6,3,4,3,8,6,10,3,0,3,9,10,9,3],
    [10,9,4,6,10,4],
    [4,9,5,7,6,11],
    [0,8,3,4,9,5,11,7,6],
    [5,0,1,5,4,0,7,6,11],
    [11,7,6,8,3,4,3,5,4,3,1,5],
    [9,5,4,10,1,2,7,6,11],
    [6,11,7,1,2,10,0,8,3,4,9,5],
    [7,6,11,5,4,10,4,2,10,4,0,2],
    [3,4,8,3,5,4,3,2,5,10,5,2,11,7,6],
    [7,2,3,7,6,2,5,4,9],
    [9,5,4,0,8,6,0,6,2,6,8,7],
    [3,6,2,3,7,6,1,5,0,5,4,0],
    [6,2,8,6,8,7,2,1,8,4,8,5,1,5,8],
    [9,5,4,10,1,6,1,7,6,1,3,7],
    [1,6,10,1,7,6,1,0,7,8,7,0,9,5,4],
    [4,0,10,4,10,5,0,3,10,6,10,7,3,7,10],
    [7,6,10,7,10,8,5,4,10,4,8,10],
    [6,9,5,6,11,9,11,8,9],
    [3,6,11,0,6,3,0,5,6,0,9,5],
    [0,11,8,0,5,11,0,1,5,5,6,11],
    [6,11,3,6,3,5,5,3,1],
    [1,2,10,9,5,11,9,11,8,11,5,6],
    [0,11,3,0,6,11,0,9,6,5,6,9,1,2,10],
    [11,8,5,11,5,6,8,0,5,10,5,2,0,2,5],
    [6,11,3,6,3,5,2,10,3,10,5,3],
    [5,8,9,5,2,8,5,6,2,3,8,2],
    [9,5,6,9,6,0,0,6,2],
    [1,5,8,1,8,0,5,6,8,3,8,2,6,2,8],
    [1,5,6,2,1,6],
    [1,3,6,1,6,10,3,8,6,5,6,9,8,9,6],
    [10,1,0,10,0,6,9,5,0,5,6,0],
    [0,3,8,5,6,10],
    [10,5,6],
    [11,5,10,7,5,11],
    [11,5,10,11,7,5,8,3,0],
    [5,11,7,5,10,11,1,9,0],
    [10,7,5,10,11,7,9,8,1,8,3,1],
    [11,1,2,11,7,1,7,5,1],
    [0,8,3,1,2,7,1,7,5,7,2,11],
    [9,7,5,9,2,7,9,0,2,2,11,7],
    [7,5,2,7,2,11,5,9,2,3,2,8,9,8,2],
    [2,5,10,2,3,5,3,7,5],
    [8,2,0,8,5,2,8,7,5,10,2,5],
    [9,0,1,5,10,3,5,3,7,3,10,2],
    [9,8,2,9,2,1,8,7,2,10,2,5,7,5,2],
    [1,3,5,3,7,5],
    [0,8,7,0,7,1,1,7,5],
    [9,0,3,9,3,5,5,3,7],
    [9,8,7,5,9,7],
    [5,8,4,5,10,8,10,11,8],
    [5,0,4,5,11,0,5,10,11,11,3,0],
    [0,1,9,8,4,10,8,10,11,10,4,5],
    [10,11,4,10,4,5,11,3,4,9,4,1,3,1,4],
    [2,5,1,2,8,5,2,11,8,4,5,8],
    [0,4,11,0,11,3,4,5,11,2,11,1,5,1,11],
    [0,2,5,0,5,9,2,11,5,4,5,8,11,8,5],
    [9,4,5,2,11,3],
    [2,5,10,3,5,2,3,4,5,3,8,4],
    [5,10,2,5,2,4,4,2,0],
    [3,10,2,3,5,10,3,8,5,4,5,8,0,1,9],
    [5,10,2,5,2,4,1,9,2,9,4,2],
    [8,4,5,8,5,3,3,5,1],
    [0,4,5,1,0,5],
    [8,4,5,8,5,3,9,0,5,0,3,5],
    [9,4,5],
    [4,11,7,4,9,11,9,10,11],
    [0,8,3,4,9,7,9,11,7,9,10,11],
    [1,10,11,1,11,4,1,4,0,7,4,11],
    [3,1,4,3,4,8,1,10,4,7,4,11,10,11,4],
    [4,11,7,9,11,4,9,2,11,9,1,2],
    [9,7,4,9,11,7,9,1,11,2,11,1,0,8,3],
    [11,7,4,11,4,2,2,4,0],
    [11,7,4,11,4,2,8,3,4,3,2,4],
    [2,9,10,2,7,9,2,3,7,7,4,9],
    [9,10,7,9,7,4,10,2,7,8,7,0,2,0,7],
    [3,7,10,3,10,2,7,4,10,1,10,0,4,0,10],
    [1,10,2,8,7,4],
    [4,9,1,4,1,7,7,1,3],
    [4,9,1,4,1,7,0,8,1,8,7,1],
    [4,0,3,7,4,3],
    [4,8,7],
    [9,10,8,10,11,8],
    [3,0,9,3,9,11,11,9,10],
    [0,1,10,0,10,8,8,10,11],
    [3,1,10,11,3,10],
    [1,2,11,1,11,9,9,11,8],
    [3,0,9,3,9,11,1,2,9,2,11,9],
    [0,2,11,8,0,11],
    [3,2,11],
    [2,3,8,2,8,10,10,8,9],
    [9,10,2,0,9,2],
    [2,3,8,2,8,10,0,1,8,1,10,8],
    [1,10,2],
    [1,3,8,9,1,8],
    [0,9,1],
    [0,3,8],
    [],
]


def face_ambiguous(case: int) -> bool:
    """does the case have a face whose diagonal corners are alike and whose neighbours differ (four crossings)?"""
    ins = [(case >> c) & 1 for c in range(8)]
    return any(ins[cyc[0]] == ins[cyc[2]] and ins[cyc[1]] == ins[cyc[3]] and ins[cyc[0]] != ins[cyc[1]] for cyc in FACES)


def validate_table(rows) -> List[Tuple[int, str]]:
    """What any marching-cubes case table must satisfy, per case (returns the violations; [] = valid):
    the triangles use exactly the crossed cube edges; they form a manifold patch (no directed edge twice, interior edges
    traversed once in each direction); its boundary segments each lie in a cube face and pass every crossed edge once in
    and once out; all triangles of all cases share one orientation with respect to inside -> outside."""
    bad: List[Tuple[int, str]] = []
    signs = set()
    for case, row in enumerate(rows):
        row = [int(v) for v in row if int(v) >= 0]
        ins = [(case >> c) & 1 for c in range(8)]
        crossed = {e for e, (a, b) in enumerate(EDGES) if ins[a] != ins[b]}
        tris = [tuple(row[i:i + 3]) for i in range(0, len(row), 3)]
        if len(row) % 3 or {e for t in tris for e in t} != crossed or any(len(set(t)) != 3 for t in tris):
            bad.append((case, "edge set"))
            continue
        de = {}
        for t in tris:
            for i in range(3):
                de[(t[i], t[(i + 1) % 3])] = de.get((t[i], t[(i + 1) % 3]), 0) + 1
        und = {}
        for (a, b), v in de.items():
            und[frozenset((a, b))] = und.get(frozenset((a, b)), 0) + v
        if any(v > 1 for v in de.values()) or any(v > 2 for v in und.values()) or \
                any((b, a) not in de for (a, b) in de if und[frozenset((a, b))] == 2):
            bad.append((case, "not a consistently oriented manifold patch"))
            continue
        boundary = [(a, b) for (a, b) in de if und[frozenset((a, b))] == 1]
        cnt = {}
        for a, b in boundary:
            cnt[a] = cnt.get(a, 0) + 1
            cnt[b] = cnt.get(b, 0) + 1
        if any(not (_FACES_OF[a] & _FACES_OF[b]) for a, b in boundary) or any(cnt.get(e, 0) != 2 for e in crossed):
            bad.append((case, "boundary"))
            continue
        mid = lambda e: (CORNERS[EDGES[e][0]] + CORNERS[EDGES[e][1]]) / 2.0
        for t in tris:
            n = np.cross(mid(t[1]) - mid(t[0]), mid(t[2]) - mid(t[0]))
            d = np.zeros(3)
            for e in t:
                a, b = EDGES[e]
                d += (CORNERS[b] - CORNERS[a]) * (1 if ins[a] else -1)
            signs.add(float(np.sign(n @ d)))
    if len(signs) > 1:
        bad.append((-1, "mixed orientation"))
    return bad


def boundary_segments(row) -> set:
    """undirected boundary segments {cube edge, cube edge} of a case's triangle patch"""
    row = [int(v) for v in row if int(v) >= 0]
    cnt = {}
    for i in range(0, len(row), 3):
        t = row[i:i + 3]
        for j in range(3):
            k = frozenset((t[j], t[(j + 1) % 3]))
            cnt[k] = cnt.get(k, 0) + 1
    return {k for k, v in cnt.items() if v == 1}


def _pad(rows) -> Tuple[np.ndarray, np.ndarray]:
    mx = max(len(r) for r in rows)
    tab = -np.ones((256, mx), dtype=np.int8)
    for i, r in enumerate(rows):
        tab[i, :len(r)] = r
    return tab, np.array([len(r) // 3 for r in rows], dtype=np.uint8)


TABLE_CLASSIC, TABLE_WATERTIGHT = 0, 1
_classic_tab, _classic_n = _pad(CLASSIC_ROWS)
_water_tab, _water_n = build_tables()
assert _classic_tab.shape == _water_tab.shape == (256, 15)
# TABLES[t] = (tri_table int8 [256][15], n_tris uint8 [256])
TABLES = {TABLE_CLASSIC: (_classic_tab, _classic_n), TABLE_WATERTIGHT: (_water_tab, _water_n)}
TRI_TABLE, N_TRIS = TABLES[TABLE_CLASSIC]          # the default table
MAX_TRIS = TRI_TABLE.shape[1] // 3

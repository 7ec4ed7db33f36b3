"""Marching-cubes case tables, derived -- not transcribed -- from the cube's geometry.

The reference extracts meshes with PyMCubes (`mcubes.marching_cubes(sdf_i, level)`, model/diff_utils/util_3d.py:217),
a table-driven marching cubes over the 256 corner-sign cases.  PyMCubes is not in this image and cannot be fetched,
so its tables cannot be compared; these are built from first principles with one rule per step:

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


TRI_TABLE, N_TRIS = build_tables()
MAX_TRIS = TRI_TABLE.shape[1] // 3

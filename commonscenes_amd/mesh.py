"""SDF -> mesh on the MI355X behind the reference's `sdf_to_mesh` (model/diff_utils/util_3d.py:194-236).

The reference pulls every SDF to the host and runs PyMCubes (`mcubes.marching_cubes(sdf_i, level)`) per object, then
wraps the lists in a pytorch3d `Meshes`.  Here the whole batch is meshed by three HIP kernels (csrc/cs_mesh.hip: count,
vertices, faces -- scan + compaction, HBM-bound) with one small read-back of the per-block totals to size the outputs.
pytorch3d is not in this image: the result is a `Meshes` stand-in exposing what the reference's consumers read
(`verts_list()`, `faces_list()`, `textures.verts_features_list()`, iteration, `len`; helpers/util.py:298-300 iterates
the batch and converts each mesh with verts/faces).

Parity note: PyMCubes is absent from the image and unfetchable, so its output cannot be compared.  Since r4 the DEFAULT
case table is the classic Lorensen-Cline table in the 256-row form PyMCubes ships (mc_tables.TABLE_CLASSIC: the triangle
set of `mcubes.marching_cubes`, normals towards decreasing value as published); `table="watertight"` (or
CS_MC_TABLE=watertight) selects the table derived from the cube geometry (r2-r3's; normals towards increasing value).
Both mesh the same watertight surface with the same vertex set (one linearly interpolated vertex per crossed grid edge;
same patch boundaries on all 256 cases) and differ in how polygons are fanned; vertex / triangle ORDER against PyMCubes
stays unpinned.  The tests check each table's invariants, the two against each other, geometry
against analytic shapes, and bit-equality with the numpy oracle.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

import os

from . import lib as L
from .mc_tables import TABLE_CLASSIC, TABLE_WATERTIGHT
from .ops import _chk, _stream

Tensor = torch.Tensor


class Textures:
    def __init__(self, verts_rgb: List[Tensor]):
        self._rgb = verts_rgb

    def verts_features_list(self) -> List[Tensor]:
        return self._rgb


class Meshes:
    """What the reference reads from pytorch3d.structures.Meshes (verts_list / faces_list / textures, iteration)."""

    def __init__(self, verts: List[Tensor], faces: List[Tensor], textures: Optional[Textures] = None):
        self._verts, self._faces, self.textures = verts, faces, textures

    def verts_list(self) -> List[Tensor]:
        return self._verts

    def faces_list(self) -> List[Tensor]:
        return self._faces

    def __len__(self) -> int:
        return len(self._verts)

    def __getitem__(self, i) -> "Meshes":
        tx = Textures([self.textures._rgb[i]]) if self.textures is not None else None
        return Meshes([self._verts[i]], [self._faces[i]], tx)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _table_id(table) -> int:
    """None -> CS_MC_TABLE (default 'classic'); 'classic' / 'watertight' or the ids of mc_tables"""
    if table is None:
        table = os.environ.get("CS_MC_TABLE", "classic")
    t = {"classic": TABLE_CLASSIC, "watertight": TABLE_WATERTIGHT}.get(table, table)
    if t not in (TABLE_CLASSIC, TABLE_WATERTIGHT):
        raise ValueError(f"marching cubes table must be 'classic' or 'watertight', got {table!r}")
    return int(t)


def marching_cubes(sdf: Tensor, level: float, vert_div: float = 1.0, vert_shift: float = 0.0, table=None):
    """Batch marching cubes.  sdf: (B, n, n, n) fp32 on the HIP device -> (verts, faces, nv, nf): all objects'
    vertices [sum nv, 3] fp32 and triangles [sum nf, 3] int64 (ids local to each object), per-object counts (lists).
    table: 'classic' (default: the table PyMCubes ships) or 'watertight' (see the module docstring)."""
    tab = _table_id(table)
    _chk(sdf, "sdf")
    if sdf.dim() != 4 or sdf.shape[1] != sdf.shape[2] or sdf.shape[2] != sdf.shape[3]:
        raise L.CsError(f"marching_cubes: expected (B, n, n, n), got {tuple(sdf.shape)}")
    sdf = sdf.contiguous()
    nb, n = sdf.shape[0], sdf.shape[1]
    lib = L.load()
    bpo = lib.cs_mc_blocks_per_object(n)
    if bpo <= 0:
        raise L.CsError(f"marching_cubes: grid size {n} not supported (2..160)")
    dev = sdf.device
    sums = torch.empty((nb, bpo, 2), dtype=torch.int32, device=dev)
    L.check(lib.cs_mc_count(sdf.data_ptr(), nb, n, float(level), tab, sums.data_ptr(), _stream()), "cs_mc_count")
    tot = sums.sum(dim=1, dtype=torch.int64)                 # [nb, 2]
    base = torch.cumsum(tot, dim=0) - tot                    # exclusive, on the device
    tot_h = tot.cpu()                                        # the one read-back: output sizes
    nv = [int(v) for v in tot_h[:, 0]]
    nf = [int(v) for v in tot_h[:, 1]]
    verts = torch.empty((max(sum(nv), 1), 3), dtype=torch.float32, device=dev)
    faces = torch.empty((max(sum(nf), 1), 3), dtype=torch.int64, device=dev)
    ws = torch.empty((nb * n ** 3,), dtype=torch.int32, device=dev)
    vb, fb = base[:, 0].contiguous(), base[:, 1].contiguous()
    L.check(lib.cs_mc_emit(sdf.data_ptr(), nb, n, float(level), tab, sums.data_ptr(), vb.data_ptr(), fb.data_ptr(),
                           verts.data_ptr(), faces.data_ptr(), ws.data_ptr(), float(vert_div), float(vert_shift),
                           _stream()), "cs_mc_emit")
    return verts[:sum(nv)], faces[:sum(nf)], nv, nf


def sdf_to_mesh(sdf: Tensor, level: float = 0.02, color: Optional[Sequence[float]] = None, render_all: bool = False,
                table=None):
    """util_3d.py:194-236: (B,1,n,n,n) SDF -> Meshes with verts in [-0.5, 0.5) (verts / n_cell - .5), int64 faces and
    per-vertex colours (ones, or `color`).  Like the reference, at most 16 objects are meshed unless render_all."""
    bs, nc = sdf.shape[:2]
    assert nc == 1
    n_cell = sdf.shape[-1]
    nimg = bs
    if not render_all:
        if bs > 16:
            print("Warning! Will not return all meshes")
        nimg = min(bs, 16)
    if nimg == 0:
        return Meshes([], [], Textures([]))
    vol = sdf[:nimg, 0]
    if not vol.is_cuda:       # the reference takes the SDF wherever it lives (util_3d.py:211 copies it to the host);
        vol = vol.to(torch.device("cuda", torch.cuda.current_device()))     # here the work is on the device: upload it
    v, f, nv, nf = marching_cubes(vol.to(torch.float32), level, vert_div=float(n_cell), vert_shift=-0.5, table=table)
    verts = list(torch.split(v, nv))
    faces = list(torch.split(f, nf))
    rgb = []
    for vi in verts:
        t = torch.ones_like(vi)
        if color is not None:
            for c in range(3):
                t[:, c] = color[c]
        rgb.append(t)
    return Meshes(verts, faces, Textures(rgb))

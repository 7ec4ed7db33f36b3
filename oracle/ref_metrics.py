"""TEST INFRASTRUCTURE: numpy restatements of the reference's point-cloud metric kernels.

PARITY UNPINNED: the reference implementations are CUDA (extension/chamfer.cu, scripts/pytorch_structural_losses/src/
approxmatch.cu) and cannot run in this image; there are no golden vectors for them in the reference's tests.  Each
function restates the kernel it cites, loop for loop (vectorised over the independent index), in float32."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def chamfer_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    """extension/chamfer.cu:155-185: two NmDistanceGradKernel launches; the atomicAdd scatters are applied in index
    order here (the CUDA order is unspecified): own term first, then hits k ascending."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    ga, gb = np.zeros_like(xyz1), np.zeros_like(xyz2)

    def one(a, bp, g_own, i_own, g_oth, i_oth, out):
        for bi in range(b):
            for j in range(a.shape[1]):
                g = f32(g_own[bi, j] * f32(2))
                acc = g * (a[bi, j] - bp[bi, i_own[bi, j]])
                for k in np.nonzero(i_oth[bi] == j)[0]:
                    gk = f32(g_oth[bi, k] * f32(2))
                    acc = acc + (-(gk * (bp[bi, k] - a[bi, j])))
                out[bi, j] = acc
    one(xyz1, xyz2, g1, idx1, g2, idx2, ga)
    one(xyz2, xyz1, g2, idx2, g1, idx1, gb)
    return ga, gb


def _sqd(p, q):
    """[n,3] x [m,3] -> [n,m] squared distances, summed x, y, z like approxmatch.cu:49."""
    d = q[None, :, :] - p[:, None, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def approxmatch(xyz1, xyz2):
    """approxmatch.cu:3-182 -> match [b][m][n].  Row sums run over the other cloud in index order (np.cumsum-free
    sequential adds are emulated with a float32 loop over 1024-point tiles only where the order matters: the adds are
    done with np.add.reduce over axis in float32, which is pairwise -- so the comparison tolerance is ~1e-5)."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    multiL, multiR = (f32(1), f32(n // m)) if n >= m else (f32(m // n), f32(1))
    match = np.zeros((b, m, n), dtype=f32)
    for i in range(b):
        remainL = np.full(n, multiL, f32)
        remainR = np.full(m, multiR, f32)
        d2 = _sqd(xyz1[i].astype(f32), xyz2[i].astype(f32)).astype(f32)         # [n, m]
        for j in range(7, -2, -1):
            level = f32(-(4.0 ** j))
            e = np.exp((level * d2).astype(f32)).astype(f32)
            suml = f32(1e-9) + (e * remainR[None, :]).sum(1, dtype=f32)
            ratioL = (remainL / suml).astype(f32)
            sumr = (e * ratioL[:, None]).sum(0, dtype=f32) * remainR
            consumption = np.minimum(remainR / (sumr + f32(1e-9)), f32(1.0)).astype(f32)
            ratioR = (consumption * remainR).astype(f32)
            remainR = np.maximum(f32(0), remainR - sumr).astype(f32)
            w = (e * ratioL[:, None] * ratioR[None, :]).astype(f32)             # [n, m]
            match[i] += w.T
            remainL = np.maximum(f32(0), remainL - w.sum(1, dtype=f32)).astype(f32)
    return match


def matchcost(xyz1, xyz2, match):
    """approxmatch.cu:184-224."""
    out = np.zeros(xyz1.shape[0], dtype=np.float64)
    for i in range(xyz1.shape[0]):
        d = np.sqrt(_sqd(xyz1[i].astype(f32), xyz2[i].astype(f32)).astype(f32)).astype(f32)   # [n, m]
        out[i] = (match[i].T.astype(np.float64) * d).sum()
    return out


def matchcost_grad(xyz1, xyz2, match):
    """approxmatch.cu:229-320."""
    g1, g2 = np.zeros_like(xyz1, dtype=np.float64), np.zeros_like(xyz2, dtype=np.float64)
    for i in range(xyz1.shape[0]):
        diff = xyz1[i][:, None, :].astype(np.float64) - xyz2[i][None, :, :].astype(np.float64)    # [n, m, 3] x1 - x2
        inv = 1.0 / np.sqrt(np.maximum((diff ** 2).sum(-1), 1e-20))
        w = match[i].T.astype(np.float64) * inv                                                   # [n, m]
        g1[i] = (diff * w[..., None]).sum(1)
        g2[i] = (-diff * w[..., None]).sum(0)
    return g1, g2

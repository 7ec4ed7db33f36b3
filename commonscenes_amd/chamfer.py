"""Chamfer distance behind the reference's `extension/dist_chamfer.py` interface (`chamferDist()(a, b) ->
(dist1, dist2)`), the metric kernel scripts/eval_3dfront.py:24-25,394-397 needs for `--evaluate_diversity`.
Forward only: the reference's backward (chamfer.cu:136-182) serves training, which is out of scope."""
from __future__ import annotations

from typing import Tuple

import torch

from . import lib as L

Tensor = torch.Tensor


def nm_distance(xyz1: Tensor, xyz2: Tensor) -> Tuple[Tensor, Tensor]:
    """(dist [b, n] fp32, idx [b, n] int32): nearest neighbour of every xyz1 point in xyz2 (squared distance)."""
    for t, name in ((xyz1, "xyz1"), (xyz2, "xyz2")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3 or t.shape[-1] != 3:
            raise L.CsError(f"{name} must be a float32 [b, n, 3] tensor on the HIP device")
    if xyz1.shape[0] != xyz2.shape[0]:
        raise L.CsError("batch sizes differ")
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    L.check(L.load().cs_chamfer_nm_distance(xyz1.data_ptr(), xyz2.data_ptr(), dist.data_ptr(), idx.data_ptr(), b, n, m,
                                            torch.cuda.current_stream().cuda_stream), "cs_chamfer_nm_distance")
    return dist, idx


class chamferDist:
    """`chamferDist()(input1, input2) -> dist1, dist2` (extension/dist_chamfer.py:49-54); the neighbour indices of
    the last call are kept as `.idx1`, `.idx2` (the reference saves them for its backward)."""

    def __call__(self, input1: Tensor, input2: Tensor) -> Tuple[Tensor, Tensor]:
        return self.forward(input1, input2)

    @torch.no_grad()
    def forward(self, input1: Tensor, input2: Tensor) -> Tuple[Tensor, Tensor]:
        dist1, self.idx1 = nm_distance(input1, input2)
        dist2, self.idx2 = nm_distance(input2, input1)
        return dist1, dist2

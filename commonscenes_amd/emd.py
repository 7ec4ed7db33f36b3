"""Approximate earth mover's distance behind the reference's StructuralLosses interface
(scripts/pytorch_structural_losses/match_cost.py: `match_cost(seta, setb) -> cost [b]`, differentiable; backend ops
ApproxMatch / MatchCost / MatchCostGrad of src/structural_loss.cpp over src/approxmatch.cu), used by
scripts/compute_mmd_cov_1nn.py:56-62 as `emd = match_cost(sample, ref); emd / N`."""
from __future__ import annotations

from typing import Tuple

import torch

from . import lib as L

Tensor = torch.Tensor


def _chk(t: Tensor, name: str):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3 or t.shape[-1] != 3:
        raise L.CsError(f"{name} must be a float32 [b, n, 3] tensor on the HIP device")


def _s():
    return torch.cuda.current_stream().cuda_stream


def ApproxMatch(seta: Tensor, setb: Tensor) -> Tuple[Tensor, Tensor]:
    """(match [b, m, n], temp [b, (n+m)*2]): the nine-level auction of approxmatch.cu:3-182."""
    _chk(seta, "set1")
    _chk(setb, "set2")
    if seta.shape[0] != setb.shape[0]:
        raise L.CsError("batch sizes differ")
    seta, setb = seta.contiguous(), setb.contiguous()
    b, n, _ = seta.shape
    m = setb.shape[1]
    match = torch.empty((b, m, n), dtype=torch.float32, device=seta.device)
    temp = torch.empty((b, (n + m) * 2), dtype=torch.float32, device=seta.device)
    L.check(L.load().cs_emd_approxmatch(seta.data_ptr(), setb.data_ptr(), match.data_ptr(), temp.data_ptr(), b, n, m,
                                        _s()), "cs_emd_approxmatch")
    return match, temp


def MatchCost(seta: Tensor, setb: Tensor, match: Tensor) -> Tensor:
    seta, setb, match = seta.contiguous(), setb.contiguous(), match.contiguous()
    b, n, _ = seta.shape
    m = setb.shape[1]
    out = torch.empty((b,), dtype=torch.float32, device=seta.device)
    L.check(L.load().cs_emd_matchcost(seta.data_ptr(), setb.data_ptr(), match.data_ptr(), out.data_ptr(), b, n, m, _s()),
            "cs_emd_matchcost")
    return out


def MatchCostGrad(seta: Tensor, setb: Tensor, match: Tensor) -> Tuple[Tensor, Tensor]:
    seta, setb, match = seta.contiguous(), setb.contiguous(), match.contiguous()
    b, n, _ = seta.shape
    m = setb.shape[1]
    g1, g2 = torch.empty_like(seta), torch.empty_like(setb)
    L.check(L.load().cs_emd_matchcost_grad(seta.data_ptr(), setb.data_ptr(), match.data_ptr(), g1.data_ptr(),
                                           g2.data_ptr(), b, n, m, _s()), "cs_emd_matchcost_grad")
    return g1, g2


class MatchCostFunction(torch.autograd.Function):
    """match_cost.py:6-43."""

    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        match, _ = ApproxMatch(seta, setb)
        ctx.match = match
        return MatchCost(seta, setb, match)

    @staticmethod
    def backward(ctx, grad_output):
        seta, setb = ctx.saved_tensors
        grada, gradb = MatchCostGrad(seta, setb, ctx.match)
        g = grad_output.unsqueeze(1).unsqueeze(2)
        return grada * g, gradb * g


match_cost = MatchCostFunction.apply

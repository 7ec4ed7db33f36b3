"""Point-cloud metric kernels on the MI355X vs their numpy restatements (oracle/ref_metrics.py; parity
against the CUDA builds is unpinned -- they cannot run here), plus properties: Chamfer backward == autograd of the
gathered distances; approximate EMD conserves mass, is ~0 for identical clouds and approaches the exact optimal
assignment cost (scipy's Hungarian solver, what compute_mmd_cov_1nn.py:35-52 uses as the slow path)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _clouds(b, n, m, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((b, n, 3), generator=g) - 0.5, torch.rand((b, m, 3), generator=g) - 0.5


def test_chamfer_backward_vs_oracle_and_autograd():
    from commonscenes_amd.chamfer import chamferDist
    from oracle import ref_metrics as RM
    a, b = _clouds(3, 300, 211, 0)
    a_d, b_d = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    cd = chamferDist()
    d1, d2 = cd(a_d, b_d)
    w1, w2 = torch.rand(d1.shape, generator=torch.Generator().manual_seed(1)), torch.rand(d2.shape, generator=torch.Generator().manual_seed(2))
    (d1 * w1.cuda()).sum().add((d2 * w2.cuda()).sum()).backward()
    torch.cuda.synchronize()
    ga, gb = RM.chamfer_backward(a.numpy(), b.numpy(), w1.numpy(), w2.numpy(), cd.idx1.cpu().numpy(), cd.idx2.cpu().numpy())
    assert np.array_equal(a_d.grad.cpu().numpy(), ga) and np.array_equal(b_d.grad.cpu().numpy(), gb)   # deterministic gather
    # the same gradient from torch autograd over explicitly gathered neighbours (float64)
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    i1, i2 = cd.idx1.cpu().long(), cd.idx2.cpu().long()
    e1 = ((a64 - torch.gather(b64, 1, i1[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    e2 = ((b64 - torch.gather(a64, 1, i2[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    ((e1 * w1.double()).sum() + (e2 * w2.double()).sum()).backward()
    assert rel_l2(a_d.grad, a64.grad) < 1e-6 and rel_l2(b_d.grad, b64.grad) < 1e-6
    assert rel_l2(d1, e1) < 1e-6


@pytest.mark.parametrize("n,m", [(256, 256), (300, 150), (128, 384), (1500, 1500)])
def test_approx_emd_vs_oracle(n, m):
    from commonscenes_amd import emd
    from oracle import ref_metrics as RM
    a, b = _clouds(2, n, m, 3)
    match, _ = emd.ApproxMatch(a.cuda(), b.cuda())
    cost = emd.MatchCost(a.cuda(), b.cuda(), match)
    g1, g2 = emd.MatchCostGrad(a.cuda(), b.cuda(), match)
    torch.cuda.synchronize()
    rm = RM.approxmatch(a.numpy(), b.numpy())
    assert match.shape == (2, m, n)
    assert rel_l2(match, torch.from_numpy(rm)) < 2e-4          # fast hardware exp vs numpy's exp, through nine levels
    assert rel_l2(cost, torch.from_numpy(RM.matchcost(a.numpy(), b.numpy(), rm))) < 2e-4
    # cost / gradient kernels on the SAME match: tight
    mm = match.cpu().numpy()
    assert rel_l2(cost, torch.from_numpy(RM.matchcost(a.numpy(), b.numpy(), mm))) < 2e-6
    r1, r2 = RM.matchcost_grad(a.numpy(), b.numpy(), mm)
    assert rel_l2(g1, torch.from_numpy(r1)) < 2e-6 and rel_l2(g2, torch.from_numpy(r2)) < 2e-6
    # mass conservation: every point of the smaller-multiplicity side ships (almost) all of its mass
    big, small = max(n, m) // min(n, m), 1
    ship1 = match.sum(1).cpu()       # per xyz1 point
    ship2 = match.sum(2).cpu()       # per xyz2 point
    assert float(ship1.max()) <= (small if n >= m else big) + 1e-3 and float(ship2.max()) <= (big if n >= m else small) + 1e-3
    assert float(ship1.mean()) > 0.9 * (small if n >= m else big)


def test_match_cost_interface_properties():
    """match_cost(sample, ref) (match_cost.py:45): ~0 for identical clouds, permutation invariant, close to the optimal
    assignment, differentiable."""
    from scipy.optimize import linear_sum_assignment
    from commonscenes_amd.emd import match_cost
    a, b = _clouds(2, 512, 512, 7)
    same = match_cost(a.cuda(), a.cuda())
    assert float(same.max()) / 512 < 2e-3
    c = match_cost(a.cuda(), b.cuda())
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(0))
    c2 = match_cost(a[:, perm].cuda(), b.cuda())
    torch.cuda.synchronize()
    assert rel_l2(c2, c) < 1e-4
    for i in range(2):
        d = torch.cdist(a[i].double(), b[i].double()).numpy()
        r, cidx = linear_sum_assignment(d)
        opt = d[r, cidx].sum()
        # the auction is an approximation from above: never below the optimum, within ~30 % of it on uniform clouds
        assert opt <= float(c[i]) * (1 + 1e-6) and float(c[i]) < 1.5 * opt, (opt, float(c[i]))
    a_d = a.cuda().requires_grad_(True)
    match_cost(a_d, b.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert a_d.grad.shape == a.shape and torch.isfinite(a_d.grad).all() and float(a_d.grad.abs().max()) > 0

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


# ---------------------------------------------------------------------------------------------------------------------
# r6: hand-derived known-answer tests read off the reference's CUDA sources (they cannot be compiled
# here: cuda_runtime.h / ATen-CUDA), so these are the only pins N4 can get without a CUDA box.  Every expected value below
# was derived by hand from the cited lines and is exactly representable (or a closed form in one exp).
# ---------------------------------------------------------------------------------------------------------------------
def test_chamfer_kat_tie_breaking_follows_the_scan_order():
    """extension/chamfer.cu:12-134 (NmDistanceKernel): inside a 512-target batch the scan keeps the FIRST of equidistant
    targets (`k==0 || d<best`, strict), and a later batch replaces the running result only if strictly closer
    (`result[...]>best`, :126) -- so among equidistant nearest targets the LOWEST index wins, also across the 512 boundary
    and in the `end_k & 3` tail (:113-122)."""
    from commonscenes_amd.chamfer import chamferDist
    cd = chamferDist()
    # one query at the origin, three targets at distance 1: index 0
    a = torch.zeros(1, 1, 3)
    b = torch.tensor([[[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0]]])
    d1, d2 = cd(a.cuda(), b.cuda())
    assert d1.cpu().tolist() == [[1.0]] and cd.idx1.cpu().tolist() == [[0]]
    assert d2.cpu().tolist() == [[1.0, 1.0, 1.0]] and cd.idx2.cpu().tolist() == [[0, 0, 0]]
    # 600 far targets; equidistant nearest ones at indices 3 (first batch), 510 (the unrolled part's last quad), 515 (second
    # batch) and 599 (the second batch's 88 = 4 * 22 targets end on a whole quad; 598 % 4 tail of a 599-target run below)
    m = 600
    far = torch.zeros(1, m, 3)
    far[0, :, 0] = 50.0 + torch.arange(m, dtype=torch.float32)
    for idx, p in ((3, (0.0, 2.0, 0.0)), (510, (0.0, -2.0, 0.0)), (515, (2.0, 0.0, 0.0)), (599, (0.0, 0.0, 2.0))):
        far[0, idx] = torch.tensor(p)
    q = torch.zeros(1, 2, 3)
    q[0, 1] = torch.tensor([0.0, 0.0, 1.5])               # second query: index 599 is strictly nearest (0.25 < ...)
    d1, _ = cd(q.cuda(), far.cuda())
    assert cd.idx1.cpu().tolist() == [[3, 599]] and d1.cpu().tolist() == [[4.0, 0.25]]
    # drop target 3: the tie is now 510 (batch one) vs 515 (batch two) -> 510; drop 510 too -> 515 vs 599 -> 515
    far[0, 3] = torch.tensor([1000.0, 0.0, 0.0])
    cd(q[:, :1].cuda(), far.cuda())
    assert cd.idx1.cpu().tolist() == [[510]]
    far[0, 510] = torch.tensor([1000.0, 0.0, 0.0])
    cd(q[:, :1].cuda(), far.cuda())
    assert cd.idx1.cpu().tolist() == [[515]]
    # a 599-target cloud: the second batch has 87 = 4 * 21 + 3 targets, its last three go through the scalar tail loop
    tail = far[:, :599].clone()
    tail[0, 515] = torch.tensor([1000.0, 0.0, 0.0])
    tail[0, 597] = torch.tensor([0.0, 3.0, 0.0])
    tail[0, 598] = torch.tensor([0.0, -3.0, 0.0])
    d1, _ = cd(q[:, :1].cuda(), tail.cuda())
    assert cd.idx1.cpu().tolist() == [[597]] and d1.cpu().tolist() == [[9.0]]


def test_chamfer_kat_gradient_of_a_three_point_cloud():
    """extension/chamfer.cu:155-185 (NmDistanceGradKernel, called twice with the clouds swapped): g = 2 * grad_dist,
    grad_xyz1[j] += g (p_j - q_idx), grad_xyz2[idx] -= the same.  Clouds p = (0,0,0), (1,0,0), (0,2,0) and
    q = (0,0,1), (1,1,0), (0,2,3): idx1 = [0, 1, 1] (d1 = 1, 1, 2), idx2 = [0, 1, 2] (d2 = 1, 1, 9); with grad_dist1 = (1, 2, 3)
    and grad_dist2 = (4, 5, 6) the two passes add up, by hand, to the integers below."""
    from commonscenes_amd.chamfer import chamferDist
    p = torch.tensor([[[0.0, 0, 0], [1.0, 0, 0], [0, 2.0, 0]]]).cuda().requires_grad_(True)
    q = torch.tensor([[[0.0, 0, 1.0], [1.0, 1.0, 0], [0, 2.0, 3.0]]]).cuda().requires_grad_(True)
    cd = chamferDist()
    d1, d2 = cd(p, q)
    assert d1.detach().cpu().tolist() == [[1.0, 1.0, 2.0]] and d2.detach().cpu().tolist() == [[1.0, 1.0, 9.0]]
    assert cd.idx1.cpu().tolist() == [[0, 1, 1]] and cd.idx2.cpu().tolist() == [[0, 1, 2]]
    g1 = torch.tensor([[1.0, 2.0, 3.0]]).cuda()
    g2 = torch.tensor([[4.0, 5.0, 6.0]]).cuda()
    ((d1 * g1).sum() + (d2 * g2).sum()).backward()
    assert p.grad.cpu().tolist() == [[[0.0, 0.0, -10.0], [0.0, -14.0, 0.0], [-6.0, 6.0, -36.0]]]
    assert q.grad.cpu().tolist() == [[[0.0, 0.0, 10.0], [6.0, 8.0, 0.0], [0.0, 0.0, 36.0]]]


def test_approxmatch_kat_nine_level_schedule():
    """scripts/pytorch_structural_losses/src/approxmatch.cu:24-27: `for (j=7;j>-2;j--) level=-powf(4,j)` runs NINE levels,
    -4^7 ... -4^0, -4^-1, and the `j==-2 -> level=0` branch is dead.  Two decoupled pairs pin that by hand:
      p1 = q1 (squared distance 0): the first level ships everything -- suml = 1e-9 + 1 = 1 (fp32), ratioL = 1, sumr = 1,
        consumption = 1, match += 1, both remainders drop to 0 and every later level adds 0;
      p0 / q0 at squared distance 121: exp(level * 121) underflows to exactly 0 for the eight levels -4^7 ... -1, so they add
        nothing and leave remainL = remainR = 1, ratioR = 1; ONLY the ninth level (-1/4) sees E = exp(-30.25):
        suml = 1e-9 + E, ratioL = 1 / suml, match += E * ratioL * ratioR = E / (1e-9 + E).
    The cross pairs are ~1e6 apart (exp = 0 everywhere).  Eight levels would leave match[0][0] = 0, a tenth (level 0) would
    ship the remaining mass and leave ~1.  match is laid out [b][m][n] (:157)."""
    from commonscenes_amd import emd
    a = torch.tensor([[[0.0, 0.0, 0.0], [1000.0, 0.0, 0.0]]])
    b = torch.tensor([[[0.0, 0.0, 11.0], [1000.0, 0.0, 0.0]]])
    match, _ = emd.ApproxMatch(a.cuda(), b.cuda())
    torch.cuda.synchronize()
    mm = match.cpu().double().numpy()[0]
    E = float(np.exp(-30.25))
    want = E / (1e-9 + E)                                    # 7.2856e-5
    assert mm[1, 1] == 1.0 and mm[0, 1] == 0.0 and mm[1, 0] == 0.0
    assert abs(mm[0, 0] - want) < 2e-4 * want, (mm[0, 0], want)
    # and the numpy restatement agrees with the same hand-derived values (pins oracle/ref_metrics.py on this KAT)
    from oracle import ref_metrics as RM
    rm = RM.approxmatch(a.numpy(), b.numpy())[0]
    assert rm[1, 1] == 1.0 and abs(float(rm[0, 0]) - want) < 1e-5 * want

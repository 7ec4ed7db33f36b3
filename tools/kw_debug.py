"""debug: where does the K-wave kernel's pair hand-over differ from the fp32 hand-over?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).cuda()

def decode_pair(t, cols):
    """interleaved pair [rows][cols as fp32 slots] -> (hi, lo) fp32 [rows][cols]"""
    h = t.view(torch.float16).view(t.shape[0], cols // 16, 4, 8).float()
    hi = torch.stack([h[:, :, 0], h[:, :, 2]], dim=2).reshape(t.shape[0], cols)
    lo = torch.stack([h[:, :, 1], h[:, :, 3]], dim=2).reshape(t.shape[0], cols)
    return hi, lo

m, c = 512, 672
x = rnd(m, c, seed=11)
pw = ops.pack_weight(rnd(c, c, seed=12, scale=c ** -0.5), rnd(c, seed=13), math=L.MATH_F16X3)
res = rnd(m, c, seed=14) + 0.5
w2 = ops.pack_weight(rnd(c, c, seed=15, scale=c ** -0.5), rnd(c, seed=16), math=L.MATH_F16X3)
for tile in (10, 3):
    y0 = ops.linear(x, pw, res=res, tile=tile)
    yp = ops.linear(x, pw, res=res, out_pair=16.0, tile=tile)
    torch.cuda.synchronize()
    assert isinstance(yp, ops.Pair16), type(yp)
    hi, lo = decode_pair(yp.t, c)
    v = y0 * 16.0
    ehi = v.half().float()
    elo = (v - ehi).half().float()
    print(f"producer tile {tile}: hi mismatches {(hi != ehi).sum().item()}, lo mismatches {(lo != elo).sum().item()} of {hi.numel()}")
    bad = (hi != ehi).nonzero()
    if len(bad):
        print("  first bad (row, col):", bad[:8].tolist())
    for ctile in (10, 3):
        zp = ops.linear(yp, w2, tile=ctile)
        zf = ops.linear(y0, w2, tile=ctile)
        torch.cuda.synchronize()
        d = (zp != zf)
        print(f"  consumer tile {ctile}: pair vs fp32 differing elements {d.sum().item()} max abs {(zp - zf).abs().max().item():.3e}")
        if d.any():
            idx = d.nonzero()
            print("    rows", sorted(set(idx[:, 0].tolist()))[:12], "cols", sorted(set(idx[:, 1].tolist()))[:12])

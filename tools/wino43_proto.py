#!/usr/bin/env python
"""What-if (r5): F(4,3) along W instead of F(2,3) -- 6 position GEMMs per 4 outputs (13.5 of 27 multiply-adds), transforms
with constants up to 8 and 1/24.  Numerics only: transforms in torch (fp32 input transform as the GroupNorm kernel would do,
fp64 weight transform), the position GEMMs on the existing 3x3x1 kernels, against fp64 and against F(2,3) / the direct form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from commonscenes_amd import lib as L, ops, synth

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def conv43(x, w, b, scale16=True):
    W = x.shape[3]
    xp = F.pad(x, (0, 0, 1, 3))                                   # x[-1] .. x[W + 2]
    d = [xp[:, :, :, j:j + W:4] for j in range(6)]                # d_j = x[4t - 1 + j]
    bt = BT.float().cuda()
    vs = [sum(bt[q, j] * d[j] for j in range(6) if BT[q, j] != 0).contiguous() for q in range(6)]
    g = w.double()
    us = [sum(G[q, j] * g[..., j] for j in range(3)) for q in range(6)]
    ms = []
    for q in range(6):
        pw = ops.pack_weight(us[q].float().unsqueeze(-1).contiguous(), None, math=L.MATH_F16X3)
        amax = float(vs[q].abs().max())
        ms.append(ops.conv_gemm(vs[q], pw, a_scale=ops.bound_a_scale(amax)))
    at = AT.float().cuda()
    ys = [sum(at[e, q] * ms[q] for q in range(6) if AT[e, q] != 0) for e in range(4)]
    y = torch.stack(ys, dim=4)                                    # [N, D, H, W/4, 4, C]
    return y.reshape(*ms[0].shape[:3], -1, ms[0].shape[-1]) + b


for sp, cin, cout in [((16, 16, 16), 224, 224), ((16, 8, 8), 448, 448), ((16, 4, 4), 672, 672), ((16, 4, 4), 1344, 672)]:
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    x = F.silu(synth.tensor_device(f"xs{sp}{cin}", (2, *sp, cin), 1.5))          # activations as the convs see them
    ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    yd = ops.conv_gemm(x, pw)
    y43 = conv43(x, w, b)
    torch.cuda.synchronize()
    e = lambda y: float((y.double() - ref).norm() / ref.norm())
    em = float(((y43.double() - ref).abs().max()) / ref.abs().max())
    print(f"{sp} {cin}->{cout}: rel-L2 vs fp64 direct {e(yd):.2e}  F(4,3) along W {e(y43):.2e}  (max-abs / max |ref| {em:.2e})", flush=True)

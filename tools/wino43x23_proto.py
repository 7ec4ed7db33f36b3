#!/usr/bin/env python
"""What-if (r6, VERDICT r5 next #2 iv): NESTED Winograd F(4,3) along W x F(2,3) along H for the 3x3x3 stride-1 convs --
24 position GEMMs with 3x1x1 kernels per (4 x 2) output tile: 6 * 4 * 3 = 72 multiply-adds per 8 outputs = 9 of the direct
form's 27 (F(4,3) along W alone: 13.5).  NUMERICS ONLY, before any kernel is built: transforms in torch (fp32 input
transform as a GroupNorm producer would do it -- along W, then along H; fp64 weight transform, split once), the position
GEMMs on the existing F16X3 kernels with 3x1x1 weights, against fp64, the direct form and F(4,3) along W.
Gate for building (VERDICT): per-conv rel-L2 under the 2e-6 long-contraction gate with margin for the 30 240-term conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from commonscenes_amd import lib as L, ops, synth

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def lin(M, xs):
    """rows of M applied to the list xs (fp32 on the device, zero coefficients skipped)"""
    return [sum(float(M[q, j]) * xs[j] for j in range(len(xs)) if M[q, j] != 0) for q in range(M.shape[0])]


def conv_nested(x, w, b):
    N, D, H, W, C = x.shape
    xp = F.pad(x, (0, 0, 1, 3, 1, 1))                             # W: x[-1] .. x[W + 2]; H: x[-1] .. x[H]
    dw = [xp[:, :, :, j:j + W:4] for j in range(6)]               # along W: d_j = x[.., 4t - 1 + j]   -> [N, D, H + 2, W/4, C]
    vw = lin(BT4, dw)                                             # six W positions, still on the padded H grid
    us_all, ms = [], {}
    g = w.double()                                                # [cout, cin, kd, kh, kw]
    for qw in range(6):
        dh = [vw[qw][:, :, j:j + H:2] for j in range(4)]          # along H: e_j = v[.., 2s - 1 + j, ..] -> [N, D, H/2, W/4, C]
        vh = lin(BT2, dh)
        gw = sum(G4[qw, j] * g[..., j] for j in range(3))         # [cout, cin, kd, kh]
        for qh in range(4):
            u = sum(G2[qh, j] * gw[..., j] for j in range(3))     # [cout, cin, kd]
            pw = ops.pack_weight(u.float().reshape(*u.shape, 1, 1).contiguous(), None, math=L.MATH_F16X3)
            v = vh[qh].contiguous()
            ms[(qw, qh)] = ops.conv_gemm(v, pw, a_scale=ops.bound_a_scale(float(v.abs().max())))
    # output transform: along H (2 outputs from 4), then along W (4 outputs from 6)
    yw = []
    for qw in range(6):
        yh = lin(AT2, [ms[(qw, qh)] for qh in range(4)])          # two H outputs
        yw.append(torch.stack(yh, dim=3).reshape(N, D, H, W // 4, -1))      # [N, D, H/2, 2, W/4, C] -> [N, D, H, W/4, C]
    yo = lin(AT4, yw)
    y = torch.stack(yo, dim=4)                                    # [N, D, H, W/4, 4, C]
    return y.reshape(N, D, H, W, -1) + b


def conv43(x, w, b):
    W = x.shape[3]
    xp = F.pad(x, (0, 0, 1, 3))
    d = [xp[:, :, :, j:j + W:4] for j in range(6)]
    vs = [v.contiguous() for v in lin(BT4, d)]
    g = w.double()
    ms = []
    for q in range(6):
        u = sum(G4[q, j] * g[..., j] for j in range(3))
        pw = ops.pack_weight(u.float().unsqueeze(-1).contiguous(), None, math=L.MATH_F16X3)
        ms.append(ops.conv_gemm(vs[q], pw, a_scale=ops.bound_a_scale(float(vs[q].abs().max()))))
    y = torch.stack(lin(AT4, ms), dim=4)
    return y.reshape(*ms[0].shape[:3], -1, ms[0].shape[-1]) + b


for sp, cin, cout in [((16, 16, 16), 224, 224), ((16, 8, 8), 448, 448), ((16, 8, 8), 1120, 448), ((16, 4, 4), 672, 672),
                      ((16, 4, 4), 1344, 672)]:
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    x = F.silu(synth.tensor_device(f"xs{sp}{cin}", (2, *sp, cin), 1.5))
    ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    yd = ops.conv_gemm(x, pw)
    y43 = conv43(x, w, b)
    yn = conv_nested(x, w, b)
    torch.cuda.synchronize()
    e = lambda y: float((y.double() - ref).norm() / ref.norm())
    em = float(((yn.double() - ref).abs().max()) / ref.abs().max())
    print(f"{sp} {cin}->{cout}: rel-L2 vs fp64  direct {e(yd):.2e}  F(4,3)_W {e(y43):.2e}  F(4,3)_W x F(2,3)_H {e(yn):.2e}  "
          f"(max-abs / max |ref| {em:.2e})", flush=True)

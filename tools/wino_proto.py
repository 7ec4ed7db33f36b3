#!/usr/bin/env python
"""What-if prototype (r5): the 3x3x3 convs as Winograd F(2,3) ALONG W only -- 4 position GEMMs of a 3x3x1 conv over
(D, H, W/2) with transformed weights [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2] on transformed inputs [d0-d2, d1+d2, d2-d1, d1-d3],
then y_even = m0+m1+m2, y_odd = m1-m2-m3: 18 instead of 27 multiply-adds per output.  Existing kernels only (the per-tap
gather path takes the 3x3x1 kernels), transforms in torch: accuracy vs fp64 and the GEMM time vs the direct conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from commonscenes_amd import lib as L, ops, synth

SHAPES = [((16, 16, 16), 224, 224), ((16, 16, 16), 672, 224), ((16, 8, 8), 448, 448), ((16, 8, 8), 1120, 448),
          ((16, 4, 4), 672, 672), ((16, 4, 4), 1344, 672)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def wino_pack(w):
    g = w.double()
    g0, g1, g2 = g[..., 0], g[..., 1], g[..., 2]
    us = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2]
    return [ops.pack_weight(u.float().unsqueeze(-1).contiguous(), None, math=L.MATH_F16X3) for u in us]


def wino_in(x):
    W = x.shape[3]
    xp = F.pad(x, (0, 0, 1, 1))
    d = [xp[:, :, :, j:j + W:2] for j in range(4)]
    return [(d[0] - d[2]).contiguous(), (d[1] + d[2]).contiguous(), (d[2] - d[1]).contiguous(), (d[1] - d[3]).contiguous()]


def wino_out(m, bias):
    y = torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], dim=4)      # [N, D, H, W/2, 2, C]
    return y.reshape(*m[0].shape[:3], -1, m[0].shape[-1]) + bias


for sp, cin, cout in SHAPES:
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    pws = wino_pack(w)
    # accuracy at two samples against fp64
    x = synth.tensor_device(f"xs{sp}{cin}", (2, *sp, cin), 1.0)
    ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
    yd = ops.conv_gemm(x, pw)
    yw = wino_out([ops.conv_gemm(v, p) for v, p in zip(wino_in(x), pws)], b)
    torch.cuda.synchronize()
    ed = float((yd.double() - ref).norm() / ref.norm())
    ew = float((yw.double() - ref).norm() / ref.norm())
    # time at 64 samples
    x = synth.tensor_device(f"xb{sp}{cin}", (64, *sp, cin), 1.0)
    vs = wino_in(x)
    hn = ops.groupnorm(x, torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 32, 1e-5, L.ACT_SILU, split16=True)
    t_pre = timeit(lambda: ops.conv_gemm(hn, pw))
    t_f32 = timeit(lambda: ops.conv_gemm(x, pw))
    t_w = timeit(lambda: [ops.conv_gemm(v, p) for v, p in zip(vs, pws)])
    fl = 2.0 * 64 * sp[0] * sp[1] * sp[2] * cin * cout * 27
    print(f"{sp} {cin}->{cout}: rel-L2 vs fp64 direct {ed:.2e} winograd-W {ew:.2e} | direct pre-split {t_pre:.3f} ms "
          f"({fl / t_pre / 1e9:.0f} TF/s) fp32-in {t_f32:.3f} | 4 position GEMMs {t_w:.3f} ms "
          f"({fl / t_w / 1e9:.0f} direct-equivalent TF/s, {fl * 2 / 3 / t_w / 1e9:.0f} executed)", flush=True)

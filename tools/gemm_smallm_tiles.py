#!/usr/bin/env python
"""3x3x3 convs of ONE object (CFG batch 2; SM_BATCH overrides): the K-sliced slab kernel on every tile that can carry it --
128x224 (2), 256x224 (4), 256x128 (6), 256x64 (7) -- over slice counts.  The 224-column tiles of the plan write
slices x M x cout partial tiles (44 MB for a 4^3-level conv at 32 slices, as much as its weights); narrower column tiles
reach the same workgroup count with fewer slices.   SM_BATCH=2 python tools/gemm_smallm_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
NB = int(os.environ.get("SM_BATCH", "2"))
SHAPES = [((16, 4, 4), 672, 672), ((16, 4, 4), 1344, 672), ((16, 8, 8), 448, 448), ((16, 8, 8), 1120, 448),
          ((16, 16, 16), 224, 224), ((16, 16, 16), 672, 224)]
BM = {2: 128, 4: 256, 6: 256, 7: 256}
BN = {2: 224, 4: 224, 6: 128, 7: 64}


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for sp, cin, cout in SHAPES:
    x = synth.tensor_device(f"x{sp}{cin}", (NB, *sp, cin), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    M = NB * sp[0] * sp[1] * sp[2]
    res = {}
    timeit(lambda: ops.conv_gemm(x, pw))
    res["auto"] = timeit(lambda: ops.conv_gemm(x, pw))
    for tile in (2, 4, 6, 7):
        tiles = ((M + BM[tile] - 1) // BM[tile]) * ((cout + BN[tile] - 1) // BN[tile])
        slots = 512 if tile == 2 else 256
        for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 14, 16, 21, 24, 32, 42):
            if tiles * s > slots * 1.3 or tiles * s < slots * 0.3:
                continue
            try:
                res[f"t{tile}/s{s}"] = timeit(lambda: ops.conv_gemm(x, pw, tile=tile, splitk=s if s > 1 else 0))
            except Exception:
                res[f"t{tile}/s{s}"] = float("nan")
    best = min((k for k in res if res[k] == res[k]), key=res.get)
    fl = 2.0 * M * cin * cout * 27
    print(f"M={M:6d} K={cin * 27:6d} N={cout:4d} | " + " ".join(f"{n}:{v:6.1f}" for n, v in res.items()) +
          f" | best {best} {res[best]:.1f} us ({fl / res[best] / 1e6:.0f} TF) auto {res['auto']:.1f}", flush=True)

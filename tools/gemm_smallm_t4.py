#!/usr/bin/env python
"""3x3x3 convs of small / medium batches: the 256x224 slab tile (one workgroup per CU) with K slices against the 128x224
tile (two per CU) the plan uses there.   SM_BATCH=14 python tools/gemm_smallm_t4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
NB = int(os.environ.get("SM_BATCH", "14"))
SHAPES = [((16, 4, 4), 672, 672), ((16, 4, 4), 1344, 672), ((16, 8, 8), 448, 448), ((16, 8, 8), 1120, 448),
          ((16, 16, 16), 224, 224), ((16, 16, 16), 672, 224)]


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
    for tile in (2, 4):
        for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 21, 24, 32):
            tiles = ((M + (127 if tile == 2 else 255)) // (128 if tile == 2 else 256)) * (cout // 224)
            if tiles * s > (640 if tile == 2 else 320):
                continue
            try:
                res[f"t{tile}/s{s}"] = timeit(lambda: ops.conv_gemm(x, pw, tile=tile, splitk=s if s > 1 else 0))
            except Exception as e:
                res[f"t{tile}/s{s}"] = float("nan")
    best = min((k for k in res if res[k] == res[k]), key=res.get)
    fl = 2.0 * M * cin * cout * 27
    print(f"M={M:6d} K={cin * 27:6d} N={cout:4d} | " + " ".join(f"{n}:{v:6.1f}" for n, v in res.items()) +
          f" | best {best} {res[best]:.1f} us ({fl / res[best] / 1e6:.0f} TF) auto {res['auto']:.1f}", flush=True)

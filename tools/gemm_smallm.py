#!/usr/bin/env python
"""Small-batch GEMM shapes (BASELINE configs[1]: one object, CFG batch 2): what is the fastest way to run each -- the
128x224 tile with split-K factor s, the 64x64 tile, the 128x128 tile -- against what cs_conv_gemm picks by itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
NB = int(os.environ.get("SM_BATCH", "2"))
SPLITS = tuple(int(v) for v in os.environ.get("SM_SPLITS", "1,2,4,8,16,32").split(","))   # e.g. 1,2,3,4,5,6,7,8,10,12,16
ONLY3 = bool(os.environ.get("SM_ONLY3"))                                                   # 3x3x3 shapes only
SHAPES = [  # (d,h,w), cin, cout, k
    ((16, 4, 4), 672, 672, 3), ((16, 4, 4), 1344, 672, 3), ((16, 8, 8), 448, 448, 3), ((16, 8, 8), 1120, 448, 3),
    ((16, 16, 16), 224, 224, 3), ((16, 16, 16), 672, 224, 3), ((16, 16, 16), 448, 448, 3),
    ((256, 1, 1), 672, 672, 1), ((256, 1, 1), 672, 2016, 1), ((256, 1, 1), 2688, 672, 1), ((256, 1, 1), 672, 5376, 1),
    ((1024, 1, 1), 448, 448, 1), ((1024, 1, 1), 448, 1344, 1), ((1024, 1, 1), 1792, 448, 1), ((1024, 1, 1), 448, 3584, 1),
    ((4096, 1, 1), 448, 224, 1),
]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for sp, cin, cout, k in SHAPES:
    if ONLY3 and k != 3:
        continue
    x = synth.tensor_device(f"x{sp}{cin}", (NB, *sp, cin), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}{k}", (cout, cin, k, k, k) if k > 1 else (cout, cin), (3.0 / (cin * k ** 3)) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    M = NB * sp[0] * sp[1] * sp[2]
    nk = k ** 3 * ((cin + 15) // 16)
    res = {"auto": timeit(lambda: ops.conv_gemm(x, pw))}
    for s in SPLITS:
        if s <= nk // 2 and cout % 224 == 0:
            res[f"t2/s{s}"] = timeit(lambda: ops.conv_gemm(x, pw, tile=2, splitk=s if s > 1 else None) if s > 1 else ops.conv_gemm(x, pw, tile=2, splitk=0))
    res["t3"] = timeit(lambda: ops.conv_gemm(x, pw, tile=3, splitk=0))
    res["t1"] = timeit(lambda: ops.conv_gemm(x, pw, tile=1, splitk=0))
    best = min(res, key=res.get)
    fl = 2.0 * M * cin * cout * k ** 3
    print(f"M={M:5d} K={cin * k ** 3:6d} N={cout:5d} taps={k ** 3:2d} | " + " ".join(f"{n}:{v:6.1f}" for n, v in res.items()) +
          f" | best {best} {res[best]:.1f} us ({fl / res[best] / 1e6:.0f} TF) auto {res['auto']:.1f}", flush=True)

#!/usr/bin/env python
"""K-sliced 1x1x1 / token GEMMs of small batches: the 256x224 tile against the 128x224 tile the sliced dispatch uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
NB = int(os.environ.get("SM_BATCH", "2"))
SHAPES = [(256, 2688, 672), (256, 672, 672), (256, 1344, 672), (1024, 1792, 448), (1024, 448, 448), (1024, 1120, 448), (4096, 448, 224)]


def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows, cin, cout in SHAPES:
    x = synth.tensor_device(f"x{rows}{cin}", (NB * rows, cin), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin), (3.0 / cin) ** 0.5)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    nk = (cin + 15) // 16
    res = {}
    timeit(lambda: ops.conv_gemm(x, pw))
    res["auto"] = timeit(lambda: ops.conv_gemm(x, pw))
    for tile in (2, 4):
        for s in (1, 2, 4, 8, 16):
            if s > 1 and s > nk // 8:
                continue
            res[f"t{tile}/s{s}"] = timeit(lambda: ops.conv_gemm(x, pw, tile=tile, splitk=s if s > 1 else 0))
    res["t3"] = timeit(lambda: ops.conv_gemm(x, pw, tile=3, splitk=0))
    best = min(res, key=res.get)
    print(f"M={NB * rows:6d} K={cin:5d} N={cout:4d} | " + " ".join(f"{n}:{v:6.1f}" for n, v in res.items()) +
          f" | best {best} {res[best]:.1f} auto {res['auto']:.1f}", flush=True)

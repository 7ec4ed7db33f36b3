#!/usr/bin/env python
"""Batch-64 conv shapes: GroupNorm + SiLU -> 3x3x3 conv in the direct form (pre-split pair -> slab kernel) against the
Winograd-W route (transformed operand -> four position GEMMs in one launch -> output transform), GEMM part and
GroupNorm-apply part timed separately.  usage (GPU box): python tools/wino_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L
if os.environ.get("CS_LIB"):          # a what-if build of the library (variants/*.so)
    L._LIB = L.load(os.environ["CS_LIB"])
from commonscenes_amd import ops, synth

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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


for sp, cin, cout in SHAPES:
    nb = NB if sp[1] < 16 else max(1, NB // 2) if cin == 224 else NB      # (the 224 -> 224 level-0 convs run in the prefix)
    rows = sp[0] * sp[1] * sp[2]
    x = synth.tensor_device(f"x{sp}{cin}", (nb, *sp, cin), 1.0)
    g = synth.tensor_device(f"g{cin}", (cin,), 1.0)
    b = synth.tensor_device(f"b{cin}", (cin,), 0.1)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    pw = ops.pack_weight_wino(ops.pack_weight(w, synth.tensor_device(f"c{cout}", (cout,), 0.1), math=L.MATH_F16X3), w)
    emb = synth.tensor_device(f"e{cout}", (nb, cout), 1.0)
    wv = ops.wants_wino(nb, *sp, pw)
    if not wv:
        print(f"{sp} {cin}->{cout} batch {nb}: not eligible"); continue
    hn = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, split16=True)
    v = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, wino=wv)
    t_gd = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, split16=True))
    t_gw = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, wino=wv))
    t_d = timeit(lambda: ops.conv_gemm(hn, pw, rowvec=emb, rv_rows=rows, stats=True))
    t_w = timeit(lambda: ops.conv_gemm(v, pw, rowvec=emb, rv_rows=rows, stats=True))
    fl = 2.0 * nb * rows * cin * cout * 27
    print(f"{sp} {cin}->{cout} batch {nb}: conv direct {t_d:.3f} ms ({fl / t_d / 1e9:.0f} TF/s) winograd-W {t_w:.3f} ms "
          f"({fl / t_w / 1e9:.0f} direct-equivalent TF/s, F({wv},3)) | GroupNorm (stats + apply) pair {t_gd:.3f} ms wino {t_gw:.3f} ms | "
          f"sum {t_d + t_gd:.3f} -> {t_w + t_gw:.3f} ms ({100 * (t_w + t_gw) / (t_d + t_gd) - 100:+.1f} %)", flush=True)

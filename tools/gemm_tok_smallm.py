#!/usr/bin/env python
"""Token / 1x1x1 GEMMs of the transformer blocks and skip connections at SMALL batches (SM_BATCH = CFG batch: 2 = one
object, 14 = the reference's mini-batch of 7): every tile the library has x K slices, against the automatic choice.
At these sizes the C x C GEMMs run at 75-170 TF/s (profiles/r04_gemm_table_c7.txt) where the same shapes reach 200-320 at
CFG batch 64.    SM_BATCH=14 python tools/gemm_tok_smallm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
NB = int(os.environ.get("SM_BATCH", "14"))
# (tokens per sample, cin, cout, kind, operand) -- kind: res = bias + residual epilogue, geglu = fused gate
SHAPES = ((1024, 448, 448, "res", "f32"), (1024, 448, 448, "res", "pair"), (1024, 448, 1344, "", "pair"),
          (1024, 1792, 448, "res", "pair"), (1024, 448, 3584, "geglu", "pair"),
          (256, 672, 672, "res", "f32"), (256, 672, 672, "res", "pair"), (256, 672, 2016, "", "pair"),
          (256, 2688, 672, "res", "pair"), (256, 672, 5376, "geglu", "pair"),
          (4096, 448, 224, "", "f32"), (4096, 672, 224, "", "f32"), (1024, 1120, 448, "", "f32"), (1024, 672, 448, "", "f32"),
          (256, 1344, 672, "", "f32"), (256, 1120, 672, "", "f32"))
TILES = (1, 2, 3, 4, 6, 7)
SPLITS = (0, 2, 4, 8)


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (tok, cin, cout, kind, opnd) in SHAPES:
    x = synth.tensor_device(f"x{tok}{cin}", (NB, tok, cin), 1.0)
    if opnd == "pair":
        g = torch.ones(cin, device="cuda"); bt = torch.zeros(cin, device="cuda")
        if cin <= 2048:
            x = ops.layernorm(x, g, bt, pair_scale=256.0)
        else:       # wider than the LayerNorm producer takes: a pair-emitting GEMM epilogue makes the operand
            w0 = synth.tensor_device(f"w0{cin}", (cin, 448), 0.05)
            x = ops.linear(synth.tensor_device(f"x0{tok}", (NB, tok, 448), 1.0),
                           ops.pack_weight(w0, torch.zeros(cin, device="cuda"), math=L.MATH_F16X3), out_pair=16.0)
            assert isinstance(x, ops.Pair16)
    r = synth.tensor_device(f"r{tok}{cout}", (NB, tok, cout), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin), 0.05)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    if kind == "geglu":
        pw, kw = ops.pack_geglu_weight(w, b), dict(act=L.ACT_GEGLU)
    else:
        pw, kw = ops.pack_weight(w, b, math=L.MATH_F16X3), (dict(res=r) if kind == "res" else {})
    res = {"auto": timeit(lambda: ops.linear(x, pw, **kw))}
    for tile in TILES:
        for s in SPLITS:
            if s and (tile in (1, 3) or kind == "geglu"):
                continue
            try:
                res[f"t{tile}" + (f"/s{s}" if s else "")] = timeit(lambda: ops.linear(x, pw, tile=tile, splitk=s, **kw))
            except Exception:
                pass
    res["auto"] = min(res["auto"], timeit(lambda: ops.linear(x, pw, **kw)))      # (again: the first one warms the clocks)
    best = min(res, key=res.get)
    fl = 2.0 * NB * tok * cin * cout
    print(f"M={NB * tok:6d} {cin:4d}->{cout:4d} {kind:5s} {opnd:4s} | " + " ".join(f"{n}:{v:5.1f}" for n, v in res.items()) +
          f" | best {best} {res[best]:.1f} us ({fl / res[best] / 1e6:.0f} TF) auto {res['auto']:.1f} ({fl / res['auto'] / 1e6:.0f} TF)",
          flush=True)

"""Token / 1x1x1 GEMM shapes of the transformer blocks at CFG batch 64: the one-tile-per-workgroup kernels (tiles 4, 2)
against the persistent ping-pong kernel (tile 5, csrc/cs_gemm_pw.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
B = 64
SHAPES = ((1024, 448, 448, "res"), (1024, 448, 1344, ""), (1024, 1792, 448, "res"), (1024, 448, 3584, "geglu"),
          (256, 672, 672, "res"), (256, 672, 2016, ""), (256, 2688, 672, "res"), (256, 672, 5376, "geglu"),
          (4096, 448, 224, "res"), (1024, 1120, 448, ""))
for (tok, cin, cout, kind) in SHAPES:
    x = synth.tensor_device(f"x{tok}{cin}", (B, tok, cin), 1.0)
    r = synth.tensor_device(f"r{tok}{cout}", (B, tok, cout), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin), 0.05)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    if kind == "geglu":
        pw, kw = ops.pack_geglu_weight(w, b), dict(act=L.ACT_GEGLU)
    else:
        pw, kw = ops.pack_weight(w, b, math=L.MATH_F16X3), (dict(res=r) if kind == "res" else {})
    line = f"tok={tok:5d} {cin:4d}->{cout:4d} {kind:5s}: "
    for tile in (4, 2, 5):
        try:
            ops.linear(x, pw, tile=tile, **kw); torch.cuda.synchronize()
        except L.CsError:
            line += f"tile{tile}     n/a          | "
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear(x, pw, tile=tile, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"tile{tile} {ms*1e3:7.1f} us {2.0*B*tok*cin*cout/ms/1e9:6.1f} TF | "
    print(line, flush=True)

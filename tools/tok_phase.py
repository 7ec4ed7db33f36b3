#!/usr/bin/env python
"""Phase budget of the token / 1x1x1 GEMMs at CFG batch TP_BATCH (default 64 = the headline): each shape x tile launched
TP_N times in a row, read from a rocprofv3 kernel trace with tools/rocpd_sequence.py.  Against the product library and
the -DCS_ABLATE=1024 (no epilogue) / 2048 (no K loop) builds:  CS_LIB=variants/libcs_ablate1024.so python tools/tok_phase.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L
if os.environ.get("CS_LIB"):
    L._LIB = L.load(os.environ["CS_LIB"])
from commonscenes_amd import ops, synth
NB = int(os.environ.get("TP_BATCH", "64"))
N = int(os.environ.get("TP_N", "20"))
TILES = [int(v) for v in os.environ.get("TP_TILES", "2,4").split(",")]
SHAPES = ((1024, 448, 448, "res", "pair"), (1024, 448, 1344, "", "pair"), (1024, 1792, 448, "res", "pair"),
          (1024, 448, 3584, "geglu", "pair"), (256, 672, 672, "res", "pair"), (256, 672, 5376, "geglu", "pair"))
for (tok, cin, cout, kind, opnd) in SHAPES:
    x = synth.tensor_device(f"x{tok}{cin}", (NB, tok, cin), 1.0)
    g = torch.ones(cin, device="cuda"); bt = torch.zeros(cin, device="cuda")
    if cin <= 2048:
        x = ops.layernorm(x, g, bt, pair_scale=256.0)
    else:
        w0 = synth.tensor_device(f"w0{cin}", (cin, 448), 0.05)
        x = ops.linear(synth.tensor_device(f"x0{tok}", (NB, tok, 448), 1.0),
                       ops.pack_weight(w0, torch.zeros(cin, device="cuda"), math=L.MATH_F16X3), out_pair=16.0)
    r = synth.tensor_device(f"r{tok}{cout}", (NB, tok, cout), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin), 0.05)
    b = synth.tensor_device(f"b{cout}", (cout,), 0.1)
    if kind == "geglu":
        pw, kw = ops.pack_geglu_weight(w, b), dict(act=L.ACT_GEGLU)
    else:
        pw, kw = ops.pack_weight(w, b, math=L.MATH_F16X3), (dict(res=r) if kind == "res" else {})
    if kind != "res":
        kw["out_pair"] = 16.0                      # as in the model: the next GEMM's operand
    for tile in TILES:
        for _ in range(N):
            ops.linear(x, pw, tile=tile, splitk=0, **kw)
        torch.cuda.synchronize()
        print(f"M={NB * tok} {cin}->{cout} {kind} tile {tile}", flush=True)

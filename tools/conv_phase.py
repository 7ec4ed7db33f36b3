#!/usr/bin/env python
"""Phase budget of the K-sliced 3x3x3 slab convs at one object (CFG batch 2): each shape launched CP_N times in a row, to
be read from a rocprofv3 kernel trace with tools/rocpd_sequence.py (GPU-side durations).  Run against the product
library and against -DCS_ABLATE=1024 (no epilogue) / 2048 (no K loop) / 3072 (neither) builds of cs_gemm_f16x3.hip:
    CS_LIB=variants/libcs_ablate1024.so python tools/conv_phase.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L
if os.environ.get("CS_LIB"):
    L._LIB = L.load(os.environ["CS_LIB"])
from commonscenes_amd import ops, synth
NB = int(os.environ.get("CP_BATCH", "2"))
N = int(os.environ.get("CP_N", "20"))
SPLITS = [int(v) for v in os.environ.get("CP_SPLITS", "0").split(",")]        # 0 = the automatic plan
SHAPES = [((16, 4, 4), 672, 672), ((16, 4, 4), 1344, 672), ((16, 8, 8), 448, 448), ((16, 8, 8), 1120, 448),
          ((16, 16, 16), 224, 224), ((16, 16, 16), 448, 224), ((16, 16, 16), 672, 224)]
for sp, cin, cout in SHAPES:
    x = synth.tensor_device(f"x{sp}{cin}", (NB, *sp, cin), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
    pw = ops.pack_weight(w, synth.tensor_device(f"b{cout}", (cout,), 0.1), math=L.MATH_F16X3)
    for s in SPLITS:
        for _ in range(N):
            ops.conv_gemm(x, pw, splitk=s or None, stats=bool(os.environ.get("CP_STATS")))
        torch.cuda.synchronize()
        print(f"M={NB * sp[0] * sp[1] * sp[2]} K={27 * cin} N={cout} splitk={s or 'auto'}", flush=True)

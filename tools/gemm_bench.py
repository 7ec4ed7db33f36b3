#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the UNet's conv / linear shapes at CFG batch 64.

    python tools/gemm_bench.py [--math fp32|f16x3|both] [--batch 64] [--iters 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth

SHAPES = [
    # name, (d,h,w), cin, cout, k
    ("conv 16^3   224->224", (16, 16, 16), 224, 224, 3),
    ("conv 16^3   672->224", (16, 16, 16), 672, 224, 3),
    ("conv 16x8x8 448->448", (16, 8, 8), 448, 448, 3),
    ("conv 16x8x8 1120->448", (16, 8, 8), 1120, 448, 3),
    ("conv 16x4x4 672->672", (16, 4, 4), 672, 672, 3),
    ("conv 16x4x4 1344->672", (16, 4, 4), 1344, 672, 3),
    ("lin  1024tok 448->7168", (1024, 1, 1), 448, 7168, 1),
    ("lin  1024tok 3584->448", (1024, 1, 1), 3584, 448, 1),
    ("lin  256tok 672->2016", (256, 1, 1), 672, 2016, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--math", default="both")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--tile", type=int, default=0)
    a = ap.parse_args()
    modes = ["fp32", "f16x3"] if a.math == "both" else [a.math]
    for si, (name, sp, cin, cout, k) in enumerate(SHAPES):
        if a.only >= 0 and si != a.only:
            continue
        x = synth.tensor_device(f"x{name}", (a.batch, *sp, cin), 1.0)
        wshape = (cout, cin, k, k, k) if k > 1 else (cout, cin)
        w = synth.tensor_device(f"w{name}", wshape, (3.0 / (cin * k ** 3)) ** 0.5)
        b = synth.tensor_device(f"b{name}", (cout,), 0.1)
        flops = 2.0 * a.batch * sp[0] * sp[1] * sp[2] * cout * cin * k ** 3
        line = f"{name:26s} M={a.batch * sp[0] * sp[1] * sp[2]:7d} K={cin * k ** 3:6d} N={cout:5d} "
        for mode in modes:
            pw = ops.pack_weight(w, b, math=L.MATH_F16X3 if mode == "f16x3" else L.MATH_FP32)
            ops.conv_gemm(x, pw, tile=a.tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv_gemm(x, pw, tile=a.tile)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            line += f"| {mode}: {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF/s "
        print(line, flush=True)


if __name__ == "__main__":
    main()

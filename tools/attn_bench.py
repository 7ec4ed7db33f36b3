#!/usr/bin/env python
"""Flash-attention kernel timings at the UNet's shapes (batch 64 = 32 objects x CFG): 5 blocks at 1024 tokens /
dh 56 and 6 blocks at 256 tokens / dh 84 per step.  Reports algorithmic TFLOP/s (4 * nb * heads * N^2 * dh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
for n, c, heads, nb in ((1024, 448, 8, 0), (256, 672, 8, 0), (512, 448, 8, 0), (64, 672, 8, 0), (4096, 256, 1, 16)):
    nb = nb or a.batch                      # (the last row: the VQ decoder's mid attention, 16 objects per call)
    qkv = synth.tensor_device(f"qkv{n}", (nb, n, 3 * c), 1.0)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    dh = c // heads
    line = f"N={n:5d} dh={dh:3d} nb={nb}: "
    for name, math in (("fp32", L.MATH_FP32), ("f16x3", L.MATH_F16X3), ("f16(opt-in)", L.MATH_F16)):
        ops.attention(q, k, v, heads, dh ** -0.5, math=math)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.attention(q, k, v, heads, dh ** -0.5, math=math)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        line += f"{name} {ms * 1e3:8.1f} us {4.0 * nb * heads * n * n * dh / ms / 1e9:7.1f} TF/s | "
    print(line, flush=True)

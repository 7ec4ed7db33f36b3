#!/usr/bin/env python
"""HBM-bound kernels of the path at the benchmark's shapes (UNet batch 64 = 32 objects x CFG): achieved GB/s of
ALGORITHMIC bytes (every element read / written once as the op requires) against the 8 TB/s HBM3E peak
(MI355X_MICROARCH.md).  GroupNorm = stats pass (read) + apply pass (read + write)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth

PEAK = 8000.0   # GB/s
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
B = a.batch


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3


def report(name, nbytes, sec):
    gbs = nbytes / sec / 1e9
    print(f"{name:58s} {nbytes / 1e6:9.1f} MB {sec * 1e6:9.1f} us {gbs:8.0f} GB/s  {gbs / PEAK * 100:5.1f} % of HBM peak", flush=True)


for (d, h, w, c) in ((16, 16, 16, 224), (16, 16, 16, 448), (16, 8, 8, 448), (16, 8, 8, 896), (16, 4, 4, 672), (16, 4, 4, 1344)):
    x = synth.tensor_device(f"x{d}{h}{c}", (B, d, h, w, c), 1.0)
    g = synth.tensor_device("g", (c,), 0.2, 1.0)
    b = synth.tensor_device("b", (c,), 0.1)
    n = x.numel() * 4
    report(f"GroupNorm32+SiLU  [{B},{d},{h},{w},{c}] (read, read+write)", 3 * n,
           timed(lambda: ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU)))
for (tok, c) in ((1024, 448), (256, 672)):
    x = synth.tensor_device(f"t{tok}", (B, tok, c), 1.0)
    g = synth.tensor_device("g", (c,), 0.2, 1.0)
    b = synth.tensor_device("b", (c,), 0.1)
    report(f"LayerNorm         [{B},{tok},{c}] (read + write)", 2 * x.numel() * 4, timed(lambda: ops.layernorm(x, g, b)))
for (d, h, w, ca, cb) in ((16, 16, 16, 224, 224), (16, 8, 8, 448, 448), (16, 4, 4, 672, 672)):
    xa = synth.tensor_device("ca", (B, d, h, w, ca), 1.0)
    xb = synth.tensor_device("cb", (B, d, h, w, cb), 1.0)
    report(f"concat channels   [{B},{d},{h},{w},{ca}+{cb}] (read + write)", 2 * (xa.numel() + xb.numel()) * 4,
           timed(lambda: ops.concat_channels(xa, xb)))
x = synth.tensor_device("dx", (B // 2, 3, 16, 16, 16), 1.0)
e = synth.tensor_device("de", (B, 3, 16, 16, 16), 1.0)
report(f"CFG + DDIM update [{B // 2} objects] (x, 2 eps in; x_prev out)", (x.numel() * 2 + e.numel()) * 4,
       timed(lambda: ops.ddim_cfg_update(x, e, 0.5, 0.6, 0.0, 0.7071, 3.0, True, want_pred_x0=False)))

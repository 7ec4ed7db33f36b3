#!/usr/bin/env python
"""Does a GroupNorm run faster when its samples are processed in chunks small enough for the statistics pass to leave
the chunk in the Infinity Cache (256 MiB) for the apply pass?  GroupNorm(+SiLU) -> split16 pair at the UNet's batch-64
shapes, whole batch vs chunks of 32 / 16 / 8 samples (per-sample results cannot change: statistics are per sample).
In a loop with a 400 MB scratch write between iterations, so that nothing is resident from the previous iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth

B = 64
scratch = torch.empty(100 * 1024 * 1024, dtype=torch.float32, device="cuda")


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        scratch.fill_(1.0)                       # evict
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


for (d, h, w, c) in ((16, 16, 16, 224), (16, 16, 16, 448), (16, 16, 16, 672), (16, 8, 8, 448), (16, 8, 8, 896), (16, 8, 8, 1120)):
    x = synth.tensor_device(f"x{d}{h}{c}", (B, d, h, w, c), 1.0)
    g, b = synth.tensor_device("g", (c,), 0.2, 1.0), synth.tensor_device("b", (c,), 0.1)
    line = f"[{B},{d},{h},{w},{c}] {x.numel() * 4 / 1e6:7.1f} MB: "
    for chunk in (64, 32, 16, 8):
        def run():
            for i in range(0, B, chunk):
                ops.groupnorm(x[i:i + chunk], g, b, 32, 1e-5, L.ACT_SILU, split16=True)
        line += f"chunk {chunk:2d}: {timed(run):7.1f} us | "
    print(line, flush=True)

#!/usr/bin/env python
"""Where do the 16-25 us of a small-M 1-tap GEMM go?  K sweep at fixed M x N on the 64x64 / 128x128 tiles: intercept =
launch + prologue + epilogue, slope = cost per 16-wide K chunk.   SM_M=512 SM_N=672 python tools/gemm_smallm_ksweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
M, N = int(os.environ.get("SM_M", "512")), int(os.environ.get("SM_N", "672"))


def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


r = synth.tensor_device("r", (M, N), 1.0)
for opnd in ("f32", "pair"):
    for tile in (3, 1):
        line = f"M={M} N={N} tile {tile} {opnd}: "
        for K in (16, 64, 224, 448, 672, 1344, 2688):
            x = synth.tensor_device(f"x{K}", (M, K), 1.0)
            if opnd == "pair":
                if K > 2048:
                    continue
                x = ops.layernorm(x, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), pair_scale=256.0)
            pw = ops.pack_weight(synth.tensor_device(f"w{K}", (N, K), 0.05), synth.tensor_device("b", (N,), 0.1), math=L.MATH_F16X3)
            line += f"K={K}: {timeit(lambda: ops.linear(x, pw, tile=tile, splitk=0, res=r)):5.1f} | "
        print(line, flush=True)
# floors: a trivial kernel of the same launch path
y = synth.tensor_device("y", (M, N), 1.0)
print(f"copy_rows {M}x{N}: {timeit(lambda: ops.copy_rows(y, r)):5.1f} us   layernorm: "
      f"{timeit(lambda: ops.layernorm(y, torch.ones(N, device='cuda'), torch.zeros(N, device='cuda'))):5.1f} us")

"""The batch-64 conv shapes of tools/gemm_bench.py fed the way the UNet feeds them: GroupNorm+SiLU emitting the pre-split
fp16 hi/lo operand pair -> F16X3 conv (the slab kernel's PRE instantiation).  usage (GPU box): python tools/conv_pre_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L
if os.environ.get("CS_LIB"):          # a what-if build of the library (variants/*.so)
    L._LIB = L.load(os.environ["CS_LIB"])
from commonscenes_amd import ops, synth

SHAPES = [("conv 16^3   224->224", (16, 16, 16), 224, 224), ("conv 16^3   672->224", (16, 16, 16), 672, 224),
          ("conv 16x8x8 448->448", (16, 8, 8), 448, 448), ("conv 16x8x8 1120->448", (16, 8, 8), 1120, 448),
          ("conv 16x4x4 672->672", (16, 4, 4), 672, 672), ("conv 16x4x4 1344->672", (16, 4, 4), 1344, 672)]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pre-only", action="store_true")
    a = ap.parse_args()
    nb, iters = 64, a.iters
    for si, (name, sp, cin, cout) in enumerate(SHAPES):
        if a.only >= 0 and si != a.only:
            continue
        x = synth.tensor_device(f"x{name}", (nb, *sp, cin), 1.0)
        g = synth.tensor_device(f"g{name}", (cin,), 1.0)
        b = synth.tensor_device(f"b{name}", (cin,), 0.1)
        w = synth.tensor_device(f"w{name}", (cout, cin, 3, 3, 3), (3.0 / (cin * 27)) ** 0.5)
        pw = ops.pack_weight(w, synth.tensor_device(f"c{name}", (cout,), 0.1), math=L.MATH_F16X3)
        flops = 2.0 * nb * sp[0] * sp[1] * sp[2] * cout * cin * 27
        line = f"{name:24s} "
        for label, s16 in (("pre-split", True),) + (() if a.pre_only else (("fp32 in", False),)):
            hn = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, split16=s16)
            ops.conv_gemm(hn, pw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.conv_gemm(hn, pw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += f"| {label}: {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF/s "
        print(line, flush=True)


if __name__ == "__main__":
    main()

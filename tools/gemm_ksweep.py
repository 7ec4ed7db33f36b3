#!/usr/bin/env python
"""Fixed cost per workgroup of the F16X3 conv kernel: time vs K (input channels) at fixed M, N -> linear fit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
B = int(os.environ.get("KS_BATCH", "64"))
pts = []
for taps in (3, 1):
    for cin in (16, 32, 64, 128, 224, 448, 672):
        x = synth.tensor_device(f"x{cin}", (B, 16, 16, 16, cin), 1.0)
        shp = (224, cin, 3, 3, 3) if taps == 3 else (224, cin)
        w = synth.tensor_device(f"w{cin}{taps}", shp, 0.05)
        b = synth.tensor_device("b", (224,), 0.1)
        pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
        ops.conv_gemm(x, pw); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_gemm(x, pw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ksteps = (27 if taps == 3 else 1) * ((cin + 15) // 16)
        print(f"taps={27 if taps == 3 else 1:2d} cin={cin:4d} k-steps={ksteps:5d}  {ms * 1e3:9.1f} us   {ms * 1e3 / ksteps:7.3f} us/k-step(all 4 rounds)", flush=True)

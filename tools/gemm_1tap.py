import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth
B = 64
for (tok, cin, cout) in ((1024, 448, 448), (1024, 448, 1344), (1024, 1792, 448), (256, 672, 672), (256, 672, 2016), (256, 2688, 672)):
    x = synth.tensor_device(f"x{tok}{cin}", (B, tok, cin), 1.0)
    r = synth.tensor_device(f"r{tok}{cout}", (B, tok, cout), 1.0)
    w = synth.tensor_device(f"w{cin}{cout}", (cout, cin), 0.05)
    b = synth.tensor_device("b", (cout,), 0.1)
    pw = ops.pack_weight(w, b, math=L.MATH_F16X3)
    line = f"tok={tok} {cin}->{cout}: "
    for tile in (4, 2):
        ops.linear(x, pw, res=r, tile=tile); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear(x, pw, res=r, tile=tile)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"tile{tile} {ms*1e3:7.1f} us {2.0*B*tok*cin*cout/ms/1e9:6.1f} TF | "
    print(line, flush=True)

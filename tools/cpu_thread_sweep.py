"""Time the CPU oracle's full-width UNet forward (CFG batch 4) at several thread counts (GPU-box host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import synth
from commonscenes_amd.unet import unet_param_shapes
from oracle import ref_torch as R

cfg = dict(R.UNET_FULL, dims=3, use_spatial_transformer=True)
sd = {k: v.cpu() for k, v in synth.synth_state_dict(unet_param_shapes(cfg), device="cuda").items()}
x = synth.gaussian_like("sweep:x", (4, 3, 16, 16, 16))
ctx = synth.gaussian_like("sweep:c", (4, 1, 1280))
t = torch.full((4,), 501, dtype=torch.long)
print("cpu_count", os.cpu_count())
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.no_grad():
        R.unet_forward(sd, cfg, x[:2], t[:2], ctx[:2])
        t0 = time.perf_counter()
        R.unet_forward(sd, cfg, x, t, ctx)
        dt = time.perf_counter() - t0
    print(f"threads {n:4d}: UNet fwd batch 4 = {dt:.2f} s  ({dt / 4:.2f} s/sample)", flush=True)

#!/usr/bin/env python
"""End-to-end timings of the other BASELINE configs on one MI355X (synthetic weights):
  C2: 1 object, 50-step DDIM (+ decode);  C3: 32 objects, 100-step DDIM + VQ-VAE decode to 64^3."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import synth
from commonscenes_amd.ddim import DDIMSampler
from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes
from oracle.ref_torch import DIFFUSION, UNET_FULL, VQ_FULL, register_schedule

ap = argparse.ArgumentParser()
ap.add_argument("--math", default="f16x3")
ap.add_argument("--configs", default="C2,C3")
a = ap.parse_args()
cfg = dict(UNET_FULL, dims=3, use_spatial_transformer=True)
df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math(a.math)
df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device="cuda"))
vq = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math(a.math)
vq.load_state_dict(synth.synth_state_dict(vqvae_param_shapes(VQ_FULL, 8192, 3), device="cuda"))
sch = register_schedule(**DIFFUSION)


class M:
    num_timesteps = 1000
    device = torch.device("cuda")
    alphas_cumprod = sch["alphas_cumprod"]

    def apply_model(self, x, t, c):
        return df(x, t, c_crossattn=[c])

    def apply_model_cfg(self, x, t, c_in):
        return df.forward_cfg(x, t, c_in)


def run(B, S, tag):
    x_T = synth.gaussian_like("e2e:x", (1, 3, 16, 16, 16)).cuda().repeat(B, 1, 1, 1, 1)
    c = synth.gaussian_like("e2e:c", (B, 1, 1280)).cuda()
    uc = synth.gaussian_like("e2e:uc", (B, 1, 1280)).cuda()
    for rep in range(2):           # first pass warms up (weight packing, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat, _ = DDIMSampler(M()).sample(S=S, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T,
                                         verbose=False, unconditional_guidance_scale=3.0,
                                         unconditional_conditioning=uc, eta=0.0,
                                         max_steps=(3 if rep == 0 else None))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sdf = torch.cat([vq.decode_no_quant(lat[i:i + 8]) for i in range(0, B, 8)])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{tag}: B={B} S={S} math={a.math}: DDIM {t1 - t0:.3f} s ({(t1 - t0) / S * 1e3:.2f} ms/step, "
          f"{S / (t1 - t0):.2f} steps/s), decode {t2 - t1:.3f} s ({(t2 - t1) / B * 1e3:.1f} ms/object), "
          f"sdf {tuple(sdf.shape)} finite={bool(torch.isfinite(sdf).all())}", flush=True)


if "C2" in a.configs:
    run(1, 50, "C2")
if "C3" in a.configs:
    run(32, 100, "C3")

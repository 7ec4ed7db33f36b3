#!/usr/bin/env python
"""VQ-VAE decode of 32 objects (decode_no_quant, 16 at a time): wall time and a per-GEMM-shape table (HIP events around
every launch, ops.GEMM_PROFILE).      python tools/decode_bench.py [--objects 32] [--iters 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import configs as K, ops, synth
from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=32)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
vq = VQVAE(K.VQVAE_DDCONFIG, K.VQVAE_N_EMBED, K.VQVAE_EMBED_DIM, device="cuda")
vq.load_state_dict(synth.synth_state_dict(vqvae_param_shapes(K.VQVAE_DDCONFIG, K.VQVAE_N_EMBED, K.VQVAE_EMBED_DIM), device="cuda"))
lat = synth.gaussian_like("bench:lat", (a.objects, 3, 16, 16, 16), scale=0.8).cuda()
vq.decode_no_quant(lat)
torch.cuda.synchronize()
ops.GEMM_PROFILE = prof = []
t0 = time.perf_counter()
for _ in range(a.iters):
    vq.decode_no_quant(lat)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
ops.GEMM_PROFILE = None
agg = {}
for r in prof:
    k = (r["taps"], r["m"], r["k"], r["n"], r["tile"], r.get("slab", 0), bool(r.get("pre")))
    t = agg.setdefault(k, [0, 0.0, 0.0])
    t[0] += 1; t[1] += r["e0"].elapsed_time(r["e1"]); t[2] += r["flops"]
print(f"decode of {a.objects} objects: {dt * 1e3:.2f} ms ({dt * 1e3 / a.objects:.3f} ms/object); GEMM launches "
      f"{sum(t[1] for t in agg.values()) / a.iters:.2f} ms of it")
print(f"{'taps':>4} {'M':>9} {'K':>6} {'N':>5} tile slab pre {'calls':>5} {'ms/decode':>9} {'TF/s':>7}")
for k, t in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:4d} {k[1]:9d} {k[2]:6d} {k[3]:5d} {k[4]:4d} {k[5]:4d} {int(k[6]):3d} {t[0] // a.iters:5d} {t[1] / a.iters:9.3f} {t[2] / t[1] / 1e9:7.1f}")

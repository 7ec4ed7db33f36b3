#!/usr/bin/env python
"""Benchmark: DDIM denoise steps/sec of the CommonScenes shape branch on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one classifier-free-guided DDIM step over this rank's 32 objects: a UNet forward at batch 64
(413.5 M parameters, 3x16^3 latents, one 1280-d context token per sample) + the fused CFG/DDIM update
(BASELINE.json configs[2], SURVEY 8d "C3").  Objects are independent, so N GPUs hold N x 32 objects
(weak scaling, no per-step collective); rank 0 produces the conditioning with the scene-graph GCN and
broadcasts the packed [x_T | c | uc] buffer over RCCL once, before the timed region.
Inputs are resident in HBM when the timed region starts.  Weights and inputs are synthetic (deterministic
hash, commonscenes_amd/synth.py) -- no checkpoints or datasets are reachable offline.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the
dominant kernel (the 3x3x3 implicit-GEMM conv on fp32 MFMA) inside the timed region; `cpu_baseline` times
the CPU oracle (oracle/ref_torch.py, plain PyTorch fp32 -- a port, the reference itself cannot travel)
on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32-input MFMA peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
UNET_GFLOP_PER_SAMPLE = 557.9   # SURVEY App. A (conv3 471.3 + linear 62.0 + conv1 13.9 + attention 10.45 + norms)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--objects", type=int, default=32, help="objects per GPU (BASELINE metric: 32)")
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-objects", type=int, default=7,
                    help="objects in the bounded CPU-baseline sample (7 = the reference's sampler mini-batch, "
                         "sdfusion_txt2shape_model.py:493)")
    ap.add_argument("--math", choices=["fp32", "f16x3"], default=os.environ.get("CS_MATH", "f16x3"),
                    help="GEMM numerics: fp32-input MFMA, or fp32 carried as fp16 hi/lo pairs on the fp16 MFMA")
    ap.add_argument("--driver", choices=["python", "native"], default="python",
                    help="who sequences the UNet's kernels: commonscenes_amd/unet.py (default; carries the per-GEMM "
                         "HIP-event hooks the roofline block needs) or the native cs_unet_step driver")
    ap.add_argument("--gemm-table", action="store_true", help="print per-shape GEMM timings to stderr (debug)")
    ap.add_argument("--small", action="store_true", help="reduced-width UNet (debug only; result is not the metric)")
    return ap.parse_args()


def cpu_baseline(df, cfg, n_obj: int, objects_per_step: int):
    """Oracle (port of the reference's PyTorch path) on the host cores: one warm-up + one timed CFG DDIM step
    for n_obj objects, scaled to the 32-object step the metric is quoted on."""
    from commonscenes_amd import synth
    from oracle import ref_torch as R
    # oneDNN's conv3d stops scaling (and regresses badly) far below this box's core count at CFG batch 4:
    # 16 threads was the fastest setting measured on the 256-core host (profiles/r01_cpu_threads.txt:
    # 16 -> 0.42, 32 -> 0.43, 64 -> 0.75, 128 -> 1.77 s/sample)
    cores = int(os.environ.get("CS_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in df.state_dict().items()}
    sch = R.register_schedule(**R.DIFFUSION)
    x_T = synth.gaussian_like("bench:xT", (1, 3, 16, 16, 16)).repeat(n_obj, 1, 1, 1, 1)
    c = synth.gaussian_like("bench:cpu:c", (n_obj, 1, 1280))
    uc = synth.gaussian_like("bench:cpu:uc", (n_obj, 1, 1280))
    fn = lambda a, t, cc: R.unet_forward(sd, cfg, a, t, cc)
    with torch.no_grad():
        R.ddim_sample(fn, sch["alphas_cumprod"], 100, x_T, c, uc, 3.0, max_steps=1)       # warm-up
        t0 = time.perf_counter()
        R.ddim_sample(fn, sch["alphas_cumprod"], 100, x_T, c, uc, 3.0, max_steps=1)
        dt = time.perf_counter() - t0
    per_obj = dt / n_obj
    return dict(value=1.0 / (per_obj * objects_per_step), unit="DDIM steps/s (32 objects)", cores=cores,
                kind="port",
                sample=f"{n_obj} of {objects_per_step} objects (CFG batch {2 * n_obj}), 1 timed DDIM step after 1 "
                       f"warm-up = {dt:.2f} s, scaled x{objects_per_step / n_obj:g}; oracle/ref_torch.py on "
                       f"torch {torch.__version__} CPU fp32")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU fallback)"
    # test hooks (a 1-GPU box cannot host two RCCL ranks): CS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # CS_DIST_BACKEND=gloo swaps the process-group backend, so the multi-rank control flow can be exercised there
    if os.environ.get("CS_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("CS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from commonscenes_amd import ops, synth
    from commonscenes_amd.ddim import DDIMSampler
    from commonscenes_amd.scene import Sg2ScVAEModel, scene_param_shapes  # noqa: F401
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from oracle.ref_torch import DIFFUSION, UNET_FULL, UNET_SMALL, register_schedule

    cfg = dict(UNET_SMALL if a.small else UNET_FULL, dims=3, use_spatial_transformer=True)
    if a.driver == "native":
        from commonscenes_amd.unet_native import NativeDiffusionUNet
        df = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device=dev, math=a.math)
    else:
        df = DiffusionUNet(cfg, conditioning_key="crossattn", device=dev).set_math(a.math)
    df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device=str(dev)))
    sch = register_schedule(**DIFFUSION)

    class M:
        num_timesteps = 1000
        device = dev
        alphas_cumprod = sch["alphas_cumprod"]

        def apply_model(self, x, t, c):
            return df(x, t, c_crossattn=[c])

        def apply_model_cfg(self, x, t, c_in):
            return df.forward_cfg(x, t, c_in)

    # ---- conditioning: rank 0 runs the scene-graph GCN for all world*B objects, broadcast over RCCL ----
    B = a.objects
    total = B * world
    packed = torch.empty((total, 2 * 1280), dtype=torch.float32, device=dev)
    x_T = torch.empty((1, 3, 16, 16, 16), dtype=torch.float32, device=dev)
    t_cond0 = time.perf_counter()
    if rank == 0:
        from commonscenes_amd.scene import GraphTripleConvNet, _MLP
        g = synth.random_scene_graph(total, seed=111)
        ssd = synth.synth_state_dict(scene_param_shapes(35, 16), device=str(dev))
        ec, relmlp = GraphTripleConvNet(ssd, "gconv_net_ec_rel", 5), _MLP(ssd, "rel_mlp", 2, False)
        O, T = g["objs"].shape[0], g["triples"].shape[0]
        tri = g["triples"].to(dev)
        obj_vecs = torch.cat([g["text_feats"].to(dev), ops.embedding(ssd["obj_embeddings_dc.weight"], g["objs"].to(dev)),
                              g["z"].to(dev)], dim=1)
        pred_vecs = torch.cat([g["rel_feats"].to(dev),
                               ops.embedding(ssd["pred_embeddings_dc.weight"], tri[:, 1].contiguous())], dim=1)
        edges = torch.stack([tri[:, 0], tri[:, 2]], dim=1).contiguous()
        rel2, _ = ec(obj_vecs, pred_vecs, edges)
        packed[:, 1280:] = relmlp(rel2)[:total]          # c  (with GCN)
        packed[:, :1280] = relmlp(obj_vecs)[:total]      # uc (without)
        x_T.copy_(synth.gaussian_like("bench:xT", (1, 3, 16, 16, 16)))
    if world > 1:
        dist.broadcast(packed, src=0)
        dist.broadcast(x_T, src=0)
    torch.cuda.synchronize()
    cond_ms = (time.perf_counter() - t_cond0) * 1e3
    mine = packed[rank * B:(rank + 1) * B]
    uc = mine[:, :1280].reshape(B, 1, 1280).contiguous()
    c = mine[:, 1280:].reshape(B, 1, 1280).contiguous()
    c_in = torch.cat([uc, c])
    x = x_T.repeat(B, 1, 1, 1, 1).contiguous()

    sampler = DDIMSampler(M())
    sampler.make_schedule(a.ddim_steps, ddim_eta=0.0, verbose=False)
    ts = np.flip(sampler.ddim_timesteps)
    S = a.ddim_steps

    def run(n0, n):
        nonlocal x
        for i in range(n0, n0 + n):
            j = i % S
            x, _ = sampler._step(x, c_in, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, a.warmup)
    barrier()
    ops.GEMM_PROFILE = prof = []
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    ops.GEMM_PROFILE = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(x).all().item())

    if rank == 0:
        # dominant kernel = the tile instantiation with the most time (3x3x3 convs + the token GEMMs that share it):
        # every one of its launches counts, so the average matches rocprofv3's per-kernel average
        by_tile = {}
        for r in prof:
            by_tile[r["tile"]] = by_tile.get(r["tile"], 0.0) + r["e0"].elapsed_time(r["e1"])
        dom_tile = max(by_tile, key=by_tile.get) if by_tile else 0
        conv = [r for r in prof if r["tile"] == dom_tile] or prof
        conv_ms = sum(r["e0"].elapsed_time(r["e1"]) for r in conv)
        conv_fl = sum(r["flops"] for r in conv)
        all_ms = sum(r["e0"].elapsed_time(r["e1"]) for r in prof)
        all_fl = sum(r["flops"] for r in prof)
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None   # None: --driver native (no hooks)
        if a.gemm_table:
            agg = {}
            for r in prof:
                k = (r["taps"], r["m"], r["k"], r["n"], r["tile"])
                t = agg.setdefault(k, [0, 0.0, 0.0])
                t[0] += 1
                t[1] += r["e0"].elapsed_time(r["e1"])
                t[2] += r["flops"]
            print(f"{'taps':>4} {'M':>7} {'K':>6} {'N':>6} tile {'calls':>5} {'ms/step':>8} {'TF/s':>7}", file=sys.stderr)
            for k, t in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"{k[0]:4d} {k[1]:7d} {k[2]:6d} {k[3]:6d} {k[4]:4d} {t[0]:5d} {t[1] / a.steps:8.3f} "
                      f"{t[2] / t[1] / 1e9:7.1f}", file=sys.stderr)
        traffic = None
        tf = ROOT / "profiles" / ("r01_traffic_f16x3.json" if a.math == "f16x3" else "r01_traffic_fp32.json")
        if tf.exists() and B == 32 and not a.small:       # PMC passes cannot run inside this process: the figure is
            traffic = json.loads(tf.read_text())["hbm_bytes_per_launch"]   # the committed rocprofv3 --pmc result
        if a.math == "f16x3":
            # three fp16 MFMA passes per fp32-grade product: the pipe ceiling for ALGORITHMIC flops is 2.5 PF / 3
            shape = {1: "<2,2,2,2,false> (128x128", 2: "<1,7,4,1,false> (128x224", 3: "<1,1,2,2,false> (64x64",
                     4: "<1,7,8,1,false> (256x224"}.get(dom_tile, "(?")
            peak, kname = F16_MFMA_PEAK_TFLOPS / 3.0, (f"conv_gemm_f16x3_kernel{shape}-tile implicit "
                                                       "GEMM, 3x v_mfma_f32_32x32x16_f16 per K=16 on hi/lo splits)")
            dtype = "f32 (operands as fp16 hi+lo pairs, fp32 accumulate)"
        else:
            peak, kname = FP32_MFMA_PEAK_TFLOPS, ("conv_gemm_f32_kernel<1,7,4,1> (128x224-tile implicit GEMM, "
                                                  "v_mfma_f32_32x32x2_f32)")
            dtype = "f32"
        res = {
            "metric": "DDIM denoise steps/sec (32 objects, 16^3 latent)",
            "value": world * a.steps / dt,
            "unit": "DDIM steps/s (32-object CFG step: UNet fwd @ batch 64 + update)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "v2_full shape branch (BASELINE configs[2]): 32 objects/GPU, CFG scale 3.0, "
                                   f"{S}-step DDIM schedule, 3x16^3 latents, UNet "
                                   f"{df.num_parameters() / 1e6:.1f}M params fp32, 1 context token",
                       "objects_per_gpu": B, "unet_batch": 2 * B, "ddim_steps": S,
                       "parallelism": f"object-sharded x{world} (replicated weights, no per-step collective)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak if achieved is not None else None, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch of the dominant kernel, FETCH_SIZE x2 (gfx950) + WRITE_SIZE, "
                                         "from profiles/r01_traffic_*.json (separate rocprofv3 --pmc passes of this command)",
                         "kernel": kname, "math": a.math,
                         "launches": len(conv), "avg_launch_ms": conv_ms / max(len(conv), 1),
                         "algorithmic_gflop_per_launch": conv_fl / max(len(conv), 1) / 1e9,
                         "share_of_step_time": conv_ms / (dt * 1e3),
                         "all_gemm_tflops": all_fl / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0,
                         "all_gemm_share_of_step_time": all_ms / (dt * 1e3),
                         "whole_step_tflops": (2 * B * UNET_GFLOP_PER_SAMPLE * 1e9 * a.steps / dt / 1e12)
                         if not a.small else None},
            "conditioning_ms": cond_ms, "finite": finite, "unet_driver": a.driver,
        }
        if not a.no_cpu_baseline and not a.small:
            res["cpu_baseline"] = cpu_baseline(df, cfg, a.cpu_objects, B)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

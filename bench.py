#!/usr/bin/env python
"""Benchmark: DDIM denoise steps/sec of the CommonScenes shape branch on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one classifier-free-guided DDIM step over this rank's 32 objects: a UNet forward at batch 64
(413.5 M parameters, 3x16^3 latents, one 1280-d context token per sample) + the fused CFG/DDIM update
(BASELINE.json configs[2], SURVEY 8d "C3").  Objects are independent, so N GPUs hold N x 32 objects (weak scaling, no
per-step collective).  The multi-rank data path is the product's own (commonscenes_amd/dist.py): rank 0 runs the
scene-graph GCN for all N x 32 objects, ONE broadcast of the packed [x_T | uc | c] buffer, every rank takes its
contiguous shard, and -- outside the timed step loop, reported separately as `end_to_end` -- every rank decodes its
latents to 64^3 SDFs and ONE all-gather returns all of them to every rank.
Inputs are resident in HBM when the timed region starts.  Weights and inputs are synthetic (deterministic hash,
commonscenes_amd/synth.py) -- no checkpoints or datasets are reachable offline.

Prints ONE JSON line (rank 0):
  value / ms_per_step   the metric: K timed steps between barrier + synchronize pairs, max over ranks;
  roofline              the dominant kernel, HIP events around every one of its launches inside the timed region;
  decode                VQ-VAE decode of the rank's 32 latents (quantise + Decoder3D, 723 GFLOP/object), own roofline block;
  end_to_end            steps/s with decode + all-gather amortised over the S-step run (what a whole sample() costs);
  c2                    BASELINE configs[1]: ONE object (N=1 only): ms/step and steps/s;
  mesh                  sdf_to_mesh (HIP marching cubes) of 32 analytic 64^3 SDFs (N=1 only);
  cpu_baseline          the CPU oracle (oracle/ref_torch.py -- a port; the reference cannot travel) on a bounded sample.
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
# kernel instantiation per (tile code, slab width, pre-split activations): conv_gemm_f16x3_kernel<WMB, WNB, WAVES_M, WAVES_N,
# PRE, SLAB>.  SLAB 0 = per-tap A gather; 32 / 64 = A operand of 3x3x3 stride-1 convs staged as one slab per (kd, channel
# chunk); PRE = the GroupNorm producer already emitted the fp16 hi / lo operand pair (no conversion in the K loop)
TILE_SHAPES = {1: ("2,2,2,2", "128x128"), 2: ("1,7,4,1", "128x224"), 3: ("1,1,2,2", "64x64"), 4: ("1,7,8,1", "256x224"),
               6: ("1,4,8,1", "256x128"), 7: ("1,2,8,1", "256x64")}


def kernel_label(key):
    tile, slab, pre = key
    if tile == 5:
        return "pw_gemm_f16x3_kernel (persistent ping-pong 2x128x224"
    wv, shape = TILE_SHAPES.get(tile, ("?", "?"))
    return (f"conv_gemm_f16x3_kernel<{wv},{'true' if pre else 'false'},{slab}> ({shape}"
            f"{', A slab' if slab else ''}{', pre-split activations' if pre else ''}")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--objects", type=int, default=32, help="objects per GPU (BASELINE metric: 32)")
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the decode / end-to-end / C2 blocks (profiling runs)")
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
    for n_obj objects, scaled to the 32-object step the metric is quoted on.  The only use of oracle/ in this file."""
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


def gemm_summary(prof, wall_ms, math):
    """dominant tile instantiation of a HIP-event profile: achieved TF/s, both roofline conventions."""
    by_tile = {}
    for r in prof:
        k = (r["tile"], r.get("slab", 0), bool(r.get("pre", False)))
        by_tile[k] = by_tile.get(k, 0.0) + r["e0"].elapsed_time(r["e1"])
    if not by_tile:
        return None
    dom = max(by_tile, key=by_tile.get)
    sel = [r for r in prof if (r["tile"], r.get("slab", 0), bool(r.get("pre", False))) == dom]
    ms = sum(r["e0"].elapsed_time(r["e1"]) for r in sel)
    fl = sum(r["flops"] for r in sel)
    all_ms = sum(r["e0"].elapsed_time(r["e1"]) for r in prof)
    all_fl = sum(r["flops"] for r in prof)
    achieved = fl / (ms * 1e-3) / 1e12
    if math == "f16x3":
        peak = F16_MFMA_PEAK_TFLOPS / 3.0
        kname = f"{kernel_label(dom)}; implicit GEMM, 3x v_mfma_f32_32x32x16_f16 per K=16 on hi/lo splits)"
        peak_note = ("peak = dense fp16 MFMA peak (2500 TF/s) / 3: every fp32-grade product costs three fp16 MFMA "
                     "passes, so 833 TF/s of ALGORITHMIC flops saturates the matrix pipe; frac_of_f16_dense_peak "
                     "prices the same algorithmic flops against the raw 2500 TF/s, frac_of_fp32_matrix_peak against "
                     "the 157.3 TF/s fp32-input MFMA the reference's dtype maps to")
    else:
        peak = FP32_MFMA_PEAK_TFLOPS
        kname = "conv_gemm_f32_kernel<1,7,4,1> (128x224-tile implicit GEMM, v_mfma_f32_32x32x2_f32)"
        peak_note = "peak = dense fp32-input MFMA peak"
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "peak_note": peak_note,
            "frac_of_f16_dense_peak": achieved / F16_MFMA_PEAK_TFLOPS if math == "f16x3" else None,
            "frac_of_fp32_matrix_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
            "issued_mfma_tflops": 3.0 * achieved if math == "f16x3" else achieved,
            "kernel": kname, "math": math, "launches": len(sel), "avg_launch_ms": ms / len(sel),
            "algorithmic_gflop_per_launch": fl / len(sel) / 1e9, "share_of_wall_time": ms / wall_ms,
            "all_gemm_tflops": all_fl / (all_ms * 1e-3) / 1e12, "all_gemm_share_of_wall_time": all_ms / wall_ms}


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

    from commonscenes_amd import configs as K
    from commonscenes_amd import dist as D
    from commonscenes_amd import ops, synth
    from commonscenes_amd.ddim import DDIMSampler
    from commonscenes_amd.scene import scene_param_shapes
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes

    cfg = K.reduced(K.UNET_CROSSATTN) if a.small else dict(K.UNET_CROSSATTN)
    if a.driver == "native":
        from commonscenes_amd.unet_native import NativeDiffusionUNet
        df = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device=dev, math=a.math)
    else:
        df = DiffusionUNet(cfg, conditioning_key="crossattn", device=dev).set_math(a.math)
    df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device=str(dev)))
    model = K.ScheduleModel(df, dev)

    # ---- conditioning: rank 0 runs the scene-graph GCN for all world*B objects, ONE broadcast (dist.py) ----
    B = a.objects
    total = B * world
    t_cond0 = time.perf_counter()
    x_T = uc_all = c_all = None
    if rank == 0:
        from commonscenes_amd.scene import GraphTripleConvNet, _MLP
        g = synth.random_scene_graph(total, seed=111)
        ssd = synth.synth_state_dict(scene_param_shapes(35, 16), device=str(dev))
        ec, relmlp = GraphTripleConvNet(ssd, "gconv_net_ec_rel", 5), _MLP(ssd, "rel_mlp", 2, False)
        tri = g["triples"].to(dev)
        obj_vecs = torch.cat([g["text_feats"].to(dev), ops.embedding(ssd["obj_embeddings_dc.weight"], g["objs"].to(dev)),
                              g["z"].to(dev)], dim=1)
        pred_vecs = torch.cat([g["rel_feats"].to(dev),
                               ops.embedding(ssd["pred_embeddings_dc.weight"], tri[:, 1].contiguous())], dim=1)
        edges = torch.stack([tri[:, 0], tri[:, 2]], dim=1).contiguous()
        rel2, _ = ec(obj_vecs, pred_vecs, edges)
        c_all = relmlp(rel2)[:total].reshape(total, 1, 1280)          # with the GCN
        uc_all = relmlp(obj_vecs)[:total].reshape(total, 1, 1280)     # without
        x_T = synth.gaussian_like("bench:xT", (1, 3, 16, 16, 16)).to(dev)
    x_T, uc_all, c_all = D.broadcast_conditioning(x_T, uc_all, c_all, total, dev, src=0)
    torch.cuda.synchronize()
    cond_ms = (time.perf_counter() - t_cond0) * 1e3
    lo, hi = D.shard_range(total, world, rank)
    uc = uc_all[lo:hi].contiguous()
    c = c_all[lo:hi].contiguous()
    c_in = torch.cat([uc, c])
    x = x_T.repeat(B, 1, 1, 1, 1).contiguous()

    sampler = DDIMSampler(model)
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

    def max_over_ranks(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    run(0, a.warmup)
    barrier()
    ops.GEMM_PROFILE = prof = []
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    ops.GEMM_PROFILE = None
    dt = max_over_ranks(dt)
    finite = bool(torch.isfinite(x).all().item())
    overflow = ops.read_status(dev) != 0

    # ---- extras, outside the timed region: decode of this rank's latents, the all-gather, one object (C2) ----
    decode = e2e = c2 = mesh = None
    if not a.no_extras and not a.small:
        vq = VQVAE(K.VQVAE_DDCONFIG, K.VQVAE_N_EMBED, K.VQVAE_EMBED_DIM, device=dev).set_math(a.math)
        vq.load_state_dict(synth.synth_state_dict(vqvae_param_shapes(K.VQVAE_DDCONFIG, K.VQVAE_N_EMBED,
                                                                     K.VQVAE_EMBED_DIM), device=str(dev)))
        lat = synth.gaussian_like("bench:lat", (B, 3, 16, 16, 16), scale=0.8).to(dev)   # code-book-scale latents
        sdf = vq.decode_no_quant(lat)                                                   # warm-up (packs the weights)
        barrier()
        ops.GEMM_PROFILE = dprof = []
        t0 = time.perf_counter()
        sdf = vq.decode_no_quant(lat)
        barrier()
        dec_s = max_over_ranks(time.perf_counter() - t0)
        ops.GEMM_PROFILE = None
        t0 = time.perf_counter()
        allsdf = D.all_gather_objects(sdf, total)
        barrier()
        gat_s = max_over_ranks(time.perf_counter() - t0)
        dec_tf = B * K.VQ_DECODE_GFLOP_PER_OBJECT / dec_s / 1e3
        if rank == 0:
            decode = {"objects": B, "ms": dec_s * 1e3, "ms_per_object": dec_s * 1e3 / B,
                      "algorithmic_gflop_per_object": K.VQ_DECODE_GFLOP_PER_OBJECT, "whole_decode_tflops": dec_tf,
                      "executed_gflop_per_object": (K.VQ_DECODE_GFLOP_EXECUTED_PER_OBJECT if ops.FOLD_UPSAMPLE
                                                    else K.VQ_DECODE_GFLOP_PER_OBJECT),
                      "flop_note": "whole_decode_tflops prices the reference's direct-form work; the two Upsample convs "
                                   "run folded onto the source grid (8/27 of their multiply-adds, cs_conv_gemm_up2)",
                      "whole_decode_frac_of_peak": dec_tf / (F16_MFMA_PEAK_TFLOPS / 3.0 if a.math == "f16x3"
                                                             else FP32_MFMA_PEAK_TFLOPS),
                      "finite": bool(torch.isfinite(allsdf).all().item()), "gathered_shape": list(allsdf.shape),
                      "roofline": gemm_summary(dprof, dec_s * 1e3, a.math)}
            step_s = dt / a.steps
            e2e = {"all_gather_ms": gat_s * 1e3, "all_gather_bytes_per_rank": int(sdf.numel() * 4),
                   "ddim_steps": S,
                   "value": world / (step_s + (dec_s + gat_s) / S),
                   "unit": "DDIM steps/s with decode + all-gather amortised over the S-step run",
                   "whole_run_s": S * step_s + dec_s + gat_s}
        del allsdf, sdf
        if world == 1:
            # the step right after the path: sdf_to_mesh of the 32 objects (render_all=True, as
            # helpers/util.py:298 calls it) on analytic SDFs -- spheres of assorted radii; the synthetic decoder's output
            # is noise-like and would mesh to an unrepresentative number of triangles
            from commonscenes_amd.mesh import sdf_to_mesh
            g = torch.arange(64, dtype=torch.float32, device=dev)
            gx, gy, gz = torch.meshgrid(g, g, g, indexing="ij")
            rad = torch.linspace(8.0, 26.0, B, device=dev).view(B, 1, 1, 1)
            vol = (torch.sqrt((gx - 31.3) ** 2 + (gy - 32.1) ** 2 + (gz - 30.7) ** 2).unsqueeze(0) - rad).unsqueeze(1) / 64.0
            m0 = sdf_to_mesh(vol, level=0.02, render_all=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m0 = sdf_to_mesh(vol, level=0.02, render_all=True)
            torch.cuda.synchronize()
            msec = time.perf_counter() - t0
            nv = sum(int(v.shape[0]) for v in m0.verts_list())
            nf = sum(int(f.shape[0]) for f in m0.faces_list())
            mbytes = B * 64 ** 3 * (3 * 4 + 8) + nv * 12 + nf * 24     # 3 volume reads + voxel word write/read + outputs
            mesh = {"objects": B, "ms": msec * 1e3, "vertices": nv, "triangles": nf,
                    "roofline": {"bound": "hbm", "achieved": mbytes / msec / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": mbytes / msec / 1e9 / 8000.0,
                                 "note": "algorithmic bytes / wall time of the whole call, including the one host "
                                         "read-back of the per-block totals that sizes the outputs"}}
        if world == 1:
            # BASELINE configs[1] (C2): ONE object; same sampler, CFG batch 2
            x1 = x_T.clone()
            c1 = torch.cat([uc[:1], c[:1]])
            df.reset_run_cache() if hasattr(df, "reset_run_cache") else None
            for j in range(3):
                x1, _ = sampler._step(x1, c1, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            n1 = 20
            t0 = time.perf_counter()
            for j in range(3, 3 + n1):
                x1, _ = sampler._step(x1, c1, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t0) / n1
            c2 = {"workload": "BASELINE configs[1]: 1 object, CFG batch 2", "ms_per_step": d1 * 1e3,
                  "steps_per_s": 1.0 / d1, "whole_step_tflops": 2 * K.UNET_GFLOP_PER_SAMPLE / d1 / 1e3,
                  "frac_of_peak": 2 * K.UNET_GFLOP_PER_SAMPLE / d1 / 1e3 / (F16_MFMA_PEAK_TFLOPS / 3.0 if a.math == "f16x3"
                                                                            else FP32_MFMA_PEAK_TFLOPS)}

    if rank == 0:
        wall_ms = dt * 1e3
        roof = gemm_summary(prof, wall_ms, a.math) or {"bound": "mfma", "achieved": None, "peak": None,
                                                       "unit": "TFLOP/s", "frac": None}   # --driver native: no hooks
        if a.gemm_table:
            agg = {}
            for r in prof:
                k = (r["taps"], r["m"], r["k"], r["n"], r["tile"] + (100 if r.get("slab") else 0))
                t = agg.setdefault(k, [0, 0.0, 0.0])
                t[0] += 1
                t[1] += r["e0"].elapsed_time(r["e1"])
                t[2] += r["flops"]
            print(f"{'taps':>4} {'M':>7} {'K':>6} {'N':>6} tile {'calls':>5} {'ms/step':>8} {'TF/s':>7}", file=sys.stderr)
            for k, t in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"{k[0]:4d} {k[1]:7d} {k[2]:6d} {k[3]:6d} {k[4]:4d} {t[0]:5d} {t[1] / a.steps:8.3f} "
                      f"{t[2] / t[1] / 1e9:7.1f}", file=sys.stderr)
        traffic = None
        for cand in ("r02_traffic_f16x3.json", "r01_traffic_f16x3.json") if a.math == "f16x3" else ("r01_traffic_fp32.json",):
            tf = ROOT / "profiles" / cand
            if tf.exists() and B == 32 and not a.small:   # PMC passes cannot run inside this process: the figure is
                traffic = json.loads(tf.read_text())["hbm_bytes_per_launch"]   # the committed rocprofv3 --pmc result
                roof["traffic_source"] = f"profiles/{cand}"
                break
        roof["traffic"] = traffic
        roof["traffic_note"] = ("HBM bytes per launch of the dominant kernel, FETCH_SIZE x2 (gfx950) + WRITE_SIZE, from "
                                "separate rocprofv3 --pmc passes of this command (tools/pmc_traffic.sh)")
        roof["whole_step_tflops"] = (2 * B * K.UNET_GFLOP_PER_SAMPLE * 1e9 * a.steps / dt / 1e12) if not a.small else None
        ex = K.UNET_GFLOP_EXECUTED_PER_SAMPLE if ops.FOLD_UPSAMPLE else K.UNET_GFLOP_PER_SAMPLE
        roof["whole_step_executed_tflops"] = (2 * B * ex * 1e9 * a.steps / dt / 1e12) if not a.small else None
        roof["whole_step_note"] = ("whole_step_tflops prices the reference's direct-form work (557.9 GFLOP per UNet sample); "
                                   "the two Upsample convs run folded onto the source grid (12/27 of their multiply-adds, "
                                   "cs_conv_gemm_up2), so the issued algorithmic work is whole_step_executed_tflops")
        res = {
            "metric": "DDIM denoise steps/sec (32 objects, 16^3 latent)",
            "value": world * a.steps / dt,
            "unit": "DDIM steps/s (32-object CFG step: UNet fwd @ batch 64 + update)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (operands as fp16 hi+lo pairs, fp32 accumulate)" if a.math == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": "v2_full shape branch (BASELINE configs[2]): 32 objects/GPU, CFG scale 3.0, "
                                   f"{S}-step DDIM schedule, 3x16^3 latents, UNet "
                                   f"{df.num_parameters() / 1e6:.1f}M params fp32, 1 context token",
                       "objects_per_gpu": B, "unet_batch": 2 * B, "ddim_steps": S,
                       "parallelism": f"object-sharded x{world} (replicated weights, no per-step collective; "
                                      "1 broadcast in, 1 all-gather out)"},
            "roofline": roof, "decode": decode, "end_to_end": e2e, "c2": c2, "mesh": mesh,
            "conditioning_ms": cond_ms, "finite": finite, "f16x3_overflow": overflow, "unet_driver": a.driver,
        }
        if not a.no_cpu_baseline and not a.small:
            res["cpu_baseline"] = cpu_baseline(df, cfg, a.cpu_objects, B)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark: DDIM denoise steps/sec of the CommonScenes shape branch on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: bench.py re-executes itself under
                                                              torch.distributed.run -- N ranks, one GPU each, RCCL)
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
  native_driver         the same 32-object and 1-object steps through the native whole-forward driver cs_unet_step;
  c7 / c7x5             the reference's sampler mini-batch of 7 (ms per object-step), and the DEFAULT product call
                        SDFusionText2ShapeModel.rel2shape on 32 objects (mini_B = 7 slices coalesced into one launch
                        batch, r4) next to the reference's own five sequential mini-batches (launch_B = 0);
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
               6: ("1,4,8,1", "256x128"), 7: ("1,2,8,1", "256x64"), 8: ("2,2,8,1", "512x64"), 9: ("2,4,8,1", "512x128")}


def kernel_label(key):
    tile, slab, pre = key[:3]
    wv, shape = TILE_SHAPES.get(tile, ("?", "?"))
    pair = len(key) > 3 and key[3]
    wino = len(key) > 4 and key[4]
    return (f"conv_gemm_f16x3_kernel<{wv},{'true' if pre else 'false'},{slab}{',false,3' if wino else ''}> ({shape}"
            f"{', A slab' if slab else ''}{', pre-split activations' if pre else ''}"
            f"{', interleaved operand pair' if pair else ''}"
            f"{', the Winograd-W position GEMMs of a 3x3x3 conv (3x3x1 taps; six for F(4,3), four for F(2,3)) in one launch' if wino else ''}")


def rocprof_name(key):
    """prefix of the instantiation's name as rocprofv3 prints it: conv_gemm_f16x3_kernel<WMB, WNB, WAVES_M, WAVES_N, PRE,
    SLAB, PAIR, ...> -- further template arguments (taps per kd, pointwise) follow; the prefix identifies the launches"""
    tile, slab, pre = key[:3]
    pair = bool(key[3]) if len(key) > 3 else False
    wv = TILE_SHAPES.get(tile, ("?",))[0].replace(",", ", ")
    wino = bool(key[4]) if len(key) > 4 else False
    # (taps per kd follow: 3 = the Winograd-W position GEMMs; the 27-tap slab kernel and every gather kernel print 9)
    return (f"conv_gemm_f16x3_kernel<{wv}, {'true' if pre else 'false'}, {slab}, {'true' if pair else 'false'},"
            + (" 3," if wino else " 9," if slab and pre else ""))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--objects", type=int, default=32, help="objects per GPU (BASELINE metric: 32)")
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the decode / end-to-end / C2 blocks (profiling runs)")
    ap.add_argument("--cpu-baseline", choices=["full", "quick"], default=os.environ.get("CS_CPU_BASELINE", "full"),
                    help="full = BASELINE.md section 4 (B=1 x 5 steps, B=32 x 2 steps as mini-batches of 7 and as one "
                         "batch; ~1-2 min of host time); quick = one mini-batch of 7, one step, scaled")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-input-MFMA comparison steps")
    ap.add_argument("--no-gemm-profile", action="store_true",
                    help="no per-GEMM HIP-event hooks in the timed region (A/B runs at small batches: two event records per "
                         "GEMM launch cost ~1.3 ms of a 6.4 ms one-object step); the roofline block is then null")
    ap.add_argument("--traffic", action="store_true",
                    help="(default at one GPU unless --no-extras / --no-traffic) measure the dominant kernel's HBM bytes per "
                         "launch now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run of this script "
                         "(tools/pmc_traffic.sh); each pass is bounded by a timeout and a failure leaves the field null")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes: roofline.traffic = null")
    ap.add_argument("--math", choices=["fp32", "f16x3"], default=os.environ.get("CS_MATH", "f16x3"),
                    help="GEMM numerics: fp32-input MFMA, or fp32 carried as fp16 hi/lo pairs on the fp16 MFMA")
    ap.add_argument("--driver", choices=["python", "native"], default="python",
                    help="who sequences the UNet's kernels: commonscenes_amd/unet.py (default; carries the per-GEMM "
                         "HIP-event hooks the roofline block needs) or the native cs_unet_step driver")
    ap.add_argument("--gemm-table", action="store_true", help="print per-shape GEMM timings to stderr (debug)")
    ap.add_argument("--small", action="store_true", help="reduced-width UNet (debug only; result is not the metric)")
    return ap.parse_args()


def cpu_baseline(df, cfg, objects_per_step: int, quick: bool = False):
    """The CPU oracle (oracle/ref_torch.py: a PORT of the reference's PyTorch path, pinned on the reference's goldens;
    the reference itself cannot travel) on this box's host cores, following BASELINE.md section 4:
      B = 1  : 5 full CFG DDIM steps (after 1 warm-up)                                  -> `b1`
      B = 32 : 2 steps on the reference's schedule of sampler mini-batches of 7 objects  -> `b32_minibatch7`
               (sdfusion_txt2shape_model.py:493-511: 7+7+7+7+4, CFG batch 14 / 8)
      B = 32 : 2 steps as ONE batch (CFG batch 64)                                       -> `b32_one_batch`
    `value` = the better of the two B = 32 rates (DDIM steps/s for 32 objects).  `quick` (CS_CPU_BASELINE=quick, or
    --cpu-baseline quick) times one mini-batch of 7 for one step and scales.  The only use of oracle/ in this file."""
    from commonscenes_amd import synth
    from oracle import ref_torch as R
    host_cores = os.cpu_count() or 1
    # oneDNN's conv3d stops scaling (and regresses) far below this box's core count: r5 swept BOTH legs at the batch sizes
    # they run (profiles/r05_cpu_threads.txt, one CFG step on the 256-core host: one batch of 32 objects 30.7 / 29.2 / 36.7 /
    # 55.5 / 159.6 s at 16 / 32 / 64 / 128 / 256 threads, a mini-batch of 7 5.65 / 4.79 / 7.55 / 15.6 / 117.8 s) -- 32
    # threads is the fastest setting for both, i.e. the strongest CPU baseline; "all host cores" 
    # would be 5x (one batch) to 25x (mini-batches) slower.  The one-batch leg also tries 64 threads and keeps the faster.
    cores = int(os.environ.get("CS_CPU_THREADS", min(host_cores, 32)))
    sd = {k: v.detach().cpu() for k, v in df.state_dict().items()}
    sch = R.register_schedule(**R.DIFFUSION)
    fn = lambda a, t, cc: R.unet_forward(sd, cfg, a, t, cc)

    def steps(n_obj, n_steps, tag):
        x_T = synth.gaussian_like("bench:xT", (1, 3, 16, 16, 16)).repeat(n_obj, 1, 1, 1, 1)
        c = synth.gaussian_like(f"bench:cpu:c{tag}", (n_obj, 1, 1280))
        uc = synth.gaussian_like(f"bench:cpu:uc{tag}", (n_obj, 1, 1280))
        t0 = time.perf_counter()
        with torch.no_grad():
            R.ddim_sample(fn, sch["alphas_cumprod"], 100, x_T, c, uc, 3.0, max_steps=n_steps)
        return time.perf_counter() - t0

    torch.set_num_threads(cores)
    out = dict(unit="DDIM steps/s (32 objects)", kind="port", cores=cores, host_cores=host_cores,
               torch=torch.__version__, oracle="oracle/ref_torch.py (CPU fp32, pinned on the reference's goldens)",
               cores_note=(f"BASELINE.md section 4 says 'all host cores'; {cores} of {host_cores} are used because that is "
                           "the FASTEST setting on this host for both legs, i.e. the strongest CPU baseline "
                           "(profiles/r05_cpu_threads.txt, one CFG step: 32 objects as one batch 30.7 / 29.2 / 36.7 / 55.5 / "
                           "159.6 s and a mini-batch of 7 5.65 / 4.79 / 7.55 / 15.6 / 117.8 s at 16 / 32 / 64 / 128 / 256 "
                           "threads); the one-batch leg also tries 64 threads and keeps the faster; CS_CPU_THREADS overrides"))
    if quick:
        steps(7, 1, "w")
        dt = steps(7, 1, "q")
        out.update(value=1.0 / (dt / 7 * objects_per_step),
                   sample=f"quick: 7 of {objects_per_step} objects (CFG batch 14), 1 timed DDIM step after 1 warm-up = "
                          f"{dt:.2f} s, scaled x{objects_per_step / 7:g}")
        return out
    steps(1, 1, "w1")
    t1 = steps(1, 5, "b1")
    out["b1"] = dict(steps=5, seconds=t1, steps_per_s=5.0 / t1, cores=cores,
                     workload="BASELINE configs[1] shape: 1 object, CFG batch 2")
    # the reference's own schedule at 32 objects: mini-batches of 7 (each runs its 2 steps before the next starts)
    sizes = [min(7, objects_per_step - i) for i in range(0, objects_per_step, 7)]
    steps(7, 1, "w7")
    t7 = sum(steps(n, 2, f"mb{i}") for i, n in enumerate(sizes))
    out["b32_minibatch7"] = dict(steps=2, seconds=t7, steps_per_s=2.0 / t7, cores=cores,
                                 workload=f"{objects_per_step} objects as sampler mini-batches {sizes} "
                                          "(sdfusion_txt2shape_model.py:493-511)")
    best = None
    for thr in sorted({cores, min(host_cores, 64)}):
        torch.set_num_threads(thr)
        tb = steps(objects_per_step, 1, f"one{thr}")
        if best is None or tb < best[0]:
            best = (tb, thr)
    torch.set_num_threads(best[1])
    tb2 = steps(objects_per_step, 2, "one")
    out["b32_one_batch"] = dict(steps=2, seconds=tb2, steps_per_s=2.0 / tb2, cores=best[1],
                                workload=f"{objects_per_step} objects as one batch (CFG batch {2 * objects_per_step}); "
                                         f"thread count = the faster of {sorted({cores, min(host_cores, 64)})} on one step")
    v7, v1 = out["b32_minibatch7"]["steps_per_s"], out["b32_one_batch"]["steps_per_s"]
    out["value"] = max(v7, v1)
    out["cores"] = best[1] if v1 >= v7 else cores
    out["sample"] = (f"BASELINE.md section 4: B=1 x 5 steps = {t1:.1f} s; B={objects_per_step} x 2 steps as mini-batches "
                     f"of 7 = {t7:.1f} s ({v7:.4f} steps/s) and as one batch = {tb2:.1f} s ({v1:.4f} steps/s); value = "
                     f"the better one; {host_cores} host cores, {out['cores']} threads used")
    return out


def measure_traffic(ksub: str, a):
    """HBM bytes per launch of kernel `ksub`, as MI355X_MICROARCH.md prescribes: two separate rocprofv3 passes
    (--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, nothing else) around a short child run of this script; FETCH_SIZE is
    calibrated (x2 on gfx950) in the same pass on ln_kernel<2>, which reads a known 65536 x 448 fp32 = 112 MiB."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    res = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", f"{td}/{c}", "-o", "p", "--",
                   sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-extras", "--no-fp32-leg", "--objects", str(a.objects), "--math", a.math]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                                   timeout=240)
            except (subprocess.TimeoutExpired, OSError):
                return None
            files = glob.glob(f"{td}/{c}/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                return None
            k_sum = k_n = l_sum = l_n = 0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != c:
                    continue
                v = float(row["Counter_Value"])
                if ksub in row["Kernel_Name"]:
                    k_sum += v
                    k_n += 1
                if "ln_kernel<2>" in row["Kernel_Name"]:
                    l_sum += v
                    l_n += 1
            res[c] = dict(launches=k_n, per_launch_bytes=k_sum * 1024 / max(k_n, 1),
                          ln2_per_launch_mib=l_sum / max(l_n, 1) / 1024)
    cal = res["FETCH_SIZE"]["ln2_per_launch_mib"]
    corr = 112.0 / cal if cal else 2.0
    fb, wb = res["FETCH_SIZE"]["per_launch_bytes"] * corr, res["WRITE_SIZE"]["per_launch_bytes"]
    return dict(kernel=ksub, hbm_bytes_per_launch=fb + wb, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                fetch_correction=corr, raw=res,
                note=f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes around "
                     f"`bench.py --steps 2 --warmup 1`; FETCH_SIZE x{corr:.3f} (calibrated on ln_kernel<2>'s 112 MiB read)")


def measure_mfma_busy(ksubs, a):
    """Matrix-pipe occupancy and effective clock of the kernels whose names contain one of `ksubs`, from ONE more
    rocprofv3 pass (--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE, nothing else) around a
    short child run of this script (the power-ceiling argument must rest on this round's binaries).
      effective clock = GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration in the same pass
      mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (MI355X_MICROARCH.md: the counter
                        counts cycles, 32 per 32x32x16 MFMA)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    ctrs = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", f"{td}/m", "-o", "p", "--",
               sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
               "--no-extras", "--no-fp32-leg", "--objects", str(a.objects), "--math", a.math]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=300)
        except (subprocess.TimeoutExpired, OSError):
            return None
        files = glob.glob(f"{td}/m/**/*counter_collection.csv", recursive=True)
        if r.returncode != 0 or not files:
            return None
        dur = {}
        for f in glob.glob(f"{td}/m/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                try:
                    dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
        per = {}                                         # (ksub, dispatch) -> counters
        for row in csv.DictReader(open(files[0])):
            ks = next((k for k in ksubs if k in row["Kernel_Name"]), None)
            if ks is None or row["Counter_Name"] not in ctrs:
                continue
            d = per.setdefault((ks, row["Dispatch_Id"]), {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                try:
                    d["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                except ValueError:
                    pass
    out = {}
    for ks in ksubs:
        rows = [dict(v, ns=v.get("ns", dur.get(did))) for (k, did), v in per.items() if k == ks]
        rows = [v for v in rows if all(c in v for c in ctrs) and v["GRBM_GUI_ACTIVE"] > 0]
        if not rows:
            continue
        gui = sum(v["GRBM_GUI_ACTIVE"] for v in rows) / 8.0
        busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in rows)
        ns = sum(v["ns"] for v in rows if v.get("ns"))
        out[ks] = dict(launches=len(rows), mfma_busy_frac=busy / (1024.0 * gui),
                       effective_clock_ghz=(gui / ns) if ns else None,
                       mfma_insts_per_launch=sum(v["SQ_INSTS_MFMA"] for v in rows) / len(rows),
                       avg_launch_us_under_pmc=(ns / len(rows) / 1e3) if ns else None)
    return out or None


def gemm_summary(prof, wall_ms, math):
    """dominant tile instantiation of a HIP-event profile: achieved TF/s, both roofline conventions."""
    by_tile = {}
    keyf = lambda r: (r["tile"], r.get("slab", 0), bool(r.get("pre", False)), bool(r.get("pair", False)),
                      bool(r.get("wino", False)))
    for r in prof:
        k = keyf(r)
        by_tile[k] = by_tile.get(k, 0.0) + r["e0"].elapsed_time(r["e1"])
    if not by_tile:
        return None
    dom = max(by_tile, key=by_tile.get)
    sel = [r for r in prof if keyf(r) == dom]
    ms = sum(r["e0"].elapsed_time(r["e1"]) for r in sel)
    fl = sum(r["flops"] for r in sel)
    # (a Winograd-W record carries a third event: its output-transform launch, an HBM-bound pass, is counted into the GEMM
    # totals' time but is not a launch of the dominant KERNEL)
    xform_ms = sum(r["e1"].elapsed_time(r["e2"]) for r in prof if "e2" in r)
    all_ms = sum(r["e0"].elapsed_time(r["e1"]) for r in prof) + xform_ms
    all_fl = sum(r["flops"] for r in prof)
    all_fl_direct = sum(r.get("flops_direct", r["flops"]) for r in prof)
    achieved = fl / (ms * 1e-3) / 1e12
    wino_dom = bool(dom[4])
    # ALGORITHMIC HBM bytes of the dominant kernel's launches (what `traffic`, the counted bytes, is to be ratioed against):
    # the A operand once (m x cin x 4 B: fp32, or the fp16 hi + lo pair), the packed weights once (k x n x 4 B: fp16 hi +
    # lo), the fp32 result once (m x n x 4 B) and the residual where the launch adds one -- per launch, averaged over the
    # kernel's launches.  The counters come out ~1.9x above this at 32 objects: a slab conv re-reads its A rows once per
    # 224-column tile of the output and per kd (the halo planes), both of which hit L2 / MALL rather than HBM only partly.
    ab = 0.0
    for r in sel:
        cin = r["k"] / max(1, r["taps"])
        if r.get("wino"):
            # the position launch: the transformed operand pair [npos][M/variant][cin] once, npos packed weight pairs, its
            # fp32 results [slices][npos][M/variant][n] (the epilogue terms belong to the output-transform launch)
            ab += r["m"] * cin * 4.0 + r.get("npos", 4) * r["k"] * r["n"] * 4.0 + r.get("slices", 1) * r["m"] * r["n"] * 4.0
        else:
            ab += r["m"] * cin * 4.0 + r["k"] * r["n"] * 4.0 + r["m"] * r["n"] * 4.0 * (2.0 if r.get("res") else 1.0)
    # r6: a position launch on the tail plan is TWO dispatches of the kernel (main + K-sliced tail): per-launch figures are per
    # DISPATCH, which is what rocprofv3's per-kernel average and the PMC bytes per launch count
    ndisp = sum(int(r.get("dispatches", 1)) for r in sel)
    alg_bytes = ab / ndisp
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
            "kernel": kname, "rocprof_kernel": rocprof_name(dom) if math == "f16x3" else "conv_gemm_f32_kernel<1, 7, 4, 1>",
            "math": math, "launches": ndisp, "avg_launch_ms": ms / ndisp, "host_calls": len(sel),
            "algorithmic_gflop_per_launch": fl / ndisp / 1e9, "algorithmic_bytes_per_launch": alg_bytes,
            "flops_note": ("achieved / frac price the multiply-adds the kernel EXECUTES: this kernel is the Winograd-W position "
                           "launch of the 3x3x3 convs (F(4,3) along W: 13.5 of the direct form's 27 multiply-adds per output; F(2,3): 18), so "
                           "the direct-form-equivalent rate of those convs -- with their output-transform launch counted in -- "
                           "is `winograd_direct_equivalent_tflops`") if wino_dom else
                          "achieved / frac price the direct form's algorithmic flops, all of which this kernel executes",
            "winograd_direct_equivalent_tflops": (
                sum(r["flops_direct"] for r in sel) / (sum(r["e0"].elapsed_time(r["e2"]) for r in sel) * 1e-3) / 1e12
                if wino_dom else None),
            "winograd_output_transform_ms_per_step_share": xform_ms / wall_ms if xform_ms else None,
            "share_of_wall_time": ms / wall_ms,
            "all_gemm_tflops": all_fl / (all_ms * 1e-3) / 1e12,
            "all_gemm_direct_equivalent_tflops": all_fl_direct / (all_ms * 1e-3) / 1e12,
            "all_gemm_share_of_wall_time": all_ms / wall_ms}


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): re-execute this file under torch.distributed.run
    with N ranks on 127.0.0.1, one GPU each, and hand back its exit code -- never a silent one-rank fallback."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and not os.environ.get("CS_BENCH_ONE_DEVICE"):
        print(f"bench.py: --gpus {a.gpus} but only {ndev} HIP device(s) are visible -- refusing to run fewer ranks than "
              "asked", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, CS_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU fallback)"
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # test hooks (a 1-GPU box cannot host two RCCL ranks): CS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # CS_DIST_BACKEND=gloo swaps the process-group backend, so the multi-rank control flow can be exercised there
    if os.environ.get("CS_BENCH_ONE_DEVICE"):
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible HIP devices")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    # CS_BENCH_FORCE_GROUP=1 (test hook, tests/_rccl_single_rank_worker.py): a ONE-rank job still creates its process group
    # through the very `backend="nccl", device_id=` call an 8-GPU job makes and keeps every barrier / all-reduce / all-gather
    # of this file on RCCL (with CS_DIST_FORCE_COLLECTIVES=1 dist.py's data collectives too), so that the init path and the
    # collectives have met real RCCL before the driver's multi-GPU node does
    force_group = world == 1 and os.environ.get("CS_BENCH_FORCE_GROUP") == "1"
    grouped = world > 1 or force_group
    if grouped:
        backend = os.environ.get("CS_DIST_BACKEND", "nccl")
        kw = {}
        if force_group and "MASTER_ADDR" not in os.environ:
            kw = dict(init_method=f"tcp://127.0.0.1:{os.environ.get('CS_RCCL_PORT', '29573')}", rank=0, world_size=1)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend=backend, **kw)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")

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

    def conditioning():
        """(x_T, uc, c) for all objects on rank 0 (None elsewhere): graph synthesis + embeddings + 5 GCN layers + rel_mlp."""
        if rank != 0:
            return None, None, None
        from commonscenes_amd.scene import GraphTripleConvNet, _MLP
        g = synth.random_scene_graph(total, seed=111)
        ssd = conditioning.ssd
        if ssd is None:
            ssd = conditioning.ssd = synth.synth_state_dict(scene_param_shapes(35, 16), device=str(dev))
            conditioning.nets = (GraphTripleConvNet(ssd, "gconv_net_ec_rel", 5), _MLP(ssd, "rel_mlp", 2, False))
        ec, relmlp = conditioning.nets
        tri = g["triples"].to(dev)
        obj_vecs = torch.cat([g["text_feats"].to(dev), ops.embedding(ssd["obj_embeddings_dc.weight"], g["objs"].to(dev)),
                              g["z"].to(dev)], dim=1)
        pred_vecs = torch.cat([g["rel_feats"].to(dev),
                               ops.embedding(ssd["pred_embeddings_dc.weight"], tri[:, 1].contiguous())], dim=1)
        edges = torch.stack([tri[:, 0], tri[:, 2]], dim=1).contiguous()
        rel2, _ = ec(obj_vecs, pred_vecs, edges)
        c_all = relmlp(rel2)[:total].reshape(total, 1, 1280)          # with the GCN
        uc_all = relmlp(obj_vecs)[:total].reshape(total, 1, 1280)     # without
        return synth.gaussian_like("bench:xT", (1, 3, 16, 16, 16)).to(dev), uc_all, c_all

    conditioning.ssd = conditioning.nets = None
    cond_ms = {}
    for leg in ("cold", "warm"):        # cold = + synthetic weights, weight packing, first-launch code loading
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        x_T, uc_all, c_all = conditioning()
        torch.cuda.synchronize()
        cond_ms[leg] = (time.perf_counter() - t_c) * 1e3
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    x_T, uc_all, c_all = D.broadcast_conditioning(x_T, uc_all, c_all, total, dev, src=0)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t_b) * 1e3
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
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not grouped:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    run(0, a.warmup)
    barrier()
    # r6: the headline is timed HOOK-FREE -- the product's own speed.  The per-GEMM HIP-event records the
    # roofline block is computed from are taken in a SECOND pass over the same K steps right after it (same tensors, same
    # launches, the schedule simply continues), itself timed so that the two are reconcilable (`hooks_ms_per_step`).
    ops.GEMM_PROFILE = None
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = []
    dt_prof = None
    if not a.no_gemm_profile:
        ops.GEMM_PROFILE = prof
        t0 = time.perf_counter()
        run(a.warmup + a.steps, a.steps)
        barrier()
        dt_prof = time.perf_counter() - t0
        ops.GEMM_PROFILE = None
    rank_ms = [dt / a.steps * 1e3]
    if grouped:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rank_ms = [float(v.item()) / a.steps * 1e3 for v in allt]
    dt = max_over_ranks(dt)
    finite = bool(torch.isfinite(x).all().item())
    overflow = ops.read_status(dev) != 0

    # ---- extras, outside the timed region: decode of this rank's latents, the all-gather, one object (C2) ----
    decode = e2e = c2 = c7 = c7x5 = mesh = fp32_leg = native = None
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
            # the reference's sampler mini-batch (sdfusion_txt2shape_model.py:493): 7 objects, CFG batch 14
            nb7 = min(7, B)
            x7 = x_T.repeat(nb7, 1, 1, 1, 1).contiguous()
            c7in = torch.cat([uc[:nb7], c[:nb7]])
            df.reset_run_cache() if hasattr(df, "reset_run_cache") else None
            for j in range(3):
                x7, _ = sampler._step(x7, c7in, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(3, 3 + 10):
                x7, _ = sampler._step(x7, c7in, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            d7 = (time.perf_counter() - t0) / 10
            c7 = {"workload": f"the reference's sampler mini-batch: {nb7} objects, CFG batch {2 * nb7}",
                  "ms_per_step": d7 * 1e3, "ms_per_object_step": d7 * 1e3 / nb7,
                  "whole_step_tflops": 2 * nb7 * K.UNET_GFLOP_PER_SAMPLE / d7 / 1e3}
            df.reset_run_cache() if hasattr(df, "reset_run_cache") else None
            # c7x5 (VERDICT r3 next #1a): what the DEFAULT product API delivers for the metric's 32 objects --
            # SDFusionText2ShapeModel.rel2shape(data, ddim_steps=S, uc_scale=3.0): mini-batches of 7 as the reference
            # slices them (sdfusion_txt2shape_model.py:493-511), coalesced into one launch batch (r4), decode included --
            # next to the reference's own schedule of five sequential sampler runs (launch_B=0)
            import tempfile
            from commonscenes_amd.sdfusion import SDFusionText2ShapeModel
            with tempfile.TemporaryDirectory(dir="/tmp") as tdir:
                pm = SDFusionText2ShapeModel(K.write_yaml_configs(tdir, unet=cfg))
            pm.df = pm.df_module = df                     # the weights already resident (same classes the model builds)
            pm.vqvae = pm.vqvae_module = vq
            data = {"sdf": torch.zeros(B, 1), "rel": c, "uc": uc}
            pm.rel2shape(data, ddim_steps=S, uc_scale=3.0, x_T=x_T, max_steps=2)            # warm the allocator
            c7x5 = {"workload": f"{B} objects through the default product call rel2shape(ddim_steps={S}, uc_scale=3.0): "
                                f"mini_B={pm.mini_B} slices, launch_B={pm.launch_B}; decode to 64^3 included"}
            for key, kw in (("default_api", {}), ("reference_minibatching", {"launch_B": 0})):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g_sdf = pm.rel2shape(data, ddim_steps=S, uc_scale=3.0, x_T=x_T, **kw)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - t0
                c7x5[key] = {"seconds": dtp, "launch_sizes": list(pm.last_launch_sizes), "steps_per_s": S / dtp,
                             "ms_per_object_step": dtp * 1e3 / (S * B), "finite": bool(torch.isfinite(g_sdf).all().item())}
            c7x5["speedup"] = c7x5["reference_minibatching"]["seconds"] / c7x5["default_api"]["seconds"]
            del g_sdf, pm
        if world == 1 and a.driver == "python":
            # r4: the NATIVE whole-forward driver (csrc/cs_unet.hip: cs_unet_step -- what a C / C++ /
            # any-FFI host of the library calls) on the same two workloads, beside the Python sequencer the metric above
            # ran on; same kernels, same bits (tests/test_unet_native_gpu.py), so the difference is host overhead only --
            # which matters at one object (~400 launches in ~7 ms), not at 32
            from commonscenes_amd.unet_native import NativeDiffusionUNet
            ndf = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device=dev, math=a.math)
            ndf.load_state_dict(df.state_dict())
            nmodel = K.ScheduleModel(ndf, dev)
            nsamp = DDIMSampler(nmodel)
            nsamp.make_schedule(a.ddim_steps, ddim_eta=0.0, verbose=False)

            def nsteps(xx, cc, n_warm, n_timed):
                for j in range(n_warm):
                    xx, _ = nsamp._step(xx, cc, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for j in range(n_warm, n_warm + n_timed):
                    xx, _ = nsamp._step(xx, cc, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / n_timed
            dn32 = nsteps(x_T.repeat(B, 1, 1, 1, 1).contiguous(), c_in, 2, a.steps)
            ndf.reset_run_cache()
            dn1 = nsteps(x_T.clone(), torch.cat([uc[:1], c[:1]]), 3, 20)
            native = {"driver": "cs_unet_step (csrc/cs_unet.hip), workspace and arena owned by the caller",
                      "ms_per_step": dn32 * 1e3, "steps_per_s": 1.0 / dn32, "steps": a.steps,
                      "c2_ms_per_step": dn1 * 1e3, "python_driver_ms_per_step": dt / a.steps * 1e3,
                      "python_driver_c2_ms_per_step": c2["ms_per_step"] if c2 else None}
            del ndf, nmodel, nsamp
        if world == 1 and a.math == "f16x3" and not a.no_fp32_leg:
            # the same workload on the fp32-input MFMA kernels (the reference's dtype on the matrix pipe it maps to), so
            # that what CS_MATH_F16X3 buys is on the record next to the metric; 1 warm-up + 2 timed steps
            df.set_math("fp32")
            xf = x_T.repeat(B, 1, 1, 1, 1).contiguous()
            xf, _ = sampler._step(xf, c_in, int(ts[0]), S - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in (1, 2):
                xf, _ = sampler._step(xf, c_in, int(ts[j]), S - j - 1, True, 3.0, want_pred_x0=False)
            torch.cuda.synchronize()
            dfp = (time.perf_counter() - t0) / 2
            fp32_leg = {"math": "fp32 (v_mfma_f32_32x32x2_f32 on fp32 operands)", "value": 1.0 / dfp,
                        "ms_per_step": dfp * 1e3, "steps": 2,
                        "whole_step_tflops": 2 * B * K.UNET_GFLOP_PER_SAMPLE / dfp / 1e3,
                        "frac_of_fp32_matrix_peak": 2 * B * K.UNET_GFLOP_PER_SAMPLE / dfp / 1e3 / FP32_MFMA_PEAK_TFLOPS}
            df.set_math("f16x3")
            del xf

    if rank == 0:
        wall_ms = (dt_prof if dt_prof is not None else dt) * 1e3       # the pass the event records were taken in
        roof = gemm_summary(prof, wall_ms, a.math) or {"bound": "mfma", "achieved": None, "peak": None,
                                                       "unit": "TFLOP/s", "frac": None}   # --driver native: no hooks
        if a.gemm_table:
            agg = {}
            for r in prof:
                k = (r["taps"], r["m"], r["k"], r["n"], r["tile"] + (100 if r.get("slab") else 0) + (1000 if r.get("wino") else 0))
                t = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
                t[0] += 1
                t[1] += r["e0"].elapsed_time(r["e1"])
                t[2] += r["flops"]
                if "e2" in r:                 # Winograd-W: the output-transform launch and the direct form's flops beside
                    t[3] += r["e1"].elapsed_time(r["e2"])
                    t[4] += r["flops_direct"]
            print(f"{'taps':>4} {'M':>7} {'K':>6} {'N':>6} tile {'calls':>5} {'ms/step':>8} {'TF/s':>7}   (tile 1104 = the "
                  "Winograd-W position GEMMs of a 3x3x3 conv: executed TF/s, then + output transform ms/step and the "
                  "direct-form-equivalent TF/s of the pair)", file=sys.stderr)
            for k, t in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
                extra = f"  + {t[3] / a.steps:6.3f}  {t[4] / (t[1] + t[3]) / 1e9:7.1f}" if t[3] else ""
                print(f"{k[0]:4d} {k[1]:7d} {k[2]:6d} {k[3]:6d} {k[4]:4d} {t[0]:5d} {t[1] / a.steps:8.3f} "
                      f"{t[2] / t[1] / 1e9:7.1f}{extra}", file=sys.stderr)
        # HBM bytes per launch of the dominant kernel come from PMC counters, which cannot be collected from inside this
        # process: --traffic runs the two rocprofv3 --pmc passes NOW (child runs of this script); without it the field
        # is null and the figure measured for this round's kernels is the committed profiles/r03_traffic.json
        roof["traffic"] = None
        roof["traffic_note"] = ("null: not measured in this run (PMC counters need rocprofv3 around the process; "
                                "`bench.py --traffic` measures it in the run that prints it: profiles/r03_traffic.json is "
                                "that measurement for this round's kernels, 562 MB per launch)")
        want_traffic = a.traffic or not (a.no_traffic or a.no_extras or a.small)
        # never from inside a profiler session (the child passes would inherit the outer tool's injection)
        traced = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_TOOL", "ROCTRACER")) or k == "HSA_TOOLS_LIB"]
        if traced and not a.traffic:
            want_traffic = False
            roof["traffic_note"] = ("null: this run is itself being profiled (" + traced[0] + " is set), the --pmc passes were "
                                    "skipped; profiles/r03_traffic.json is the measurement for this round's kernels")
        if want_traffic and world == 1 and roof.get("rocprof_kernel"):
            try:
                tr = measure_traffic(roof["rocprof_kernel"], a)
            except Exception as e:                       # a profiler hiccup must not cost the bench line
                tr = None
                roof["traffic_note"] = f"null: the rocprofv3 --pmc passes failed ({type(e).__name__}: {e})"
            if tr:
                roof["traffic"] = tr["hbm_bytes_per_launch"]
                if roof.get("algorithmic_bytes_per_launch"):
                    roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
                roof["traffic_note"] = tr["note"]
                roof["traffic_detail"] = tr
            # r4: matrix-pipe occupancy + effective clock of the dominant kernel, the pointwise token kernel and the UNet's
            # 1024-token attention kernel, measured on THIS run's binaries (one more --pmc pass)
            tok = "conv_gemm_f16x3_kernel<1, 7, 8, 1, false, 0, true, 9, true>"
            att = "attn_f16x3_kernel<2, 64"
            try:
                mb = measure_mfma_busy([roof["rocprof_kernel"], tok, att], a)
            except Exception as e:
                mb = None
                roof["mfma_busy_note"] = f"null: the rocprofv3 --pmc pass failed ({type(e).__name__}: {e})"
            if mb:
                dk = mb.get(roof["rocprof_kernel"])
                if dk:
                    roof["mfma_busy_frac"] = dk["mfma_busy_frac"]
                    roof["effective_clock_ghz"] = dk["effective_clock_ghz"]
                roof["mfma_busy_detail"] = {"dominant": dk, "token_gemm_pointwise_pair": mb.get(tok),
                                            "attention_1024_tokens": mb.get(att)}
                roof["mfma_busy_note"] = ("measured in this run: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                                          "SQ_INSTS_MFMA GRBM_GUI_ACTIVE around `bench.py --steps 2 --warmup 1`; busy = "
                                          "MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8 XCDs), clock = GUI_ACTIVE / 8 / duration")
        roof["whole_step_tflops"] = (2 * B * K.UNET_GFLOP_PER_SAMPLE * 1e9 * a.steps / dt / 1e12) if not a.small else None
        # executed work = what the launches of the timed region actually issued (HIP-event records of every GEMM) + the
        # self-attention and norm terms of SURVEY App. A; it is below the reference's direct form because (a) the two
        # Upsample convs run folded onto the source grid (12/27 of their multiply-adds) and (b) under classifier-free
        # guidance the skip half of output blocks 5-8 is shared by the two halves and its part of the convs runs once
        if prof and not a.small:
            ex_step = sum(r["flops"] for r in prof) / a.steps + 2 * B * (10.45 + 0.3) * 1e9
            roof["whole_step_executed_tflops"] = ex_step / (dt / a.steps) / 1e12
            roof["executed_gflop_per_step"] = ex_step / 1e9
            # the WHOLE step against the same ceiling the dominant kernel's `frac` is priced on (833 = 2500 / 3 passes)
            roof["whole_step_executed_frac"] = roof["whole_step_executed_tflops"] / (
                F16_MFMA_PEAK_TFLOPS / 3.0 if a.math == "f16x3" else FP32_MFMA_PEAK_TFLOPS)
        else:
            roof["whole_step_executed_tflops"] = None
            roof["whole_step_executed_frac"] = None
        roof["measured_in"] = ("a second pass of the same K steps right after the hook-free timed region (two HIP events per GEMM "
                               "launch); `value` / `ms_per_step` are the hook-free pass")
        roof["profiled_pass_ms_per_step"] = dt_prof / a.steps * 1e3 if dt_prof is not None else None
        roof["hooks_ms_per_step"] = (dt_prof - dt) / a.steps * 1e3 if dt_prof is not None else None
        roof["whole_step_note"] = ("whole_step_tflops prices the reference's direct-form work (557.9 GFLOP per UNet sample: "
                                   f"{2 * B * K.UNET_GFLOP_PER_SAMPLE:.0f} GFLOP per step); whole_step_executed_tflops "
                                   "counts what was issued: the Upsample convs run folded onto the source grid (12/27 of "
                                   "their multiply-adds, cs_conv_gemm_up2) and the skip half of output blocks 5-8, identical "
                                   "for the two guidance halves, is convolved once (unet.py::_res_split)")
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
            "roofline": roof, "decode": decode, "end_to_end": e2e, "c2": c2, "c7": c7, "c7x5": c7x5, "native_driver": native, "fp32_mfma": fp32_leg,
            "mesh": mesh,
            "conditioning_ms": cond_ms["warm"], "conditioning_cold_ms": cond_ms["cold"],
            "conditioning_note": "rank 0: graph synthesis + embeddings + 5 GCN layers + rel_mlp for all objects; cold = the "
                                 "first call (synthetic scene weights, weight packing, code loading), warm = the second",
            "ranks": {"world_size": world, "backend": ("rccl (torch 'nccl')" if backend == "nccl" else backend),
                      "rccl_ranks": world if backend == "nccl" else (1 if world == 1 else 0),
                      "process_group": ("forced one-rank group (CS_BENCH_FORCE_GROUP=1)" if force_group
                                        else ("yes" if grouped else "none (single process)")),
                      "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms), "ms_per_step_by_rank": rank_ms,
                      "broadcast_ms": bcast_ms, "broadcast_bytes": int(4 * (3 * 16 ** 3 + 2 * total * 1280)),
                      "all_gather_ms": e2e["all_gather_ms"] if e2e else None,
                      "launched_by": "bench.py (self-spawned torch.distributed.run)" if os.environ.get("CS_BENCH_SPAWNED")
                                     else ("external launcher" if world > 1 else "single process")},
            "finite": finite, "f16x3_overflow": overflow, "unet_driver": a.driver,
        }
        if not a.no_cpu_baseline and not a.small and world == 1:      # rank 0 at N = 1 only: the other ranks would idle
            res["cpu_baseline"] = cpu_baseline(df, cfg, B, quick=a.cpu_baseline == "quick")
        print(json.dumps(res), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

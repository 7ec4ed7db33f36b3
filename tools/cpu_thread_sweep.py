"""Thread sweep of the CPU baseline on the GPU box's host (16 of 256 cores is a protocol deviation from
BASELINE.md section 4's "all host cores" unless the table says it is the fastest setting at the batch sizes the baseline
legs use).  Times ONE CFG DDIM step of the CPU oracle (oracle/ref_torch.py, the port pinned on the reference's goldens)
  * as one batch of 32 objects (CFG batch 64: bench.py's `b32_one_batch` leg) and
  * as one sampler mini-batch of 7 objects (CFG batch 14: the reference's own schedule, `b32_minibatch7`)
at each thread count.  Usage: python tools/cpu_thread_sweep.py [threads ...] > profiles/rNN_cpu_threads.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import synth
from commonscenes_amd.unet import unet_param_shapes
from oracle import ref_torch as R

cfg = dict(R.UNET_FULL, dims=3, use_spatial_transformer=True)
dev = "cuda" if torch.cuda.is_available() else "cpu"
sd = {k: v.cpu() for k, v in synth.synth_state_dict(unet_param_shapes(cfg), device=dev).items()}
sch = R.register_schedule(**R.DIFFUSION)
fn = lambda a, t, cc: R.unet_forward(sd, cfg, a, t, cc)


def one_step(n_obj):
    x_T = synth.gaussian_like("sweep:xT", (1, 3, 16, 16, 16)).repeat(n_obj, 1, 1, 1, 1)
    c = synth.gaussian_like("sweep:c", (n_obj, 1, 1280))
    uc = synth.gaussian_like("sweep:uc", (n_obj, 1, 1280))
    t0 = time.perf_counter()
    with torch.no_grad():
        R.ddim_sample(fn, sch["alphas_cumprod"], 100, x_T, c, uc, 3.0, max_steps=1)
    return time.perf_counter() - t0


host = os.cpu_count() or 1
counts = [int(a) for a in sys.argv[1:]] or sorted({16, 32, 64, 128, host})
print(f"# host cores {host}, torch {torch.__version__}; one CFG DDIM step of oracle/ref_torch.py (shipped UNet, fp32)")
print(f"# {'threads':>7s} {'32 objects, one batch (s)':>26s} {'steps/s @32':>12s} {'7 objects (s)':>14s} {'steps/s @32 as 7+7+7+7+4':>25s}")
torch.set_num_threads(min(host, 16))
one_step(1)                                    # warm-up (allocator, oneDNN primitive caches)
for n in counts:
    torch.set_num_threads(n)
    one_step(2)
    t32 = one_step(32)
    t7 = one_step(7)
    print(f"  {n:7d} {t32:26.2f} {1.0 / t32:12.4f} {t7:14.2f} {1.0 / (t7 * 32 / 7):25.4f}", flush=True)

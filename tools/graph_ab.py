"""eager loop vs hipGraph replay of the DDIM step at small batches (DDIMSampler.use_graph): ms per step over a 60-step run"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import configs as K, synth
from commonscenes_amd.ddim import DDIMSampler
from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes

dev = torch.device("cuda", 0)
cfg = dict(K.UNET_CROSSATTN)
df = DiffusionUNet(cfg, conditioning_key="crossattn", device=dev).set_math("f16x3")
df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device="cuda"))
model = K.ScheduleModel(df, dev)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 7]:
    x_T = synth.gaussian_like("g:xT", (1, 3, 16, 16, 16)).cuda().repeat(B, 1, 1, 1, 1)
    c = synth.gaussian_like("g:c", (B, 1, 1280)).cuda()
    uc = synth.gaussian_like("g:uc", (B, 1, 1280)).cuda()
    res = {}
    outs = {}
    for graph in (False, True, False, True):
        smp = DDIMSampler(model)
        smp.use_graph = graph
        df.reset_run_cache()
        smp.sample(S=100, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T, verbose=False,
                   unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0, max_steps=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, _ = smp.sample(S=100, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T, verbose=False,
                          unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0, max_steps=60)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        res.setdefault(graph, []).append(dt * 1e3)
        outs[graph] = x
    print(f"objects={B}: eager {res[False][0]:.3f} / {res[False][1]:.3f} ms/step, graph (capture included) {res[True][0]:.3f} / {res[True][1]:.3f}; "
          f"bit-equal {bool(torch.equal(outs[False], outs[True]))}", flush=True)

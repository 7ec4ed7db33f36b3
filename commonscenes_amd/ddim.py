"""DDIM sampler behind the reference's `DDIMSampler` interface, device work in HIP.

Mirrors model/networks/diffusion_networks/samplers/ddim.py:15-244 (make_schedule, sample,
ddim_sampling, p_sample_ddim) and ldm_diffusion_util.py:68-96 (timestep selection, sampling
parameters).  The coefficient tables are built on the host exactly as the reference builds them
(including its fp32 / float64 rounding points); the per-step work -- classifier-free
guidance combine + x0 prediction + x_{t-1} update -- is ONE fused elementwise kernel
(cs_ddim_cfg_update) instead of ~12 ATen launches, and the loop issues no host synchronisation
(the reference's CrossAttention NaN traps cost 44 syncs per UNet forward).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import ops

Tensor = torch.Tensor


def make_ddim_timesteps(ddim_discr_method: str, num_ddim_timesteps: int, num_ddpm_timesteps: int,
                        verbose: bool = False) -> np.ndarray:
    """ldm_diffusion_util.py:68-82."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums: Tensor, ddim_timesteps: np.ndarray, eta: float, verbose: bool = False):
    """ldm_diffusion_util.py:85-96.  `alphacums` is the fp32 CPU tensor of cumulative alphas; the
    result dtypes follow the reference: alphas fp32, alphas_prev float64 holding fp32 values."""
    alphas = alphacums[ddim_timesteps]                                   # fp32 tensor
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    a64 = alphas.numpy().astype(np.float64)
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - a64) * (1 - a64 / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
    return sigmas, alphas, alphas_prev


class DDIMSampler(object):
    def __init__(self, model, schedule: str = "linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        # Optional: replay ONE captured HIP graph of a sampling step for every timestep (ddim_sampling).  Bit-identical
        # to the eager loop and kernel for kernel the same work.  Measured on MI355X it buys nothing here (1 object:
        # 17.8 vs 17.2 ms/step, 32 objects: 101.7 vs 100.5) -- the loop is GPU-bound at every batch size, the host
        # runs ahead of the device -- so it is opt-in (use_graph = True or CS_DDIM_GRAPH=1).
        self.use_graph = os.environ.get("CS_DDIM_GRAPH", "0") == "1"

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        ac = alphas_cumprod.detach().to(torch.float32).cpu()
        self.register_buffer("alphas_cumprod", ac)
        sig, a, ap = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta, verbose)
        # host-side tables (python floats at use); the values match the reference's buffers bit for bit
        self.register_buffer("ddim_sigmas", np.asarray(sig, dtype=np.float64))
        self.register_buffer("ddim_alphas", a.numpy().astype(np.float32))
        self.register_buffer("ddim_alphas_prev", np.asarray(ap, dtype=np.float64))
        self.register_buffer("ddim_sqrt_one_minus_alphas", torch.sqrt(1.0 - a).numpy().astype(np.float32))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1.,
               noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None,
               mm_cls_free=False, **kwargs):
        if isinstance(conditioning, dict):
            raise NotImplementedError("dict conditioning (ddim.py:85-88) is not on the rel2shape path: pass the "
                                      "conditioning tensor ([B,1,1280] crossattn / [B,1,16,16,16] concat)")
        if conditioning is not None:
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        for name, val, ok in (("mask", mask, None), ("x0", x0, None), ("score_corrector", score_corrector, None)):
            if val is not ok:
                raise NotImplementedError(f"DDIMSampler.sample({name}=...) is not on the rel2shape path")
        if quantize_x0 or mm_cls_free or noise_dropout > 0. or temperature != 1.:
            raise NotImplementedError("quantize_x0 / mm_cls_free / noise_dropout / temperature are not on the "
                                      "rel2shape path (sdfusion_txt2shape_model.py:498-507)")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if len(shape) == 4:
            C_, D, H, W = shape
            size = (batch_size, C_, D, H, W)
        else:
            C_, H, W = shape
            size = (batch_size, C_, H, W)
        if verbose:
            print(f"Data shape for DDIM sampling is {size}, eta {eta}")
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  max_steps=kwargs.get("max_steps"))

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, max_steps=None):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        timesteps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        # hoisted out of the loop: the conditioning batch [uc; c] never changes (ddim.py:208)
        c_in = torch.cat([unconditional_conditioning, cond]) if cfg else cond
        if (self.use_graph and img.is_cuda and callback is None and img_callback is None
                and not np.any(np.asarray(self.ddim_sigmas) != 0.0)):
            return self._graph_sampling(img, c_in, time_range, total_steps, cfg, unconditional_guidance_scale,
                                        log_every_t, max_steps, intermediates)
        for i, step in enumerate(time_range):
            if max_steps is not None and i >= max_steps:
                break
            index = total_steps - i - 1
            want_p0 = bool(img_callback) or index % log_every_t == 0 or index == total_steps - 1
            img, pred_x0 = self._step(img, c_in, int(step), index, cfg, unconditional_guidance_scale,
                                      want_pred_x0=want_p0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def _graph_sampling(self, img, c_in, time_range, total_steps, cfg, scale, log_every_t, max_steps,
                        intermediates):
        """ddim_sampling's loop (ddim.py:146-179) as hipGraph replays: the ~600 launches of one step (UNet forward
        at [uc; c] + fused guidance/update) are captured once on torch's capture stream and replayed per timestep.
        What changes between replays lives in device memory: the timestep vector (feeds cs_timestep_embedding)
        and the 5-float coefficient block (cs_ddim_cfg_update_dev).  Deterministic sampler only (eta == 0)."""
        dev = img.device
        b = img.shape[0]
        n = len(time_range) if max_steps is None else min(int(max_steps), len(time_range))
        if n <= 0:
            return img, intermediates
        idxs = [total_steps - i - 1 for i in range(n)]
        tab = torch.tensor([ops.ddim_coefficients(float(self.ddim_alphas[k]), float(self.ddim_alphas_prev[k]), 0.0,
                                                  float(self.ddim_sqrt_one_minus_alphas[k])) for k in idxs],
                           dtype=torch.float32, device=dev)
        x_buf = img.contiguous().clone()
        p0_buf = torch.empty_like(x_buf)
        t_buf = torch.empty((b,), dtype=torch.long, device=dev)
        coef = torch.empty((5,), dtype=torch.float32, device=dev)

        def body():
            eps = self._eps(x_buf, t_buf, c_in, cfg)
            ops.ddim_cfg_update_dev(x_buf, eps, coef, float(scale), cfg, pred_x0=p0_buf, out=x_buf)

        def stage(i):
            t_buf.fill_(int(time_range[i]))
            coef.copy_(tab[i])

        # warm-up outside the capture (fills the model's per-run caches, e.g. the one-token context vectors)
        stage(0)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            body()
        cur.wait_stream(side)
        x_buf.copy_(img)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        x_buf.copy_(img)          # capture does not execute, but keep the invariant explicit
        for i in range(n):
            stage(i)
            graph.replay()
            k = idxs[i]
            if k % log_every_t == 0 or k == total_steps - 1:
                intermediates["x_inter"].append(x_buf.clone())
                intermediates["pred_x0"].append(p0_buf.clone())
        out = x_buf.clone()
        del graph
        return out, intermediates

    def _eps(self, x, t, c_in, cfg: bool):
        """e_t for the batch (ddim.py:200-210): [uc; c] halves when cfg is on.  `t`: int64 [b] on the device."""
        fast = getattr(self.model, "apply_model_cfg", None) if cfg else None
        if fast is not None:
            # the two guidance halves share (x, t): let the model evaluate the context-free prefix once
            return fast(x, t, c_in)
        if cfg:
            return self.model.apply_model(torch.cat([x, x]), torch.cat([t, t]), c_in)
        return self.model.apply_model(x, t, c_in)

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, mm_cls_free=False):
        """Reference-signature single step (ddim.py:182-244): returns (x_prev, pred_x0)."""
        if (repeat_noise or use_original_steps or quantize_denoised or temperature != 1. or noise_dropout > 0.
                or score_corrector is not None or mm_cls_free):
            raise NotImplementedError("only the options used by rel2shape are implemented")
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        c_in = torch.cat([unconditional_conditioning, c]) if cfg else c
        return self._step(x, c_in, int(t.flatten()[0].item()), int(index), cfg, unconditional_guidance_scale)

    @torch.no_grad()
    def _step(self, x, c_in, step: int, index: int, cfg: bool, scale: float, want_pred_x0: bool = True):
        """One fused DDIM step; c_in is [uc; c] when cfg is on."""
        b = x.shape[0]
        eps = self._eps(x, torch.full((b,), step, device=x.device, dtype=torch.long), c_in, cfg)
        sigma = float(self.ddim_sigmas[index])
        noise = torch.randn_like(x) if sigma != 0.0 else None     # eta == 0: sigma_t * noise == 0 exactly
        return ops.ddim_cfg_update(x, eps, float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index]),
                                   sigma, float(self.ddim_sqrt_one_minus_alphas[index]), float(scale), cfg,
                                   noise=noise, want_pred_x0=want_pred_x0)

"""PLMS sampler behind the reference's `PLMSSampler` interface, device work in HIP.

Mirrors model/networks/diffusion_networks/samplers/plms.py:15-236: make_schedule (eta must be 0), sample,
plms_sampling, p_sample_plms -- pseudo improved Euler on the first step (two model evaluations), then
Adams-Bashforth orders 2-4 over the last three noise predictions.  Per step the guidance combine, the multistep
combination and the x0 / x_{t-1} update are ONE fused kernel (cs_plms_update); the UNet evaluation goes through the
same `apply_model` / `apply_model_cfg` surface as the DDIM sampler, so both samplers run the same HIP forward.
(The reference file imports from a package called `models`; the tree only has `model`, so it is not importable
as shipped -- tools/make_goldens.py aliases the name to generate the golden this sampler is tested against.)
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from . import ops
from .ddim import make_ddim_sampling_parameters, make_ddim_timesteps

Tensor = torch.Tensor


class PLMSSampler(object):
    def __init__(self, model, schedule: str = "linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")                                    # plms.py:29-30
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        ac = alphas_cumprod.detach().to(torch.float32).cpu()
        self.register_buffer("alphas_cumprod", ac)
        sig, a, ap = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta, verbose)
        self.register_buffer("ddim_sigmas", np.asarray(sig, dtype=np.float64))
        self.register_buffer("ddim_alphas", a.numpy().astype(np.float32))
        self.register_buffer("ddim_alphas_prev", np.asarray(ap, dtype=np.float64))
        self.register_buffer("ddim_sqrt_one_minus_alphas", torch.sqrt(1.0 - a).numpy().astype(np.float32))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1.,
               noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if isinstance(conditioning, dict):
            raise NotImplementedError("dict conditioning is not on the rel2shape path: pass the conditioning tensor")
        if conditioning is not None and conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        for name, val in (("mask", mask), ("x0", x0), ("score_corrector", score_corrector)):
            if val is not None:
                raise NotImplementedError(f"PLMSSampler.sample({name}=...) is not on the rel2shape path")
        if quantize_x0 or noise_dropout > 0. or temperature != 1.:
            raise NotImplementedError("quantize_x0 / noise_dropout / temperature are not on the rel2shape path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if len(shape) == 4:
            C_, D, H, W = shape
            size = (batch_size, C_, D, H, W)
        else:
            C_, H, W = shape
            size = (batch_size, C_, H, W)
        if verbose:
            print(f"Data shape for PLMS sampling is {size}")
        return self.plms_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  max_steps=kwargs.get("max_steps"))

    def _eps(self, x, step: int, c_in, cfg: bool):
        """the model at (x, t = step) for the whole batch: [eps_uc; eps_c] when cfg is on (plms.py:181-189)."""
        t = torch.full((x.shape[0],), int(step), device=x.device, dtype=torch.long)
        fast = getattr(self.model, "apply_model_cfg", None) if cfg else None
        if fast is not None:
            return fast(x, t, c_in)
        if cfg:
            return self.model.apply_model(torch.cat([x, x]), torch.cat([t, t]), c_in)
        return self.model.apply_model(x, t, c_in)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, max_steps=None):
        device = self.model.device
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        timesteps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        scale = float(unconditional_guidance_scale)
        c_in = torch.cat([unconditional_conditioning, cond]) if cfg else cond
        old_eps: List[Tensor] = []                      # oldest first, at most three (plms.py:164-166)
        for i, step in enumerate(time_range):
            if max_steps is not None and i >= max_steps:
                break
            index = total_steps - i - 1
            step_next = int(time_range[min(i + 1, len(time_range) - 1)])
            a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
            s1m = float(self.ddim_sqrt_one_minus_alphas[index])
            eps = self._eps(img, int(step), c_in, cfg)
            hist = old_eps[::-1]                        # newest first
            if not hist:
                # pseudo improved Euler (plms.py:217-221): a plain step, the model again at (x_prev, t_next), then the
                # real step with the average of the two predictions
                x_half, _, e_t = ops.plms_update(img, eps, [], ops.PLMS_PLAIN, a_t, a_prev, s1m, scale, cfg,
                                                 want_pred_x0=False)
                eps_next = self._eps(x_half, step_next, c_in, cfg)
                img, pred_x0, _ = ops.plms_update(img, eps_next, [e_t], ops.PLMS_EULER_AVG, a_t, a_prev, s1m, scale,
                                                  cfg, want_e=False)
            else:
                mode = (ops.PLMS_AB2, ops.PLMS_AB3, ops.PLMS_AB4)[len(hist) - 1]
                img, pred_x0, e_t = ops.plms_update(img, eps, hist, mode, a_t, a_prev, s1m, scale, cfg)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

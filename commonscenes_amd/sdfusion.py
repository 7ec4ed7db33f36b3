"""Shape-diffusion model behind the reference's `SDFusionText2ShapeModel` interface (inference side).

Mirrors model/sdfusion_txt2shape_model.py:
  __init__ (:51-132)  builds df (DiffusionUNet), the DDPM schedule, the DDIM sampler, loads the VQ-VAE
  register_schedule (:184-236), apply_model (:275-291), set_input (:238-256), rel2shape (:459-516)
Like the reference's BaseModel (model/base_model.py:31) this is a plain class, not an nn.Module
(SURVEY F11): `.df` / `.vqvae` carry their own state and are saved under 'df' / 'vqvae' keys.

Extensions over the reference (SURVEY 8b "Extension the build adds"):
  * rel2shape(..., x_T=None, mini_B=None, launch_B=None): inject the shared initial noise (the reference seeds it
    from time.time(), F7) and choose the sampler mini-batch.  `mini_B` defaults to the reference's hard-coded 7
    (sdfusion_txt2shape_model.py:493): the objects are SLICED exactly as the reference slices them.  Since r4 the
    slices are then COALESCED: with the deterministic sampler (eta = 0: no per-mini-batch noise draw) consecutive
    mini-batches are launched as one sampler run of up to `launch_B` objects (default 32 -> ceil(32 / 7) = 5
    mini-batches per launch; `model.launch_B`, CS_LAUNCH_B; 0 = one sampler run per mini-batch, the r3 behaviour).
    Objects are independent -- batch dimensions never mix -- so a coalesced launch gives every object the result of
    its own mini-batch to ~1e-5 (different GEMM tilings add in a different order; gated by
    tests/test_parity_depth_gpu.py), and the default API delivers the benchmarked 32-object throughput
    (2.6 instead of 3.3 ms per object-step on an MI355X) instead of five sequential 7-object runs.
"""
from __future__ import annotations

import logging
import os

import time
from functools import partial
from pathlib import Path
from typing import Optional

import numpy as np
import torch
import yaml

from . import dist
from . import lib as L
from . import ops
from .ddim import DDIMSampler
from .unet import DiffusionUNet
from .unet_native import NativeDiffusionUNet
from .vqvae_native import NativeVQVAE
from .vqvae import VQVAE, load_vqvae

Tensor = torch.Tensor
_log = logging.getLogger("commonscenes_amd.sdfusion")


class AttrDict(dict):
    """OmegaConf-lite: attribute access over the YAML mapping (OmegaConf is not installed here)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_attr(v) for v in o]
    return o


def load_yaml(path) -> AttrDict:
    with open(path) as f:
        return to_attr(yaml.safe_load(f))


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3) -> np.ndarray:
    """ldm_diffusion_util.py:43-65 ('linear' is what the shipped config uses)."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


class SDFusionText2ShapeModel:
    def __init__(self, opt, resolve_dir: Optional[Path] = None):
        """`opt`: the v2_full.yaml mapping (hyper / network / misc sections).  Relative df_cfg / vq_cfg /
        vq_ckpt paths are resolved against `resolve_dir` (default: cwd, like the reference)."""
        self.opt = opt if isinstance(opt, AttrDict) else to_attr(opt)
        self.isTrain = False
        self.model_name = self.name()
        self.device = torch.device(self.opt.hyper.device)
        if self.device.type != "cuda":
            raise RuntimeError("commonscenes_amd runs on the HIP device only (hyper.device must be 'cuda')")
        base = Path(resolve_dir) if resolve_dir else Path.cwd()
        rp = lambda p: str(p) if Path(p).is_absolute() else str(base / p)
        assert self.opt.network.df_cfg is not None and self.opt.network.vq_cfg is not None
        df_conf = load_yaml(rp(self.opt.network.df_cfg))
        vq_conf = load_yaml(rp(self.opt.network.vq_cfg))

        ddconfig = vq_conf.model.params.ddconfig                       # :68-72
        shape_res = ddconfig.resolution
        z_ch, n_down = ddconfig.z_channels, len(ddconfig.ch_mult) - 1
        z_sp = shape_res // (2 ** n_down)
        self.z_shape = (z_ch, z_sp, z_sp, z_sp)

        # two interchangeable sequencers of the same HIP kernels: the Python ones (unet.py, vqvae.py) and the native
        # whole-forward drivers (csrc/cs_unet.hip cs_unet_step, csrc/cs_vqvae.hip cs_vqvae_decode); bit-identical
        driver = str(self.opt.network.get("unet_driver") or os.environ.get("CS_UNET_DRIVER", "python"))
        if driver not in ("python", "native"):
            raise ValueError(f"unet_driver must be 'python' or 'native', got {driver!r}")
        unet_cls = NativeDiffusionUNet if driver == "native" else DiffusionUNet
        self.df = unet_cls(df_conf.unet.params, vq_conf=vq_conf,
                           conditioning_key=df_conf.model.params.conditioning_key, device=self.device)
        self.init_diffusion_params(uc_scale=3., df_model_params=df_conf.model.params)
        self.ddim_sampler = DDIMSampler(self)
        ck = self.opt.network.get("vq_ckpt")
        mp = vq_conf.model.params
        if driver == "native":
            self.vqvae = NativeVQVAE(mp.ddconfig, mp.n_embed, mp.embed_dim, device=self.device)
            if ck is not None and Path(rp(ck)).exists():
                vsd = torch.load(rp(ck), map_location="cpu")
                self.vqvae.load_state_dict(vsd["vqvae"] if "vqvae" in vsd else vsd)
        elif ck is not None and Path(rp(ck)).exists():
            self.vqvae = load_vqvae(vq_conf, rp(ck), device=str(self.device))
        else:   # weights to be supplied through load_state_dict (e.g. a full checkpoint's 'vqvae' entry)
            self.vqvae = VQVAE(mp.ddconfig, mp.n_embed, mp.embed_dim, device=self.device)
        self.df_module = self.df
        self.vqvae_module = self.vqvae
        self.ddim_steps = 100                                           # :128 (hard-coded in the reference)
        self.mini_B = int(os.environ.get("CS_MINI_B", "7"))               # :493 (hard-coded 7 in the reference)
        # objects per sampler LAUNCH: consecutive mini-batches of the deterministic sampler run as one batch (see the
        # module docstring); 0 = never coalesce
        self._launch_B = int(os.environ.get("CS_LAUNCH_B", "32"))
        self._launch_B_explicit = "CS_LAUNCH_B" in os.environ      # (rel2shape_many's fill-the-chip default)
        # which fall-backs this model has taken (sticky, like the math mode itself): `unet_fp32` / `vqvae_fp32` = an F16X3
        # operand left the fp16 range and the module continues on the fp32-input MFMA kernels (4-5x slower);
        # `splitk_two_kernel` = a fused split-K launch was not resident.  Every rel2shape* call reports a copy in `last_meta`
        # (and returns it with return_meta=True); each transition is logged at WARNING.
        self.fell_back = {"unet_fp32": False, "vqvae_fp32": False, "splitk_two_kernel": False}
        self.last_meta = None

    @property
    def launch_B(self) -> int:
        return self._launch_B

    @launch_B.setter
    def launch_B(self, v):           # an assignment is an explicit cap: rel2shape_many honours it
        self._launch_B = int(v)
        self._launch_B_explicit = True

    def name(self):
        return "SDFusion-Text2Shape-Model"

    # ---- schedule (:152-236) ----
    def init_diffusion_params(self, uc_scale=3., df_model_params=None):
        self.parameterization = "eps"
        self.register_schedule(timesteps=df_model_params.timesteps, linear_start=df_model_params.linear_start,
                               linear_end=df_model_params.linear_end)
        self.uc_scale = uc_scale
        self.scale = uc_scale     # the reference reads an undefined self.scale when uc_scale=None (:474)

    def register_schedule(self, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2):
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        to_torch = partial(torch.tensor, dtype=torch.float32)
        # kept on the host: the sampler only ever indexes them from Python (ddim.py:31-57)
        self.betas = to_torch(betas)
        self.alphas_cumprod = to_torch(alphas_cumprod)
        self.alphas_cumprod_prev = to_torch(alphas_cumprod_prev)
        self.sqrt_alphas_cumprod = to_torch(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = to_torch(np.sqrt(1. - alphas_cumprod))

    # ---- reference surface ----
    def set_input(self, input=None, max_sample=None):
        self.x = input["sdf"]
        self.rel = input["rel"]
        self.uc_rel = input["uc"]
        if self.df.conditioning_key == "concat":                        # :246-248: the condition is a volume
            B = self.rel.shape[0]
            C_, D, H, W = self.z_shape
            self.rel = self.rel.reshape(B, -1, D, H, W)
            self.uc_rel = self.uc_rel.reshape(B, -1, D, H, W)
        if max_sample is not None:
            self.x, self.rel, self.uc_rel = self.x[:max_sample], self.rel[:max_sample], self.uc_rel[:max_sample]

    def switch_eval(self):
        self.df.eval()
        self.vqvae.eval()

    def switch_train(self):
        raise NotImplementedError("training is out of scope (SURVEY 3.4)")

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """:275-291"""
        if isinstance(cond, dict):
            return self.df(x_noisy, t, **cond)
        if not isinstance(cond, list):
            cond = [cond]
        key = "c_concat" if self.df_module.conditioning_key == "concat" else "c_crossattn"
        return self.df(x_noisy, t, **{key: cond})

    def apply_model_cfg(self, x, t, c_in):
        """[eps_uc; eps_c] for the guidance pair batch (ddim.py:206-209) without duplicating (x, t)."""
        return self.df.forward_cfg(x, t, c_in)

    # ---- F16X3 overflow policy -------------------------------------------------------------------------------------
    # CS_MATH_F16X3 carries activations as fp16 pairs of a * 16: |a| >= ~4094 cannot be represented.  Every F16X3
    # kernel reports that in a sticky device word (CS_STATUS_F16X3_OVERFLOW); it is read back ONCE per sampler run /
    # decode.  'fp32' (default): re-run that mini-batch on the fp32-input MFMA kernels and stay there;
    # 'raise': propagate CsOverflowError.
    overflow_policy = os.environ.get("CS_OVERFLOW_POLICY", "fp32")

    def _sample_minibatch(self, sampler, ddim_steps, shape, c, uc, noise, uc_scale, ddim_eta, max_steps):
        def run():
            if hasattr(self.df, "reset_run_cache"):
                self.df.reset_run_cache()           # per-run caches (one-token context vectors) never outlive a run
            # the status word is sticky per device: clear what earlier, unchecked F16X3 launches (a direct decode, a
            # bench loop, a call that raised before its own check) may have left, so only THIS run is attributed.
            # Stream-ordered clear, no read-back: the one host sync per run is check_overflow's below.
            ops.clear_status(self.device)
            out, _ = sampler.sample(S=ddim_steps, batch_size=c.shape[0], shape=shape, conditioning=c, x_T=noise,
                                    verbose=False, unconditional_guidance_scale=uc_scale,
                                    unconditional_conditioning=uc, eta=ddim_eta, max_steps=max_steps)
            ops.check_overflow(self.device, "DDIM sampling (UNet)")
            return out
        try:
            try:
                return run()
            except L.CsSplitKTimeout as e:
                # a fused split-K launch was not resident (CUs held by another stream / a CU mask): same arithmetic on the
                # two-kernel form, which needs no co-residency -- for the rest of the process
                import warnings
                warnings.warn(f"{e}; re-running with the two-kernel split-K")
                _log.warning("fused split-K reduce timed out: re-running this mini-batch with no_fused_reduce=1")
                L.debug_set(no_fused_reduce=1)
                self.fell_back["splitk_two_kernel"] = True
                return run()
        except L.CsOverflowError:
            if self.overflow_policy != "fp32" or getattr(self.df, "math", None) != L.MATH_F16X3:
                raise
            import warnings
            warnings.warn("F16X3 activation overflow in the UNet: re-running this mini-batch (and continuing) on the "
                          "fp32-input MFMA kernels (set_math('fp32'))")
            _log.warning("F16X3 activation overflow in the UNet: this model continues on the fp32-input MFMA kernels "
                         "(~4-5x slower); tools/check_checkpoint.py prints the per-layer bounds of a checkpoint")
            self.df.set_math("fp32")
            self._df_fell_back = True
            self.fell_back["unet_fp32"] = True
            return run()

    def _decode_checked(self, samples):
        ops.clear_status(self.device)               # see _sample_minibatch: attribute only this decode's kernels
        out = self.vqvae_module.decode_no_quant(samples)
        try:
            try:
                ops.check_overflow(self.device, "VQ-VAE decode")
            except L.CsSplitKTimeout as e:              # see _sample_minibatch
                _log.warning("fused split-K reduce timed out in the VQ-VAE decode: re-running with no_fused_reduce=1 (%s)", e)
                L.debug_set(no_fused_reduce=1)
                self.fell_back["splitk_two_kernel"] = True
                ops.clear_status(self.device)
                out = self.vqvae_module.decode_no_quant(samples)
                ops.check_overflow(self.device, "VQ-VAE decode")
        except L.CsOverflowError:
            if (self.overflow_policy != "fp32" or getattr(self.vqvae, "math", None) != L.MATH_F16X3
                    or not hasattr(self.vqvae, "set_math")):
                raise
            import warnings
            warnings.warn("F16X3 activation overflow in the VQ-VAE decoder: re-running on the fp32-input MFMA kernels")
            _log.warning("F16X3 activation overflow in the VQ-VAE decoder: this model continues on the fp32-input MFMA kernels")
            self.vqvae.set_math("fp32")
            self._vq_fell_back = True
            self.fell_back["vqvae_fp32"] = True
            out = self.vqvae_module.decode_no_quant(samples)
        return out

    def _sync_math_across_ranks(self):
        """Sharded runs: the fp32 fall-back is decided per rank (only the rank whose shard overflowed switches), which
        would leave the ranks on different numerics for every later call -- shards then stop matching the single-rank
        run bit for bit.  One tiny all-reduce after the shard is done: if ANY rank fell back, all do."""
        # (explicit fell-back flags, set where the fall-back is taken: a module without a `math` attribute, or one that
        # was STARTED in fp32 (CS_MATH=fp32), must not read as "fell back" and drag its peers to fp32 -- ADVICE r3)
        flags = torch.tensor([float(bool(getattr(self, "_df_fell_back", False))),
                              float(bool(getattr(self, "_vq_fell_back", False)))], device=self.device)
        flags = dist.all_reduce_max(flags)
        if flags[0] > 0 and getattr(self.df, "math", None) == L.MATH_F16X3:
            _log.warning("another rank's UNet fell back to fp32: this rank follows (all ranks stay on the same numerics)")
            self.df.set_math("fp32")
            self._df_fell_back = True
            self.fell_back["unet_fp32"] = True
        if flags[1] > 0 and getattr(self.vqvae, "math", None) == L.MATH_F16X3 and hasattr(self.vqvae, "set_math"):
            _log.warning("another rank's VQ-VAE decoder fell back to fp32: this rank follows")
            self.vqvae.set_math("fp32")
            self._vq_fell_back = True
            self.fell_back["vqvae_fp32"] = True

    def _launch_slices(self, lo: int, hi: int, mini_B: Optional[int], launch_B: Optional[int], ddim_eta: float):
        """The sampler launches over objects [lo, hi).  :493-511 slices the objects into ceil(B / 7) mini-batches and runs
        the sampler once per slice.  Every object starts from its scene's x_T and the batch dimension never mixes, so with
        eta = 0 (no noise draw inside the loop) the slices are independent of how they are grouped: WHOLE mini-batches
        are coalesced into launches.  launch_B counts objects and is rounded UP to whole mini-batches (launch_B = 32,
        mini_B = 7: up to five slices = 35 objects per launch -- a cap on memory should be given as a multiple of
        mini_B); the mini-batches are then dealt EVENLY over the fewest launches that respects it, so 40 objects run as
        21 + 19, not as 35 + 5 with an inefficient tail launch.  eta > 0 keeps one launch per slice: the
        sampler draws one noise tensor per slice and step (ddim.py:240), which grouping would re-order."""
        mb = max(1, int(mini_B or self.mini_B))
        lb = int(self.launch_B if launch_B is None else launch_B)
        n = hi - lo
        if n <= 0:
            return []
        n_mb = -(-n // mb)
        per = -(-lb // mb) if (lb > mb and float(ddim_eta) == 0.0) else 1
        n_launch = -(-n_mb // per)
        base, extra = divmod(n_mb, n_launch)
        out, i = [], lo
        for k in range(n_launch):
            cnt = (base + (1 if k < extra else 0)) * mb
            out.append(slice(i, min(i + cnt, hi)))
            i += cnt
        return out

    @torch.no_grad()
    def rel2shape_many(self, datas, ddim_steps=100, ddim_eta=0.0, uc_scale=None, x_Ts=None, mini_B: Optional[int] = None,
                       return_latents: bool = False, max_steps: Optional[int] = None, sampler: str = "ddim",
                       launch_B: Optional[int] = None, return_meta: bool = False):
        """Extension: rel2shape for SEVERAL scenes in one coalesced sampler + decode.
        Single-rank only (it has no object sharding: under torch.distributed with more than one rank it raises -- call
        rel2shape per scene there); `launch_B=None` = model.launch_B when that was set explicitly (attribute / CS_LAUNCH_B),
        else the fill-the-chip default of 64 objects per launch.

        The reference's evaluation loop calls rel2shape once per scene (scripts/eval_3dfront.py:484-513 ->
        VAEGAN_V2FULL.py:600-618 -> sdfusion_txt2shape_model.py:459-516) with 5-15 shaped objects each: small batches, the
        regime furthest from the roofline.  Objects never mix along the batch dimension (GroupNorm / LayerNorm /
        attention are per sample), so the scenes' objects are concatenated and sampled together; what is kept per scene is
        its SHARED x_T (:489-491: one noise volume repeated over the scene's objects) -- x_Ts[i] (or a fresh time-seeded
        draw per scene) is repeated over scene i's objects only.  Returns a list of per-scene gen_df (and latents).
        Per-object results equal the per-scene call bit for bit whenever the GEMM plan is the same (the plan depends on
        the launch's object count), within fp32 summation order otherwise.  Deterministic samplers only (eta = 0)."""
        if float(ddim_eta) != 0.0:
            raise ValueError("rel2shape_many coalesces scenes: deterministic sampling (ddim_eta = 0) only")
        if dist.world()[1] > 1:
            raise RuntimeError("rel2shape_many is single-rank (no object sharding / rank-failure agreement): call "
                               "rel2shape(sharded=True) per scene under torch.distributed")
        if x_Ts is not None and len(x_Ts) != len(datas):
            raise ValueError(f"rel2shape_many: {len(datas)} scenes but {len(x_Ts)} x_Ts (one per scene, or None)")
        self.switch_eval()
        if sampler == "ddim":
            smp = DDIMSampler(self)
        elif sampler == "plms":
            from .plms import PLMSSampler
            smp = PLMSSampler(self)
        else:
            raise ValueError(f"sampler must be 'ddim' or 'plms', got {sampler!r}")
        if ddim_steps is None:
            ddim_steps = self.ddim_steps
        if uc_scale is None:
            uc_scale = self.scale
        shape = self.z_shape
        C_, D, H, W = shape
        counts, cs, ucs, noises = [], [], [], []
        for i, data in enumerate(datas):
            self.set_input(data)
            c_i = self.rel.to(device=self.device, dtype=torch.float32)
            cs.append(c_i)
            ucs.append(self.uc_rel.to(device=self.device, dtype=torch.float32))
            counts.append(int(c_i.shape[0]))
            if x_Ts is None or x_Ts[i] is None:
                torch.manual_seed(int(time.time()) + i)                  # :489 (time-seeded in the reference), per scene
                n_i = torch.randn((1, C_, D, H, W), device=self.device)
            else:
                n_i = x_Ts[i].to(device=self.device, dtype=torch.float32).reshape(1, C_, D, H, W)
            noises.append(n_i.expand(counts[-1], C_, D, H, W))
        total = sum(counts)
        r = self.vqvae.cfg["resolution"] if hasattr(self.vqvae, "cfg") else 64
        och = self.vqvae.cfg.get("out_ch", 1) if hasattr(self.vqvae, "cfg") else 1
        self.last_launch_sizes = []
        if total == 0:
            gens = [torch.empty((0, och, r, r, r), dtype=torch.float32, device=self.device) for _ in counts]
            lat = [torch.empty((0, C_, D, H, W), dtype=torch.float32, device=self.device) for _ in counts]
            return (gens, lat) if return_latents else gens
        c_all, uc_all = torch.cat(cs, dim=0), torch.cat(ucs, dim=0)
        noise_all = torch.cat(noises, dim=0).contiguous()
        gen, lats = [], []
        if launch_B is None:
            # scenes are batched to fill the chip (up to 64 objects per launch) -- unless the operator capped launch_B
            # (memory): an explicit attribute / CS_LAUNCH_B is honoured, never silently raised
            launch_B = int(self.launch_B) if self._launch_B_explicit else 64
        for sl in self._launch_slices(0, total, mini_B, launch_B, ddim_eta):
            self.last_launch_sizes.append(sl.stop - sl.start)
            samples = self._sample_minibatch(smp, ddim_steps, shape, c_all[sl], uc_all[sl], noise_all[sl].contiguous(),
                                             uc_scale, ddim_eta, max_steps)
            lats.append(samples)
            gen.append(self._decode_checked(samples))
        g_all, l_all = torch.cat(gen, dim=0), torch.cat(lats, dim=0)
        gens = list(torch.split(g_all, counts, dim=0))
        lat = list(torch.split(l_all, counts, dim=0))
        self.last_latents, self.gen_df = l_all, g_all
        meta = self._meta(ddim_steps, sampler)
        out = (gens, lat) if return_latents else gens
        if return_meta:
            return (*out, meta) if return_latents else (out, meta)
        return out

    def _meta(self, ddim_steps, sampler):
        """what the last rel2shape* call ran as: kept in `last_meta`, returned with return_meta=True."""
        self.last_meta = {"fell_back": dict(self.fell_back), "any_fell_back": any(self.fell_back.values()),
                          "launch_sizes": list(self.last_launch_sizes), "ddim_steps": int(ddim_steps), "sampler": sampler,
                          "unet_math": {L.MATH_FP32: "fp32", L.MATH_F16X3: "f16x3"}.get(getattr(self.df, "math", None)),
                          "vqvae_math": {L.MATH_FP32: "fp32", L.MATH_F16X3: "f16x3"}.get(getattr(self.vqvae, "math", None))}
        return self.last_meta

    @torch.no_grad()
    def rel2shape(self, data, ddim_steps=100, ddim_eta=0.0, uc_scale=None, x_T: Optional[Tensor] = None,
                  mini_B: Optional[int] = None, return_latents: bool = False, max_steps: Optional[int] = None,
                  sharded: Optional[bool] = None, sampler: str = "ddim", launch_B: Optional[int] = None,
                  return_meta: bool = False):
        """:459-516.  data = {'sdf': (B,...) only its batch size is used, 'rel': (B,1,1280), 'uc': (B,1,1280)}.
        `return_meta=True` appends a dict (also kept as `self.last_meta`): `fell_back` (which fall-backs this model has taken:
        fp32 re-runs after an F16X3 range overflow, the two-kernel split-K), the launch sizes, the math modes.

        Extensions: `x_T` injection (the reference seeds from the clock), `mini_B` (reference: 7),
        `launch_B` (objects per sampler launch: consecutive mini-batches of the deterministic sampler are coalesced, see
        the module docstring; None = self.launch_B, 0 = one launch per mini-batch as in the reference),
        `sampler` ('ddim' | 'plms', samplers/plms.py), and `sharded`: with torch.distributed initialised (one process
        per GPU, RCCL) the objects are split contiguously over the ranks -- rank 0's (x_T, uc, c) is broadcast once,
        every rank samples + decodes its shard, one all-gather returns ALL objects on every rank.  No per-step
        collective.  Default (None): shard iff a process group with more than one rank exists and CS_SHARD != 0."""
        self.switch_eval()
        self.set_input(data)
        if sampler == "ddim":
            smp = DDIMSampler(self)
        elif sampler == "plms":
            from .plms import PLMSSampler
            smp = PLMSSampler(self)
        else:
            raise ValueError(f"sampler must be 'ddim' or 'plms', got {sampler!r}")
        if ddim_steps is None:
            ddim_steps = self.ddim_steps
        if uc_scale is None:
            uc_scale = self.scale
        uc = self.uc_rel.to(device=self.device, dtype=torch.float32)
        c_text = self.rel.to(device=self.device, dtype=torch.float32)
        B = c_text.shape[0]
        shape = self.z_shape
        C_, D, H, W = shape
        if x_T is None:
            torch.manual_seed(int(time.time()))                          # :489 (time-seeded in the reference)
            single_noise = torch.randn((1, C_, D, H, W), device=self.device)
        else:
            single_noise = x_T.to(device=self.device, dtype=torch.float32).reshape(1, C_, D, H, W)
        rank, ws = dist.world()
        if sharded is None:
            sharded = ws > 1 and os.environ.get("CS_SHARD", "1") != "0"
        multi = ws > 1 or dist._force()     # (_force: test-only, drives the collectives through a one-rank RCCL group)
        lo, hi = 0, B
        if sharded and multi:
            # one broadcast of the packed [x_T | uc | c] buffer: every rank then holds rank 0's bits (x_T is
            # time-seeded, and the GCN that produced uc / c ran on every rank or only on rank 0 -- either way)
            single_noise, uc, c_text = dist.broadcast_conditioning(
                single_noise, uc, c_text, B, self.device, src=0, latent_shape=shape,
                ctx_dim=int(c_text[0].numel()) if B else 0, cond_shape=tuple(c_text.shape[1:]))
            lo, hi = dist.shard_range(B, ws, rank)
        r = self.vqvae.cfg["resolution"] if hasattr(self.vqvae, "cfg") else 64
        och = self.vqvae.cfg.get("out_ch", 1) if hasattr(self.vqvae, "cfg") else 1
        self.last_launch_sizes = []
        gen, lats = [], []
        failure: Optional[BaseException] = None
        try:
            for sl in self._launch_slices(lo, hi, mini_B, launch_B, ddim_eta):
                self.last_launch_sizes.append(sl.stop - sl.start)
                num = sl.stop - sl.start
                noise = single_noise.repeat(num, 1, 1, 1, 1)             # every object shares one x_T (:491)
                samples = self._sample_minibatch(smp, ddim_steps, shape, c_text[sl], uc[sl], noise, uc_scale, ddim_eta,
                                                 max_steps)
                lats.append(samples)
                gen.append(self._decode_checked(samples))
        except Exception as e:              # noqa: BLE001 -- re-raised below, after the other ranks have been told
            if not (sharded and multi):
                raise
            failure = e
        if sharded and multi:
            # a rank that raised must not leave its peers blocked in the all-gather: agree on failure first
            bad = dist.any_rank_failed(failure is not None, self.device)
            if failure is not None:
                raise failure
            if bad:
                raise RuntimeError("rel2shape: another rank failed while sampling its shard (see its traceback); "
                                   "no SDFs were gathered")
            self._sync_math_across_ranks()
        if gen:
            local, llat = torch.cat(gen, dim=0), torch.cat(lats, dim=0)
        else:       # an empty shard (more ranks than objects) or B == 0: nothing to sample, still join the gather
            local = torch.empty((0, och, r, r, r), dtype=torch.float32, device=self.device)
            llat = torch.empty((0, C_, D, H, W), dtype=torch.float32, device=self.device)
        if sharded and multi:
            local = dist.all_gather_objects(local, B)
            if return_latents:
                llat = dist.all_gather_objects(llat, B)
        self.last_latents = llat            # this rank's (or, with return_latents, all) sampled latents: diagnostics
        self.gen_df = local
        meta = self._meta(ddim_steps, sampler)
        if return_latents:
            return (self.gen_df, llat, meta) if return_meta else (self.gen_df, llat)
        return (self.gen_df, meta) if return_meta else self.gen_df

    # ---- the other callers of the same sampler (sdfusion_txt2shape_model.py:368-457) -----------------------------------
    # The reference runs these over ALL objects in one sampler call (no mini-batching) with fresh noise per object
    # (DDIMSampler draws torch.randn when x_T is None); same here, plus the x_T / max_steps injection of rel2shape.
    def _sample_all(self, c, uc, ddim_steps, ddim_eta, uc_scale, x_T=None, max_steps=None):
        if ddim_steps is None:
            ddim_steps = self.ddim_steps
        if uc_scale is None:
            uc_scale = self.uc_scale
        c = c.to(device=self.device, dtype=torch.float32)
        uc = None if uc is None else uc.to(device=self.device, dtype=torch.float32)
        samples = self._sample_minibatch(self.ddim_sampler, ddim_steps, self.z_shape, c, uc, x_T, uc_scale, ddim_eta,
                                         max_steps)
        self.last_latents = samples
        self.gen_df = self._decode_checked(samples)
        return self.gen_df

    @torch.no_grad()
    def inference(self, data, ddim_steps=None, ddim_eta=0., uc_scale=None, infer_all=False, max_sample=16,
                  x_T: Optional[Tensor] = None, max_steps: Optional[int] = None):
        """:423-457: set_input (at most `max_sample` objects unless infer_all), one guided sampler run over all of them,
        decode into self.gen_df.  (The reference then calls switch_train(): training is out of scope here.)"""
        self.switch_eval()
        self.set_input(data, max_sample=None if infer_all else max_sample)
        return self._sample_all(self.rel, self.uc_rel, ddim_steps, ddim_eta, uc_scale, x_T, max_steps)

    @torch.no_grad()
    def graph2shape(self, num_obj=6, ddim_steps=100, ddim_eta=0.0, uc_scale=None, x_T: Optional[Tensor] = None,
                    max_steps: Optional[int] = None):
        """:390-421: the first num_obj objects of the CURRENT input (set_input must have run), sampled with
        unconditional_conditioning=None -- i.e. WITHOUT classifier-free guidance, whatever uc_scale says (ddim.py:200-201)
        -- and decoded.  Returns gen_df."""
        self.switch_eval()
        return self._sample_all(self.rel[:num_obj], None, ddim_steps, ddim_eta, uc_scale, x_T, max_steps)

    @torch.no_grad()
    def gen_shape_after_foward(self, num_obj, ddim_steps=None, uc_scale=None, ddim_eta=0., x_T: Optional[Tensor] = None,
                               max_steps: Optional[int] = None):
        """:368-386 (sic): guided sampling of the first num_obj objects of the current input, decode into self.gen_df."""
        self.switch_eval()
        return self._sample_all(self.rel[:num_obj], self.uc_rel[:num_obj], ddim_steps, ddim_eta, uc_scale, x_T, max_steps)

    # ---- checkpoint surface (VAEGAN_V2FULL.py:687-699 stores these under 'df' / 'vqvae') ----
    def state_dict(self):
        return {"df": self.df.state_dict(), "vqvae": self.vqvae.state_dict()}

    def load_state_dict(self, sd):
        if "df" in sd:
            self.df.load_state_dict(sd["df"])
        if "vqvae" in sd:
            self.vqvae.load_state_dict(sd["vqvae"])
        return self

#!/usr/bin/env python
"""Generate tests/golden/*.npz by importing the REFERENCE (read-only, /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  Follows the
harness recipe of SURVEY.md Appendix B: stub absent third-party modules, apply three harness-side
monkeypatches (reference files untouched), load deterministic synthetic weights
(commonscenes_amd/synth.py) into the reference modules with their own load_state_dict, run, and
dump small input/output fixtures.  The fixtures are DATA (inputs + expected outputs); nothing of
the reference's source is copied.

    python tools/make_goldens.py [--only NAME ...]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import time
import types
from pathlib import Path
from unittest import mock

import numpy as np
import torch
import yaml

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF))

from commonscenes_amd import synth                       # noqa: E402
from commonscenes_amd.unet import unet_param_shapes      # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


# ---------------------------------------------------------------------------------------------
# harness: stubs + patches (App. B steps 2-3)
# ---------------------------------------------------------------------------------------------
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return _AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return _ListConfig(_wrap(v) for v in o)
    return o


class _ListConfig(list):
    pass


def install_stubs():
    for name in ["cv2", "mcubes", "termcolor", "torchvision", "torchvision.utils", "torchvision.transforms",
                 "fvcore", "fvcore.common", "fvcore.common.param_scheduler", "pytorch3d", "pytorch3d.io",
                 "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.transforms", "pytorch3d.ops",
                 "trimesh", "h5py", "imageio", "skimage", "skimage.measure", "open3d", "tensorboardX", "clip"]:
        if name not in sys.modules:
            m = mock.MagicMock(name=name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    sys.modules["termcolor"].cprint = lambda *a, **k: None
    oc = types.ModuleType("omegaconf")

    class OmegaConf:
        @staticmethod
        def load(path):
            with open(path) as f:
                return _wrap(yaml.safe_load(f))
    oc.OmegaConf = OmegaConf
    lc = types.ModuleType("omegaconf.listconfig")
    lc.ListConfig = _ListConfig
    oc.listconfig = lc
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = lc


def install_patches():
    from model.networks.diffusion_networks.samplers import ddim as ddim_mod
    ddim_mod.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)   # F10
    ddim_mod.tqdm = lambda it, **k: it
    torch.Tensor.cuda = lambda self, *a, **k: self
    _randn = torch.randn

    def randn(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        shape = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else a
        if INJECT.get("x_T") is not None and tuple(shape) == tuple(INJECT["x_T"].shape):
            return INJECT["x_T"].clone()
        return _randn(*a, **k)
    torch.randn = randn
    import model.sdfusion_txt2shape_model as m
    m.init_mesh_renderer = lambda **k: None


INJECT: dict = {}


def save(name: str, **arrs):
    OUT.mkdir(parents=True, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    np.savez_compressed(OUT / f"{name}.npz", **conv)
    sz = (OUT / f"{name}.npz").stat().st_size / 1e6
    print(f"[golden] {name}.npz  {sz:.2f} MB  keys={list(conv)}", flush=True)


# ---------------------------------------------------------------------------------------------
# configs
# ---------------------------------------------------------------------------------------------
def unet_params(small: bool, concat: bool = False):
    with open(REF / "config" / ("sdfusion-txt2shape_concat.yaml" if concat else "sdfusion-txt2shape.yaml")) as f:
        y = yaml.safe_load(f)
    p = dict(y["unet"]["params"])
    if small:
        p["model_channels"] = 32
    p["use_checkpoint"] = False
    return p


def build_ref_unet(small: bool, concat: bool = False):
    from model.networks.diffusion_networks.network import DiffusionUNet
    p = unet_params(small, concat)
    df = DiffusionUNet(_wrap(p), conditioning_key="concat" if concat else "crossattn").eval()
    shapes = unet_param_shapes(p)
    ref_shapes = {k: tuple(v.shape) for k, v in df.state_dict().items()}
    assert ref_shapes == dict(shapes), "unet_param_shapes disagrees with the reference state_dict"
    sd = synth.synth_state_dict(shapes)
    df.load_state_dict(sd, strict=True)
    return df, p, sd


def vq_conf():
    with open(REF / "config" / "vqvae_snet.yaml") as f:
        return _wrap(yaml.safe_load(f))


def build_ref_vqvae():
    from model.networks.vqvae_networks.network import VQVAE
    from commonscenes_amd.vqvae import vqvae_param_shapes
    mp = vq_conf().model.params
    vq = VQVAE(mp.ddconfig, mp.n_embed, mp.embed_dim).eval()
    shapes = vqvae_param_shapes(dict(mp.ddconfig), mp.n_embed, mp.embed_dim)
    ref_shapes = {k: tuple(v.shape) for k, v in vq.state_dict().items()}
    sub = {k: ref_shapes[k] for k in shapes}
    assert sub == dict(shapes), "vqvae_param_shapes disagrees with the reference state_dict"
    sd = synth.synth_state_dict(shapes)
    missing, unexpected = vq.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), missing
    return vq, sd


# ---------------------------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------------------------
def g_schedule():
    from model.networks.diffusion_networks.ldm_diffusion_util import (make_beta_schedule, make_ddim_timesteps,
                                                                       make_ddim_sampling_parameters)
    from model.networks.diffusion_networks.samplers.ddim import DDIMSampler
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    class M:
        num_timesteps = 1000
        device = "cpu"
    m = M()
    m.betas = torch.tensor(betas, dtype=torch.float32)
    m.alphas_cumprod = torch.tensor(alphas_cumprod, dtype=torch.float32)
    m.alphas_cumprod_prev = torch.tensor(np.append(1.0, alphas_cumprod[:-1]), dtype=torch.float32)
    out = dict(betas=m.betas, alphas_cumprod=m.alphas_cumprod)
    for S in (50, 100):
        s = DDIMSampler(m)
        s.make_schedule(S, ddim_eta=0.0, verbose=False)
        out[f"timesteps_{S}"] = np.asarray(s.ddim_timesteps)
        out[f"alphas_{S}"] = np.asarray(s.ddim_alphas, dtype=np.float64)
        out[f"alphas_prev_{S}"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
        out[f"sigmas_{S}"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
        out[f"sqrt_one_minus_alphas_{S}"] = np.asarray(s.ddim_sqrt_one_minus_alphas, dtype=np.float64)
    save("schedule", **out)


def _unet_inputs(B, tag):
    x = synth.gaussian_like(f"{tag}:x", (B, 3, 16, 16, 16))
    ctx = synth.gaussian_like(f"{tag}:ctx", (B, 1, 1280))
    return x, ctx


def g_unet(small: bool):
    name = "unet_small" if small else "unet_full"
    df, p, sd = build_ref_unet(small)
    x, ctx = _unet_inputs(2, name)
    t = torch.tensor([981, 11], dtype=torch.long)
    hooks = {}
    if small:
        net = df.diffusion_net
        watch = {"input_blocks.1": net.input_blocks[1], "input_blocks.3": net.input_blocks[3],
                 "input_blocks.4": net.input_blocks[4], "middle_block": net.middle_block,
                 "output_blocks.2": net.output_blocks[2], "output_blocks.8": net.output_blocks[8]}
        hs = [m.register_forward_hook(lambda mod, i, o, k=k: hooks.__setitem__(k, o.detach().clone()))
              for k, m in watch.items()]
    t0 = time.time()
    with torch.no_grad():
        y = df(x, t, c_crossattn=[ctx])
    print(f"[{name}] reference forward {time.time() - t0:.1f}s  out rms {y.pow(2).mean().sqrt():.4f}")
    arrs = dict(x=x, t=t, ctx=ctx, eps=y)
    for k, v in hooks.items():
        arrs["hook:" + k] = v
    save(name, **arrs)


def g_unet_concat(small: bool):
    """UNet3DModel at config/sdfusion-txt2shape_concat.yaml (dims=4, AttentionBlock) behind DiffusionUNet's concat
    branch (network.py:25-27): x (B,3,16^3) + one condition volume (B,1,16^3)."""
    name = "unet_concat_small" if small else "unet_concat_full"
    df, p, sd = build_ref_unet(small, concat=True)
    x = synth.gaussian_like(f"{name}:x", (2, 3, 16, 16, 16))
    cvol = synth.gaussian_like(f"{name}:c", (2, 1, 16, 16, 16))
    t = torch.tensor([981, 11], dtype=torch.long)
    hooks = {}
    if small:
        net = df.diffusion_net
        watch = {"input_blocks.1": net.input_blocks[1], "input_blocks.3": net.input_blocks[3],
                 "input_blocks.4": net.input_blocks[4], "middle_block": net.middle_block,
                 "output_blocks.2": net.output_blocks[2], "output_blocks.8": net.output_blocks[8]}
        hs = [m.register_forward_hook(lambda mod, i, o, k=k: hooks.__setitem__(k, o.detach().clone()))
              for k, m in watch.items()]
    t0 = time.time()
    with torch.no_grad():
        y = df(x, t, c_concat=[cvol])
    print(f"[{name}] reference forward {time.time() - t0:.1f}s  out rms {y.pow(2).mean().sqrt():.4f}")
    arrs = dict(x=x, t=t, c=cvol, eps=y)
    for k, v in hooks.items():
        arrs["hook:" + k] = v
    save(name, **arrs)


def g_ddim_concat():
    """3 CFG DDIM steps through the reference DDIMSampler with the concat-conditioned reduced-width UNet."""
    from model.networks.diffusion_networks.samplers.ddim import DDIMSampler
    name = "ddim_concat_small"
    df, p, sd = build_ref_unet(True, concat=True)
    m = _ref_model_for_sampler(df, key="c_concat")
    B, S, k = 2, 50, 3
    x_T = synth.gaussian_like(f"{name}:xT", (1, 3, 16, 16, 16)).repeat(B, 1, 1, 1, 1)
    c = synth.gaussian_like(f"{name}:c", (B, 1, 16, 16, 16))
    uc = synth.gaussian_like(f"{name}:uc", (B, 1, 16, 16, 16))
    sampler = DDIMSampler(m)
    sampler.make_schedule(S, ddim_eta=0.0, verbose=False)
    img, xs, p0s = x_T, [], []
    ts = np.flip(sampler.ddim_timesteps)
    for i in range(k):
        tt = torch.full((B,), int(ts[i]), dtype=torch.long)
        img, pred = sampler.p_sample_ddim(img, c, tt, index=S - i - 1, unconditional_guidance_scale=3.0,
                                          unconditional_conditioning=uc)
        xs.append(img.clone())
        p0s.append(pred.clone())
    save(name, x_T=x_T, c=c, uc=uc, S=np.int64(S), steps=np.int64(k), scale=np.float32(3.0),
         x=torch.stack(xs), pred_x0=torch.stack(p0s))


def _ref_model_for_sampler(df, key="c_crossattn"):
    """A minimal stand-in for SDFusionText2ShapeModel exposing what DDIMSampler reads
    (ddim.py:16-20,31-37,134,188), with the reference's own schedule code."""
    from model.networks.diffusion_networks.ldm_diffusion_util import make_beta_schedule
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)

    class M:
        num_timesteps = 1000
        device = "cpu"

        def apply_model(self, x, t, c):
            return df(x, t, **{key: [c]})
    m = M()
    m.betas = torch.tensor(betas, dtype=torch.float32)
    m.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
    m.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
    return m


def g_ddim(small: bool):
    """k-step CFG DDIM trajectories through the reference DDIMSampler (ddim.py:60-244)."""
    from model.networks.diffusion_networks.samplers.ddim import DDIMSampler
    name = "ddim_small" if small else "ddim_full"
    df, p, sd = build_ref_unet(small)
    m = _ref_model_for_sampler(df)
    B = 2 if small else 1
    S, k = (50, 3) if small else (50, 2)
    x_T = synth.gaussian_like(f"{name}:xT", (1, 3, 16, 16, 16)).repeat(B, 1, 1, 1, 1)
    c = synth.gaussian_like(f"{name}:c", (B, 1, 1280))
    uc = synth.gaussian_like(f"{name}:uc", (B, 1, 1280))
    sampler = DDIMSampler(m)
    sampler.make_schedule(S, ddim_eta=0.0, verbose=False)
    # truncate the loop after k steps: run ddim_sampling manually over the first k flipped timesteps
    img = x_T
    xs, p0s = [], []
    ts = np.flip(sampler.ddim_timesteps)
    t0 = time.time()
    for i in range(k):
        index = S - i - 1
        tt = torch.full((B,), int(ts[i]), dtype=torch.long)
        img, pred = sampler.p_sample_ddim(img, c, tt, index=index, unconditional_guidance_scale=3.0,
                                          unconditional_conditioning=uc)
        xs.append(img.clone())
        p0s.append(pred.clone())
    print(f"[{name}] {k} reference DDIM steps {time.time() - t0:.1f}s")
    save(name, x_T=x_T, c=c, uc=uc, S=np.int64(S), steps=np.int64(k), scale=np.float32(3.0),
         x=torch.stack(xs), pred_x0=torch.stack(p0s))


def g_vq():
    vq, sd = build_ref_vqvae()
    h = synth.gaussian_like("vq:latent", (1, 3, 16, 16, 16), scale=0.8)
    with torch.no_grad():
        quant, _, info = vq.quantize(h, is_voxel=True)
        t0 = time.time()
        dec = vq.decode_no_quant(h)
        print(f"[vq] reference decode_no_quant {time.time() - t0:.1f}s")
        dec_nq = vq.decode_no_quant(h[:, :, :8, :8, :8].contiguous(), force_not_quantize=True)
        dec_q = vq.decode(quant)
    assert torch.equal(dec, dec_q)
    save("vq_decode", latent=h, indices=info[2], quant=quant, dec=dec,
         latent_nq=h[:, :, :8, :8, :8].contiguous(), dec_nq=dec_nq)


def _scene_yaml(tmp: Path, small: bool, vq_ckpt: Path, concat: bool = False) -> Path:
    with open(REF / "config" / ("v2_full_concat.yaml" if concat else "v2_full.yaml")) as f:
        y = yaml.safe_load(f)
    y["hyper"]["device"] = "cpu"
    y["hyper"]["logs_dir"] = str(tmp / "logs")
    y["hyper"]["results_dir"] = str(tmp / "logs")
    df_yaml = REF / "config" / ("sdfusion-txt2shape_concat.yaml" if concat else "sdfusion-txt2shape.yaml")
    if small:
        with open(df_yaml) as f:
            d = yaml.safe_load(f)
        d["unet"]["params"]["model_channels"] = 32
        d["unet"]["params"]["use_checkpoint"] = False
        df_yaml = tmp / "df_small.yaml"
        with open(df_yaml, "w") as f:
            yaml.safe_dump(d, f)
    y["network"]["df_cfg"] = str(df_yaml)
    y["network"]["vq_cfg"] = str(REF / "config" / "vqvae_snet.yaml")
    y["network"]["vq_ckpt"] = str(vq_ckpt)
    out = tmp / "v2_full_cpu.yaml"
    with open(out, "w") as f:
        yaml.safe_dump(y, f)
    return out


def build_ref_scene(tmp: Path, small: bool = True, concat: bool = False):
    """Construct the reference Sg2ScVAEModel the way eval does (model/VAE.py:60-62), App. B step 5."""
    from model.VAEGAN_V2FULL import Sg2ScVAEModel
    from commonscenes_amd.scene import scene_param_shapes
    vq, vq_sd = build_ref_vqvae()
    ck = tmp / "vq_synth.pth"
    torch.save(vq.state_dict(), ck)
    n_obj_cls, n_pred = 35, 16
    vocab = dict(object_idx_to_name=[f"obj{i}\n" for i in range(n_obj_cls)],
                 pred_idx_to_name=[f"pred{i}\n" for i in range(n_pred)],
                 object_idx_to_name_grained=[f"objg{i}\n" for i in range(n_obj_cls)])
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        model = Sg2ScVAEModel(vocab, str(_scene_yaml(tmp, small, ck, concat)), diffusion_bs=16, embedding_dim=64,
                              decoder_cat=True, mlp_normalization="batch", gconv_num_layers=5, use_angles=True,
                              distribution_before=True, use_E2=True, replace_latent=True, num_box_params=6,
                              residual=True, clip=True).eval()
    finally:
        os.chdir(cwd)
    shapes = scene_param_shapes(n_obj_cls, n_pred, rel_dims=(1280, 4096) if concat else (960, 1280))
    ref_shapes = {k: tuple(v.shape) for k, v in torch.nn.Module.state_dict(model).items()}
    sub = {k: ref_shapes[k] for k in shapes}
    assert sub == dict(shapes), "scene_param_shapes disagrees with the reference state_dict"
    sd = synth.synth_state_dict(shapes)
    model.load_state_dict(sd, strict=False)
    p = unet_params(small, concat)
    df_sd = synth.synth_state_dict(unet_param_shapes(p))
    model.Diff.df.load_state_dict(df_sd, strict=True)
    model.Diff.df.eval()
    return model, sd, df_sd, vq_sd


def g_gcn(tmp: Path, concat: bool = False):
    model, sd, _, _ = build_ref_scene(tmp, small=True, concat=concat)
    g = synth.random_scene_graph(6, seed=7)
    with torch.no_grad():
        uc, c = model.encoder_2(g["z"], g["objs"], g["triples"], g["text_feats"], g["rel_feats"], None)
    save("gcn_encoder2_concat" if concat else "gcn_encoder2", objs=g["objs"], triples=g["triples"], text_feats=g["text_feats"],
         rel_feats=g["rel_feats"], z=g["z"], uc=uc, c=c)


def g_e2e(tmp: Path, concat: bool = False, full: bool = False, steps: int = 2):
    """Sg2ScVAEModel.sample(gen_shape=True) end to end (VAEGAN_V2FULL.py:600-618): 8 shaped objects (mini-batch
    boundary at 7), 2 DDIM steps; reduced-width UNet, or (full=True) the shipped 413.5 M-parameter one.
    steps=100 (fixture `e2e100_small`) is the metric's own depth: rel2shape's default ddim_steps
    (sdfusion_txt2shape_model.py:459-516), every latent / code index / SDF after the whole 100-step schedule."""
    import functools
    model, sd, df_sd, vq_sd = build_ref_scene(tmp, small=not full, concat=concat)
    nobj = 8
    g = synth.random_scene_graph(nobj, seed=11)
    O = g["objs"].shape[0]
    dec_sdfs = torch.zeros(O, 1, 4, 4, 4)
    dec_sdfs[:nobj] = 1.0                       # floor and _scene_ carry all-zero SDFs (dropped by the mask)
    x_T = synth.gaussian_like("e2e:xT", (1, 3, 16, 16, 16))
    INJECT["x_T"] = x_T
    rec = {}
    enc2 = model.encoder_2

    def enc2_rec(z, *a, **k):
        rec["z"] = z.detach().clone()
        r = enc2(z, *a, **k)
        rec["uc"], rec["c"] = r[0].detach().clone(), r[1].detach().clone()
        return r
    model.encoder_2 = enc2_rec
    model.Diff.rel2shape = functools.partial(model.Diff.rel2shape, ddim_steps=steps)
    lat = []
    dnq = model.Diff.vqvae_module.decode_no_quant

    def dnq_rec(h, *a, **k):
        lat.append(h.detach().clone())
        return dnq(h, *a, **k)
    model.Diff.vqvae_module.decode_no_quant = dnq_rec
    idx_rec = _record_vq_indices(model.Diff.vqvae_module)
    np.random.seed(111)
    t0 = time.time()
    with torch.no_grad():
        boxes, gen_sdf = model.sample(None, np.zeros(64), np.eye(64), g["objs"], g["triples"], dec_sdfs,
                                      g["text_feats"], g["rel_feats"], attributes=None, gen_shape=True)
    print(f"[e2e] reference sample() {time.time() - t0:.1f}s  gen_sdf {tuple(gen_sdf.shape)}")
    INJECT["x_T"] = None
    arrs = dict(objs=g["objs"], triples=g["triples"], text_feats=g["text_feats"], rel_feats=g["rel_feats"],
                dec_sdfs_nonzero=(dec_sdfs.flatten(1).abs().sum(1) > 0), z=rec["z"], uc=rec["uc"], c=rec["c"],
                x_T=x_T, latents=torch.cat(lat, 0), gen_sdf_sub=gen_sdf[:, :, ::2, ::2, ::2].contiguous(),
                gen_sdf_obj7=gen_sdf[7], indices=torch.cat([i.reshape(-1, 16, 16, 16) for i in idx_rec], 0))
    if isinstance(boxes, tuple):
        arrs["boxes"], arrs["angles"] = boxes[0], boxes[1]
    else:
        arrs["boxes"] = boxes
    arrs["ddim_steps"] = np.int64(steps)
    if steps != 2:
        assert not concat
        save(f"e2e{steps}_full" if full else f"e2e{steps}_small", **arrs)
    elif full:
        # keep the fixture small: every other voxel of every object, one object in full
        save("e2e_full", **arrs)
    else:
        save("e2e_concat_small" if concat else "e2e_small", **arrs)


def g_box():
    """BASELINE configs[0]: the layout-only v2_box model (model/VAEGAN_V2BOX.py as model/VAE.py:52-56 builds it) on one
    synthetic 8-node scene graph: encoder, decoder, manipulate, decoder_with_changes, decoder_with_additions."""
    from model.VAEGAN_V2BOX import Sg2ScVAEModel as v2_box
    from commonscenes_amd.scene_box import box_param_shapes
    n_obj_cls, n_pred = 35, 16
    vocab = dict(object_idx_to_name=[f"obj{i}\n" for i in range(n_obj_cls)],
                 pred_idx_to_name=[f"pred{i}\n" for i in range(n_pred)])
    model = v2_box(vocab, embedding_dim=64, decoder_cat=True, mlp_normalization="batch", input_dim=6,
                   replace_latent=True, use_angles=True, residual=True, gconv_pooling="avg", gconv_num_layers=5).eval()
    shapes = box_param_shapes(n_obj_cls, n_pred)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert ref_shapes == dict(shapes), "box_param_shapes disagrees with the reference state_dict"
    sd = synth.synth_state_dict(shapes)
    model.load_state_dict(sd, strict=False)
    g = synth.random_scene_graph(6, seed=13)
    O = g["objs"].shape[0]
    boxes_gt = synth.gaussian_like("box:gt", (O, 6))
    angles_gt = torch.from_numpy(np.floor((synth.hash_uniform("box:ang", O) + 1.0) * 12.0)).long().clamp(0, 23)
    a = (g["objs"], g["triples"])
    tf, rf = g["text_feats"], g["rel_feats"]
    with torch.no_grad():
        mu, logvar = model.encoder(*a, boxes_gt, None, tf, rf, angles_gt)
        z = synth.gaussian_like("box:z", (O, 64))
        d3, ang = model.decoder(z, *a, tf, rf, None)
        man = model.manipulate(torch.cat([z, synth.gaussian_like("box:chg", (O, 64))], dim=1), *a, tf, rf, None)
        # manipulation: node 2 was added (its latent is missing from z), node 4 was relabelled
        z_in = torch.cat([z[:2], z[3:]], dim=0)
        np.random.seed(1234)
        (d3c, angc), keepc = model.decoder_with_changes(z_in, *a, tf, rf, None, [2], [4])
        np.random.seed(99)
        (d3a, anga), keepa = model.decoder_with_additions(z_in, *a, tf, rf, None, [2], [4],
                                                          distribution=(np.zeros(64), np.eye(64)))
        np.random.seed(5)
        d3s, angs = model.sampleBoxes(np.zeros(64), np.eye(64), *a, tf, rf, None)
    save("box_small", objs=g["objs"], triples=g["triples"], text_feats=tf, rel_feats=rf, boxes_gt=boxes_gt,
         angles_gt=angles_gt, mu=mu, logvar=logvar, z=z, d3=d3, angles=ang, man=man, z_in=z_in,
         d3_changes=d3c, angles_changes=angc, keep_changes=keepc, d3_add=d3a, angles_add=anga, keep_add=keepa,
         d3_sample=d3s, angles_sample=angs)


def g_full_manip(tmp: Path):
    """v2_full manipulation surface (VAEGAN_V2FULL.py:185-218 encoder, :244-259 manipulate, :291-396
    decoder_with_additions / decoder_with_changes incl. the gen_shape branch with 2 DDIM steps)."""
    import functools
    model, sd, df_sd, vq_sd = build_ref_scene(tmp, small=True)
    nobj = 6
    g = synth.random_scene_graph(nobj, seed=17)
    O = g["objs"].shape[0]
    a = (g["objs"], g["triples"])
    tf, rf = g["text_feats"], g["rel_feats"]
    boxes_gt = synth.gaussian_like("fm:gt", (O, 6))
    angles_gt = torch.from_numpy(np.floor((synth.hash_uniform("fm:ang", O) + 1.0) * 12.0)).long().clamp(0, 23)
    dec_sdfs = torch.zeros(O, 1, 4, 4, 4)
    dec_sdfs[:nobj] = 1.0
    z = synth.gaussian_like("fm:z", (O, 64))
    z_in = torch.cat([z[:2], z[3:]], dim=0)
    x_T = synth.gaussian_like("fm:xT", (1, 3, 16, 16, 16))
    model.Diff.rel2shape = functools.partial(model.Diff.rel2shape, ddim_steps=steps)
    idx_rec = _record_vq_indices(model.Diff.vqvae_module)
    lat = []
    dnq = model.Diff.vqvae_module.decode_no_quant

    def dnq_rec(h, *aa, **k):
        lat.append(h.detach().clone())
        return dnq(h, *aa, **k)
    model.Diff.vqvae_module.decode_no_quant = dnq_rec
    with torch.no_grad():
        mu, logvar = model.encoder(*a, boxes_gt, None, tf, rf, angles_gt)
        INJECT["x_T"] = x_T
        np.random.seed(1234)
        (d3c, angc), gen_sdf, keepc = model.decoder_with_changes(z_in, *a, tf, rf, dec_sdfs, None, [2], [4],
                                                                 gen_shape=True)
        INJECT["x_T"] = None
        np.random.seed(99)
        (d3a, anga), none_sdf, keepa = model.decoder_with_additions(z_in, *a, tf, rf, dec_sdfs, None, [2], [4],
                                                                    distribution=(np.zeros(64), np.eye(64)))
    assert none_sdf is None and gen_sdf.shape == (nobj, 1, 64, 64, 64)
    save("full_manip_small", objs=g["objs"], triples=g["triples"], text_feats=tf, rel_feats=rf, boxes_gt=boxes_gt,
         angles_gt=angles_gt, dec_sdfs_nonzero=(dec_sdfs.flatten(1).abs().sum(1) > 0), mu=mu, logvar=logvar, z_in=z_in,
         x_T=x_T, d3_changes=d3c, angles_changes=angc, keep_changes=keepc,
         gen_sdf_sub=gen_sdf[:, :, ::2, ::2, ::2].contiguous(), d3_add=d3a, angles_add=anga, keep_add=keepa,
         latents=torch.cat(lat, 0), indices=torch.cat([i.reshape(-1, 16, 16, 16) for i in idx_rec], 0))


def _record_vq_indices(vq):
    """Wrap VQVAE.quantize (quantizer.py:68-119) so every call's argmin indices are kept: the e2e fixtures carry
    them so that a test can tell a VQ code flip (an fp32 near-tie) from a decoder error."""
    rec = []
    fwd = vq.quantize.forward

    def forward(z, *a, **k):
        out = fwd(z, *a, **k)
        rec.append(out[2][2].detach().clone())
        return out
    vq.quantize.forward = forward
    return rec


TRAJ_KEEP = (1, 2, 3, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50)


TRAJ100_KEEP = (1, 2, 5, 10, 25, 50, 75, 100)


def g_traj(small: bool, S: int = 50):
    """S=100 (fixture `traj100_full`): BASELINE configs[2]'s own schedule depth (ddim_steps=100,
    sdfusion_txt2shape_model.py:128,460; loop at samplers/ddim.py:154), one object at the shipped width.
    BASELINE configs[1] (C2): ONE object, the whole 50-step classifier-free-guided DDIM run through the
    reference's own DDIMSampler.sample() loop (ddim.py:60-179), reduced width (small) or the shipped 413.5 M-parameter
    UNet (full).  x after the steps in TRAJ_KEEP is kept so a test can report the per-step growth of the deviation."""
    from model.networks.diffusion_networks.samplers.ddim import DDIMSampler
    name = ("traj_small" if small else "traj_full") if S == 50 else f"traj{S}_{'small' if small else 'full'}"
    keep = TRAJ_KEEP if S == 50 else TRAJ100_KEEP
    df, p, sd = build_ref_unet(small)
    m = _ref_model_for_sampler(df)
    B = 1
    x_T = synth.gaussian_like(f"{name}:xT", (1, 3, 16, 16, 16))
    c = synth.gaussian_like(f"{name}:c", (B, 1, 1280))
    uc = synth.gaussian_like(f"{name}:uc", (B, 1, 1280))
    t0 = time.time()
    with torch.no_grad():
        x, inter = DDIMSampler(m).sample(S=S, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T,
                                         verbose=False, unconditional_guidance_scale=3.0,
                                         unconditional_conditioning=uc, eta=0.0, log_every_t=1)
    print(f"[{name}] {S} reference DDIM steps {time.time() - t0:.1f}s  final rms {x.pow(2).mean().sqrt():.4f}")
    xi = inter["x_inter"]                      # [x_T, x after step 1, ..., x after step S]
    assert len(xi) == S + 1 and torch.equal(xi[-1], x)
    save(name, x_T=x_T, c=c, uc=uc, S=np.int64(S), scale=np.float32(3.0), keep=np.asarray(keep, dtype=np.int64),
         x=torch.stack([xi[k] for k in keep]), pred_x0_final=inter["pred_x0"][-1])


def g_plms():
    """N4: the reference PLMSSampler (samplers/plms.py:61-236) on the reduced-width UNet: B=2, S=50, CFG 3.0, the whole
    50-step run (pseudo improved Euler start-up, then Adams-Bashforth orders 2-4).  plms.py imports from a package
    called `models` (the tree has `model`): the harness aliases the name, no reference file is touched."""
    import importlib
    for sub in ("", ".networks", ".networks.diffusion_networks", ".networks.diffusion_networks.ldm_diffusion_util"):
        sys.modules["models" + sub] = importlib.import_module("model" + sub)
    from model.networks.diffusion_networks.samplers import plms as plms_mod
    plms_mod.PLMSSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    plms_mod.tqdm = lambda it, **k: it
    name = "plms_small"
    df, p, sd = build_ref_unet(True)
    m = _ref_model_for_sampler(df)
    B, S = 2, 50
    x_T = synth.gaussian_like(f"{name}:xT", (1, 3, 16, 16, 16)).repeat(B, 1, 1, 1, 1)
    c = synth.gaussian_like(f"{name}:c", (B, 1, 1280))
    uc = synth.gaussian_like(f"{name}:uc", (B, 1, 1280))
    t0 = time.time()
    with torch.no_grad():
        x, inter = plms_mod.PLMSSampler(m).sample(S=S, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T,
                                                  verbose=False, unconditional_guidance_scale=3.0,
                                                  unconditional_conditioning=uc, eta=0.0, log_every_t=1)
    print(f"[{name}] {S} reference PLMS steps {time.time() - t0:.1f}s  final rms {x.pow(2).mean().sqrt():.4f}")
    xi = inter["x_inter"]
    assert len(xi) == S + 1 and torch.equal(xi[-1], x)
    save(name, x_T=x_T, c=c, uc=uc, S=np.int64(S), scale=np.float32(3.0), keep=np.asarray(TRAJ_KEEP, dtype=np.int64),
         x=torch.stack([xi[k] for k in TRAJ_KEEP]), pred_x0_final=inter["pred_x0"][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    install_stubs()
    install_patches()
    todo = a.only or ["schedule", "unet_small", "unet_full", "ddim_small", "ddim_full", "vq", "gcn", "e2e",
                      "unet_concat_small", "unet_concat_full", "ddim_concat_small", "gcn_concat", "e2e_concat", "box",
                      "full_manip", "e2e_full", "traj_small", "traj_full", "plms", "traj100_full", "e2e100_small", "e2e100_full"]
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td)
        for name in todo:
            print(f"=== {name}", flush=True)
            if name == "schedule":
                g_schedule()
            elif name == "unet_small":
                g_unet(True)
            elif name == "unet_full":
                g_unet(False)
            elif name == "ddim_small":
                g_ddim(True)
            elif name == "ddim_full":
                g_ddim(False)
            elif name == "vq":
                g_vq()
            elif name == "gcn":
                g_gcn(tmp)
            elif name == "e2e":
                g_e2e(tmp)
            elif name == "unet_concat_small":
                g_unet_concat(True)
            elif name == "unet_concat_full":
                g_unet_concat(False)
            elif name == "ddim_concat_small":
                g_ddim_concat()
            elif name == "gcn_concat":
                g_gcn(tmp, concat=True)
            elif name == "e2e_concat":
                g_e2e(tmp, concat=True)
            elif name == "e2e_full":
                g_e2e(tmp, full=True)
            elif name == "traj_small":
                g_traj(True)
            elif name == "traj_full":
                g_traj(False)
            elif name == "plms":
                g_plms()
            elif name == "traj100_full":
                g_traj(False, S=100)
            elif name == "e2e100_small":
                g_e2e(tmp, steps=100)
            elif name == "e2e100_full":          # r6: the shipped width at the metric's depth (~45 CPU-minutes on 8 cores)
                g_e2e(tmp, full=True, steps=100)
            elif name == "box":
                g_box()
            elif name == "full_manip":
                g_full_manip(tmp)
            else:
                raise SystemExit(f"unknown fixture {name}")


if __name__ == "__main__":
    main()

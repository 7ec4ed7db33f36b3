"""The shipped configurations of the shape branch, as plain dicts (the reference keeps them in YAML files that are
resolved relative to its checkout: config/sdfusion-txt2shape.yaml, config/sdfusion-txt2shape_concat.yaml,
config/vqvae_snet.yaml).  `SDFusionText2ShapeModel` still reads the caller's YAML; these are for callers that have
no checkout at hand (bench.py, the C host demo, tests on the GPU box)."""
from __future__ import annotations

# config/sdfusion-txt2shape.yaml:14-38 -- crossattn family: H,W-only resampling (dims: 3), SpatialTransformer3D blocks
UNET_CROSSATTN = dict(image_size=16, in_channels=3, out_channels=3, model_channels=224, num_res_blocks=2,
                      attention_resolutions=(4, 2), channel_mult=(1, 2, 3), num_heads=8, context_dim=1280,
                      dims=3, use_spatial_transformer=True)
# config/sdfusion-txt2shape_concat.yaml:14-38 -- concat family: condition volume as a 4th input channel, dims: 4
UNET_CONCAT = dict(image_size=16, in_channels=4, out_channels=3, model_channels=224, num_res_blocks=2,
                   attention_resolutions=(4, 2), channel_mult=(1, 2, 3), num_heads=8, context_dim=None,
                   dims=4, use_spatial_transformer=False)
# config/vqvae_snet.yaml:8-19
VQVAE_DDCONFIG = dict(double_z=False, z_channels=3, resolution=64, in_channels=1, out_ch=1, ch=64, ch_mult=(1, 2, 4),
                      num_res_blocks=1, attn_resolutions=(), dropout=0.0)
VQVAE_N_EMBED, VQVAE_EMBED_DIM = 8192, 3
# config/sdfusion-txt2shape.yaml:1-7
DIFFUSION = dict(linear_start=0.00085, linear_end=0.012, timesteps=1000)

# algorithmic work, used by bench.py's roofline blocks
UNET_GFLOP_PER_SAMPLE = 557.9        # conv3 471.3 + linear 62.0 + conv1 13.9 + self-attention 10.45 + norms 0.3
VQ_DECODE_GFLOP_PER_OBJECT = 723.4   # conv3 696.9 + conv1 8.6 + GroupNorm 0.7 + attention 17.2
# ... of which the Upsample convs (69.4 / 347.9 GFLOP in the direct, 27-tap form) are EXECUTED on the source grid with
# pre-summed taps (cs_conv_gemm_up2): 12/27 of their multiply-adds in the UNet (H, W doubled), 8/27 in the decoder.
# Throughput figures quoted on the algorithmic numbers above therefore overstate the issued arithmetic by these ratios:
UNET_GFLOP_EXECUTED_PER_SAMPLE = 557.9 - 38.5        # 519.4
VQ_DECODE_GFLOP_EXECUTED_PER_OBJECT = 723.4 - 244.8  # 478.6


def write_yaml_configs(dirpath, unet: dict = None, conditioning_key: str = "crossattn") -> dict:
    """The `opt` mapping SDFusionText2ShapeModel / Sg2ScVAEModel take (config/v2_full.yaml's hyper / network / misc
    sections), with df_cfg / vq_cfg written under `dirpath` from the dicts above -- for callers without a reference
    checkout (bench.py's product-API block, the C host demo)."""
    from pathlib import Path
    import yaml
    d = Path(dirpath)
    d.mkdir(parents=True, exist_ok=True)
    ucfg = dict(unet if unet is not None else (UNET_CONCAT if conditioning_key == "concat" else UNET_CROSSATTN))
    (d / "df.yaml").write_text(yaml.safe_dump(dict(
        model=dict(params=dict(conditioning_key=conditioning_key, **DIFFUSION)),
        unet=dict(params={k: (list(v) if isinstance(v, tuple) else v) for k, v in ucfg.items()}))))
    (d / "vq.yaml").write_text(yaml.safe_dump(dict(model=dict(params=dict(
        embed_dim=VQVAE_EMBED_DIM, n_embed=VQVAE_N_EMBED,
        ddconfig={k: (list(v) if isinstance(v, tuple) else v) for k, v in VQVAE_DDCONFIG.items()})))))
    return dict(hyper=dict(device="cuda", batch_size=4),
                network=dict(df_cfg=str(d / "df.yaml"), vq_cfg=str(d / "vq.yaml"), vq_ckpt=None), misc=dict(seed=111))


def reduced(cfg: dict, model_channels: int = 32) -> dict:
    """the same topology at a smaller width (tests / debugging; never the benchmarked configuration)."""
    return dict(cfg, model_channels=model_channels)


class ScheduleModel:
    """The slice of SDFusionText2ShapeModel a sampler reads (ddim.py:16-20,31-37,134,188): the DDPM schedule
    (sdfusion_txt2shape_model.py:184-236) and apply_model over a DiffusionUNet."""

    def __init__(self, df, device, linear_start=DIFFUSION["linear_start"], linear_end=DIFFUSION["linear_end"],
                 timesteps=DIFFUSION["timesteps"]):
        import numpy as np
        import torch
        from .sdfusion import make_beta_schedule
        betas = make_beta_schedule("linear", timesteps, linear_start=linear_start, linear_end=linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = int(timesteps)
        self.device = device
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.df = df

    def apply_model(self, x, t, c):
        key = "c_concat" if self.df.conditioning_key == "concat" else "c_crossattn"
        return self.df(x, t, **{key: [c]})

    def apply_model_cfg(self, x, t, c_in):
        return self.df.forward_cfg(x, t, c_in)

"""MI355X-native 3D UNet denoiser behind the reference's `DiffusionUNet` interface.

Mirrors (API, parameter names, semantics) of
  model/networks/diffusion_networks/network.py:11-42        DiffusionUNet(unet_params, vq_conf, conditioning_key)
  model/networks/diffusion_networks/openai_model_3d.py:452-789  UNet3DModel
  model/networks/diffusion_networks/attention.py:154-351   CrossAttention / BasicTransformerBlock / SpatialTransformer3D
but is NOT a translation: activations live channels-last (NDHWC == token-major, so the
`b c d h w -> b (d h w) c` rearranges of the reference vanish), every conv / linear is one
implicit-GEMM MFMA kernel with bias / timestep-embedding / residual fused into its epilogue,
q/k/v projections are one GEMM, self-attention is a flash kernel, and the one-token
cross-attention collapses to a per-sample row vector folded into the attn1 output GEMM.

`state_dict()` / `load_state_dict()` speak the reference's key layout.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes as C

import torch

from . import lib as L
from . import ops

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# architecture description (openai_model_3d.py:558-728)
# ------------------------------------------------------------------------------------------------
def _cfg(unet_params) -> dict:
    g = (lambda k, d=None: unet_params.get(k, d)) if isinstance(unet_params, dict) else (
        lambda k, d=None: getattr(unet_params, k, d))
    cfg = dict(
        image_size=int(g("image_size", 16)), in_channels=int(g("in_channels", 3)),
        out_channels=int(g("out_channels", 3)), model_channels=int(g("model_channels")),
        num_res_blocks=int(g("num_res_blocks")), attention_resolutions=tuple(g("attention_resolutions")),
        channel_mult=tuple(g("channel_mult")), num_heads=int(g("num_heads", -1)),
        context_dim=g("context_dim"), dims=int(g("dims", 3)),
        use_spatial_transformer=bool(g("use_spatial_transformer", True)),
    )
    # the two shipped families: crossattn (config/sdfusion-txt2shape.yaml: dims=3 -> H,W-only resampling,
    # SpatialTransformer3D blocks) and concat (config/sdfusion-txt2shape_concat.yaml: dims=4 -> Conv3d with true
    # 3-D stride-2 resampling, conv_nd / Downsample / Upsample at openai_model_3d.py:146-199, AttentionBlock blocks)
    if cfg["dims"] not in (3, 4) or cfg["num_heads"] <= 0:
        raise NotImplementedError("supported: dims in (3, 4) and num_heads > 0 (config/sdfusion-txt2shape*.yaml)")
    if g("use_scale_shift_norm", False) or g("resblock_updown", False) or g("num_classes") is not None:
        raise NotImplementedError("use_scale_shift_norm / resblock_updown / num_classes are not on the path")
    if g("num_head_channels", -1) not in (-1, None) or g("use_new_attention_order", False):
        raise NotImplementedError("num_head_channels / use_new_attention_order are not used by the shipped configs")
    if cfg["use_spatial_transformer"]:
        if cfg["context_dim"] is None:
            raise ValueError("use_spatial_transformer=True needs context_dim (openai_model_3d.py:512-513)")
        cfg["context_dim"] = int(cfg["context_dim"])
    else:
        cfg["context_dim"] = None
    return cfg


def unet_blocks(cfg: dict):
    """Block list with channel bookkeeping.  Each block: list of layer dicts
    {kind: conv_in|res|attn|down|up, idx, cin, cout}."""
    mc, mult, nres = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    inp = [[dict(kind="conv_in", idx=0, cin=cfg["in_channels"], cout=mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [dict(kind="res", idx=0, cin=ch, cout=m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(dict(kind="attn", idx=1, cin=ch, cout=ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([dict(kind="down", idx=0, cin=ch, cout=ch)])
            chans.append(ch)
            ds *= 2
    mid = [dict(kind="res", idx=0, cin=ch, cout=ch), dict(kind="attn", idx=1, cin=ch, cout=ch),
           dict(kind="res", idx=2, cin=ch, cout=ch)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            layers = [dict(kind="res", idx=0, cin=ch + ich, cout=mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(dict(kind="attn", idx=1, cin=ch, cout=ch))
            if level and i == nres:
                layers.append(dict(kind="up", idx=len(layers), cin=ch, cout=ch))
                ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def unet_param_shapes(cfg: dict, prefix: str = "diffusion_net.") -> "OrderedDict[str, Tuple[int, ...]]":
    """Every state_dict entry of the reference UNet3DModel for this config: name -> shape."""
    cfg = _cfg(cfg)
    mc = cfg["model_channels"]
    ted = 4 * mc
    ctx = cfg["context_dim"]
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def lin(p, o, i, bias=True):
        S[p + ".weight"] = (o, i)
        if bias:
            S[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        S[p + ".weight"] = (o, i, k, k, k)
        S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,)
        S[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin)
        conv(p + ".in_layers.2", cout, cin, 3)
        lin(p + ".emb_layers.1", cout, ted)
        norm(p + ".out_layers.0", cout)
        conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cout, cin, 1)

    def attn(p, c):
        if not cfg["use_spatial_transformer"]:          # AttentionBlock (openai_model_3d.py:316-366): Conv1d k=1
            norm(p + ".norm", c)
            S[p + ".qkv.weight"] = (3 * c, c, 1)
            S[p + ".qkv.bias"] = (3 * c,)
            S[p + ".proj_out.weight"] = (c, c, 1)
            S[p + ".proj_out.bias"] = (c,)
            return
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        t = p + ".transformer_blocks.0"
        for a, cd in (("attn1", c), ("attn2", ctx)):
            lin(f"{t}.{a}.to_q", c, c, bias=False)
            lin(f"{t}.{a}.to_k", c, cd, bias=False)
            lin(f"{t}.{a}.to_v", c, cd, bias=False)
            lin(f"{t}.{a}.to_out.0", c, c)
        lin(f"{t}.ff.net.0.proj", 8 * c, c)
        lin(f"{t}.ff.net.2", c, 4 * c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{t}.{n}", c)
        conv(p + ".proj_out", c, c, 1)

    def block(bp, layers):
        for l in layers:
            p = f"{bp}.{l['idx']}"
            if l["kind"] == "conv_in":
                conv(p, l["cout"], l["cin"], 3)
            elif l["kind"] == "res":
                res(p, l["cin"], l["cout"])
            elif l["kind"] == "attn":
                attn(p, l["cin"])
            elif l["kind"] == "down":
                conv(p + ".op", l["cout"], l["cin"], 3)
            elif l["kind"] == "up":
                conv(p + ".conv", l["cout"], l["cin"], 3)

    lin(prefix + "time_embed.0", ted, mc)
    lin(prefix + "time_embed.2", ted, ted)
    inp, mid, out, ch = unet_blocks(cfg)
    for i, layers in enumerate(inp):
        block(f"{prefix}input_blocks.{i}", layers)
    block(prefix + "middle_block", mid)
    for i, layers in enumerate(out):
        block(f"{prefix}output_blocks.{i}", layers)
    norm(prefix + "out.0", ch)
    conv(prefix + "out.2", cfg["out_channels"], mc, 3)
    return S


# ------------------------------------------------------------------------------------------------
# the module
# ------------------------------------------------------------------------------------------------
class DiffusionUNet:
    """Drop-in for reference `DiffusionUNet` (network.py:11-42): `df(x, t, c_crossattn=[ctx])`,
    `.conditioning_key`, `.state_dict()`, `.load_state_dict()`, `.to()`, `.eval()`, `.parameters()`.
    Inference only (the sampler runs under torch.no_grad() in the reference too)."""

    def __init__(self, unet_params, vq_conf=None, conditioning_key: Optional[str] = None,
                 device: str | torch.device = "cuda"):
        self.cfg = _cfg(unet_params)
        self.conditioning_key = conditioning_key
        self.device = torch.device(device)
        self.prefix = "diffusion_net."
        self.shapes = unet_param_shapes(self.cfg, self.prefix)
        self._sd: Dict[str, Tensor] = {}
        self._packed = None
        self.training = False
        self._ctx_cache = None
        self.math = L.DEFAULT_MATH      # F16X3 unless CS_MATH=fp32
        self.attn_math: Optional[int] = None        # None: follow self.math; L.MATH_F16: plain-fp16 attention (opt-in)
        self.trace: Optional[Dict[str, Tensor]] = None   # set to {} to capture per-block outputs (tests)
        self._split_min_rows: Optional[int] = None     # see the split_min_rows property
        self._split_info: Dict[str, Tuple[int, int]] = {}
        self._ngb: Dict[str, Tuple[float, float]] = {}

    @property
    def split_min_rows(self) -> int:
        """Channel-split ResBlocks (see _pack) are taken from this many rows of the concatenation.  Unless assigned, the
        library's threshold: CsDebug.cfg_split_min_rows (CS_CFG_SPLIT_MIN_ROWS, default 65536) -- the number the native
        driver reads too."""
        return self._split_min_rows if self._split_min_rows is not None else int(L.debug().cfg_split_min_rows)

    @split_min_rows.setter
    def split_min_rows(self, v) -> None:
        self._split_min_rows = None if v is None else int(v)

    # ---- nn.Module-like surface -----------------------------------------------------------
    def state_dict(self) -> "OrderedDict[str, Tensor]":
        return OrderedDict((k, self._sd[k]) for k in self.shapes if k in self._sd)

    def load_state_dict(self, sd, strict: bool = True):
        missing = [k for k in self.shapes if k not in sd]
        unexpected = [k for k in sd if k not in self.shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"DiffusionUNet.load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for k, shp in self.shapes.items():
            if k in sd:
                t = sd[k]
                if tuple(t.shape) != tuple(shp):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {shp}")
                self._sd[k] = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._packed = None
        return self

    def parameters(self):
        return iter(self._sd.values())

    def to(self, device):
        self.device = torch.device(device)
        self._sd = {k: v.to(self.device) for k, v in self._sd.items()}
        self._packed = None
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("the MI355X-native UNet is an inference (sampler) implementation")
        return self

    def set_math(self, mode) -> "DiffusionUNet":
        """GEMM numerics: 'fp32' (fp32-input MFMA, bit-equal to an fp32 fma chain) or 'f16x3' (fp32 operands
        carried as fp16 hi/lo pairs on the fp16 MFMA, ~2^-22 per product; csrc/cs_gemm_f16x3.hip)."""
        m = {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(mode, mode)
        if m not in (L.MATH_FP32, L.MATH_F16X3):
            raise ValueError(f"unknown math mode {mode!r}")
        if m != self.math:
            self.math = m
            self._packed = None
        return self

    def set_attention_math(self, mode) -> "DiffusionUNet":
        """None / 'same': attention follows set_math().  'f16': self-attention on plain fp16 operands (one MFMA pass,
        fp32 softmax / accumulate) -- the "fp16 MFMA attention" option of BASELINE configs[4]; reduced precision,
        outside the fp32 parity gates."""
        self.attn_math = {None: None, "same": None, "f16": L.MATH_F16}[mode]
        return self

    def reset_run_cache(self) -> None:
        """Drop per-sampling-run caches (the one-token context vectors).  The cache key is the context tensor's
        identity + autograd version, which in-place writes through raw pointers (C-ABI kernels, collectives
        refilling a reused buffer) do not bump: rel2shape calls this at the start of every sampler run."""
        self._ctx_cache = None

    def check_overflow(self) -> None:
        """Raise CsOverflowError if an F16X3 kernel on this device met an activation beyond the fp16 range since the
        last check (one 4-byte read-back; not called inside the sampling loop)."""
        ops.check_overflow(self.device, "DiffusionUNet")

    def num_parameters(self) -> int:
        n = 0
        for s in self.shapes.values():
            k = 1
            for v in s:
                k *= v
            n += k
        return n

    # ---- weight packing -------------------------------------------------------------------
    @property
    def _up(self):
        """dims == 3: Upsample doubles the inner two dims only (openai_model_3d.py:150-153); dims == 4: all three."""
        return (0, 1, 1) if self.cfg["dims"] == 3 else (1, 1, 1)

    def _pack(self):
        if self.device.type != "cuda":
            raise L.CsError("DiffusionUNet: weights must be on the HIP device (no CPU path)")
        sd, P = self._sd, self.prefix
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise RuntimeError(f"DiffusionUNet: weights not loaded ({len(missing)} tensors missing)")
        pk: Dict[str, object] = {}

        def pw(p, cin_pad=None, fold_up=None):
            pk[p] = ops.pack_weight(sd[p + ".weight"], sd.get(p + ".bias"), cin_pad=cin_pad, math=self.math,
                                    fold_up=fold_up)

        pw(P + "time_embed.0")
        pw(P + "time_embed.2")
        inp, mid, out, ch = unet_blocks(self.cfg)

        def pack_block(bp, layers):
            for l in layers:
                p = f"{bp}.{l['idx']}"
                k = l["kind"]
                if k == "conv_in":
                    pw(p, cin_pad=(l["cin"] + 3) // 4 * 4)
                elif k == "res":
                    pw(p + ".in_layers.2")
                    pw(p + ".out_layers.3")
                    # r5: the Winograd-W pack beside the direct one (openai_model_3d.py:294-314 convs; dims = 3 family only:
                    # the concat family's volumes are 16^3 -> 8^3 -> 4^3 with W / 2 down to 2 -- eligible too, the rule decides)
                    for nm in (".in_layers.2", ".out_layers.3"):
                        if self.math == L.MATH_F16X3:
                            ops.pack_weight_wino(pk[p + nm], sd[p + nm + ".weight"])
                    if l["cin"] != l["cout"]:
                        pw(p + ".skip_connection")
                elif k == "attn" and not self.cfg["use_spatial_transformer"]:
                    # QKVAttentionLegacy (openai_model_3d.py:388-416) reads qkv channels as [head][q|k|v][ch];
                    # permuting the Conv1d's output rows to [q|k|v][head][ch] makes q, k, v plain column slices
                    c, heads = l["cin"], self.cfg["num_heads"]
                    perm = torch.arange(3 * c, device=self.device).view(heads, 3, c // heads).permute(1, 0, 2).reshape(-1)
                    pk[p + ".qkv"] = ops.pack_weight(sd[p + ".qkv.weight"].view(3 * c, c)[perm].contiguous(),
                                                     sd[p + ".qkv.bias"][perm].contiguous(), math=self.math)
                    pk[p + ".proj_out"] = ops.pack_weight(sd[p + ".proj_out.weight"].view(c, c),
                                                          sd[p + ".proj_out.bias"], math=self.math)
                elif k == "attn":
                    pw(p + ".proj_in")
                    pw(p + ".proj_out")
                    t = p + ".transformer_blocks.0"
                    wqkv = torch.cat([sd[f"{t}.attn1.to_q.weight"], sd[f"{t}.attn1.to_k.weight"],
                                      sd[f"{t}.attn1.to_v.weight"]], dim=0)
                    pk[t + ".attn1.qkv"] = ops.pack_weight(wqkv, math=self.math)
                    pw(t + ".attn1.to_out.0")
                    pw(t + ".attn2.to_q")
                    pw(t + ".attn2.to_k")
                    pw(t + ".attn2.to_v")
                    pw(t + ".attn2.to_out.0")
                    if (self.math == L.MATH_F16X3 and (4 * l["cin"]) % 112 == 0
                            and not L.debug().no_fused_geglu):
                        pk[t + ".ff.geglu"] = ops.pack_geglu_weight(sd[t + ".ff.net.0.proj.weight"],
                                                                    sd[t + ".ff.net.0.proj.bias"])
                    else:
                        pw(t + ".ff.net.0.proj")
                    pw(t + ".ff.net.2")
                elif k == "down":
                    pw(p + ".op")
                elif k == "up":        # Upsample's conv runs on the source grid with per-parity pre-summed taps
                    pw(p + ".conv", fold_up=self._up)

        for i, layers in enumerate(inp):
            pack_block(f"{P}input_blocks.{i}", layers)
        pack_block(P + "middle_block", mid)
        for i, layers in enumerate(out):
            pack_block(f"{P}output_blocks.{i}", layers)
        # the eps head (openai_model_3d.py:733-737: 3x3x3 conv to out_channels = 3): taps as columns, see ops.py
        if ops.tapcol_ok(sd[P + "out.2.weight"], self.math):
            pk[P + "out.2"] = ops.pack_weight_tapcol(sd[P + "out.2.weight"], sd.get(P + "out.2.bias"))
        else:
            pw(P + "out.2")
        # Channel-split ResBlocks (r3).  Output block j concatenates [h | skip_j]; for the blocks whose skip comes from
        # the CONTEXT-FREE PREFIX of the input path (everything before the first attention block: conv_in, two
        # ResBlocks, the first Downsample) that skip is the same tensor for both classifier-free-guidance halves
        # (samplers/ddim.py:206-209 duplicates x).  The part of in_layers' conv / skip_connection that reads whole
        # GroupNorm groups of skip channels is therefore the same for both halves: it is evaluated as its own GEMM
        # (K = the shared channels, at batch B under forward_cfg) and enters the other part's epilogue as a residual.
        # The split point Ks is the first multiple of 16 channels from which every GroupNorm group lies inside the
        # skip (672 = 448 + 224 channels in 21-channel groups: Ks = 464; 448 = 224 + 224: Ks = 224).  The same
        # decomposition runs without guidance pairs (then at the full batch), so per-sample results do not depend on it.
        # It is taken when the concatenation has at least `split_min_rows` rows (CS_CFG_SPLIT_MIN_ROWS, default 65536:
        # >= 8 objects at the 16^3 level, >= 32 at 16x8x8) -- below that the extra launches cost more than the saved
        # multiply-adds (r3, one box: 1 object 7.47 -> 7.80 ms/step, 7 objects 24.8 -> 25.0, 32 objects 85.2 -> 83.0);
        # the unsplit weights stay packed for the small-batch route.
        self._split_info = {}
        n_prefix = next((i for i, layers in enumerate(inp) if any(l["kind"] == "attn" for l in layers)), len(inp))
        if self.cfg["use_spatial_transformer"] and not L.debug().no_cfg_split:
            for j, layers in enumerate(out):
                src = len(inp) - 1 - j
                l = layers[0]
                if src >= n_prefix or l["kind"] != "res":
                    continue
                ch_s = inp[src][-1]["cout"]
                C = l["cin"]
                ch_h, cpg = C - ch_s, C // 32
                ks = next((k for k in range((ch_h + 15) // 16 * 16, C, 16) if (k // cpg) * cpg >= ch_h), None)
                if ks is None or C % 32 or (C - ks) % 16 or l["cin"] == l["cout"]:
                    continue
                q = f"{P}output_blocks.{j}.0"
                for name in (".in_layers.2",):      # (skip_connection stays whole: splitting the 1x1x1 conv -- K = 224 /
                    # 208 / 464 -- measured slower than the unsplit one, its pieces run at 115-190 TF/s: r03_c_gemm_table)
                    wfull = sd[q + name + ".weight"]
                    am = float(wfull.abs().max().item())
                    pk[q + name + ":h"] = ops.pack_weight(wfull[:, :ks].contiguous(), sd[q + name + ".bias"],
                                                          math=self.math, amax=am)
                    pk[q + name + ":s"] = ops.pack_weight(wfull[:, ks:].contiguous(), None, math=self.math, amax=am)
                    if self.math == L.MATH_F16X3:     # r5: the halves' Winograd-W packs, scaled by the whole tensor's maximum
                        ops.pack_weight_wino(pk[q + name + ":h"], wfull[:, :ks].contiguous(), amax=am)
                        ops.pack_weight_wino(pk[q + name + ":s"], wfull[:, ks:].contiguous(), amax=am)
                self._split_info[q] = (ks, ch_h)
        # all 17 ResBlock `emb_layers` Linears read the same SiLU(emb): one GEMM [B,896] x [896, sum(cout)]
        # instead of 17 launch-bound ones; each ResBlock takes its column slice as the conv's row vector.
        names, off, ws, bs = [], 0, [], []
        self._emb_slices = {}
        for bp, layers in ([(f"{P}input_blocks.{i}", l) for i, l in enumerate(inp)] + [(P + "middle_block", mid)]
                           + [(f"{P}output_blocks.{i}", l) for i, l in enumerate(out)]):
            for l in layers:
                if l["kind"] == "res":
                    q = f"{bp}.{l['idx']}"
                    ws.append(sd[q + ".emb_layers.1.weight"])
                    bs.append(sd[q + ".emb_layers.1.bias"])
                    self._emb_slices[q] = (off, off + l["cout"])
                    off += l["cout"]
        pk["emb_all"] = ops.pack_weight(torch.cat(ws, dim=0), torch.cat(bs, dim=0), math=self.math)
        # |gamma|, |beta| maxima of every normalisation layer: with them a norm-fed GEMM's F16X3 operand scale is derived
        # from the producer's bound instead of guessed (ops.norm_a_scale).  One reduction + one read-back, at load time.
        norms = [k[:-7] for k in self.shapes if k.endswith(".weight") and len(self.shapes[k]) == 1]
        if self.math == L.MATH_F16X3 and norms:
            mx = torch.stack([torch.stack([sd[n + ".weight"].abs().max(), sd[n + ".bias"].abs().max()]) for n in norms]).cpu()
            self._ngb = {n: (float(mx[i, 0]), float(mx[i, 1])) for i, n in enumerate(norms)}
        else:
            self._ngb = {}
        # r5: what bounds the operands BORN INSIDE a transformer block (attention.py:237-245) -- q / k /
        # v, the attention output, the GEGLU product, t2 -- from the weights alone.  Per block: max row 2-norms of the
        # Linear weights (a row's output is <= ||row||_2 ||input||_2), max |bias|, and the LayerNorms' max |gamma| and
        # ||beta||_2 (||LN(x)||_2 <= max|gamma| sqrt(C) + ||beta||_2: a normalised token has 2-norm sqrt(C)).  One stacked
        # reduction + one read-back at load time; _static_scales() turns them into operand scales.
        self._tstat: Dict[str, object] = {}
        if self.math == L.MATH_F16X3 and self.cfg["use_spatial_transformer"]:
            lib = L.load()
            jobs = []                          # (block, field, tensor as [rows, cols], which of {row norm, abs max})
            for bp, layers in ([(f"{P}input_blocks.{i}", l) for i, l in enumerate(inp)] + [(P + "middle_block", mid)]
                               + [(f"{P}output_blocks.{i}", l) for i, l in enumerate(out)]):
                for l in layers:
                    if l["kind"] != "attn":
                        continue
                    p = f"{bp}.{l['idx']}"
                    t = p + ".transformer_blocks.0"
                    wff, bff = sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"]
                    h = wff.shape[0] // 2
                    row = lambda v: v.reshape(1, -1)
                    mat = lambda v: v.reshape(v.shape[0], -1)
                    for fld, ten, which in (
                            ("rq", mat(sd[t + ".attn1.to_q.weight"]), 0), ("rk", mat(sd[t + ".attn1.to_k.weight"]), 0),
                            ("rv", mat(sd[t + ".attn1.to_v.weight"]), 0), ("ro", mat(sd[t + ".attn1.to_out.0.weight"]), 0),
                            ("bo", row(sd[t + ".attn1.to_out.0.bias"]), 1), ("rx", mat(wff[:h]), 0), ("bx", row(bff[:h]), 1),
                            ("rg", mat(wff[h:]), 0), ("bg", row(bff[h:]), 1), ("r2", mat(sd[t + ".ff.net.2.weight"]), 0),
                            ("b2", row(sd[t + ".ff.net.2.bias"]), 1), ("rpi", mat(sd[p + ".proj_in.weight"]), 0),
                            ("bpi", row(sd[p + ".proj_in.bias"]), 1), ("g1", row(sd[t + ".norm1.weight"]), 1),
                            ("be1", row(sd[t + ".norm1.bias"]), 0), ("g3", row(sd[t + ".norm3.weight"]), 1),
                            ("be3", row(sd[t + ".norm3.bias"]), 0)):
                        jobs.append((t, fld, ten.contiguous(), which))
            if jobs:
                slots = torch.zeros((len(jobs), 2), dtype=torch.float32, device=self.device)
                for k_, (_, _, ten, _) in enumerate(jobs):        # the SAME kernel the native driver's pack uses: same values
                    L.check(lib.cs_weight_rowstats(ten.data_ptr(), int(ten.shape[0]), int(ten.shape[1]), slots[k_].data_ptr(),
                                                   ops._stream()), "cs_weight_rowstats")
                host = slots.cpu().tolist()
                for (t, fld, _, which), v_ in zip(jobs, host):
                    st = self._tstat.setdefault(t, L.CsTransformerStats())
                    setattr(st, fld, float(v_[which]))
        # r6: the AttentionBlocks' (concat family) static-bound statistics: max row 2-norm of the fused qkv weight, max |bias|
        self._abstat: Dict[str, Tuple[float, float]] = {}
        if self.math == L.MATH_F16X3 and not self.cfg["use_spatial_transformer"]:
            for bp, layers in ([(f"{P}input_blocks.{i}", l) for i, l in enumerate(inp)] + [(P + "middle_block", mid)]
                               + [(f"{P}output_blocks.{i}", l) for i, l in enumerate(out)]):
                for l in layers:
                    if l["kind"] != "attn":
                        continue
                    p = f"{bp}.{l['idx']}"
                    wq, bq = sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]
                    l2, _ = ops.weight_rowstats([wq.reshape(wq.shape[0], -1)])
                    _, bm = ops.weight_rowstats([bq.reshape(1, -1)])
                    self._abstat[p] = (l2, bm)
        self._packed = pk
        self._blocks = (inp, mid, out)
        self._ctx_cache = None

    # ---- forward --------------------------------------------------------------------------
    def _nas(self, norm: str, n: int) -> Optional[float]:
        """F16X3 operand scale for the GEMM fed by normalisation layer `norm` taking its statistics over n elements
        (None in fp32 mode: the default is used and ignored)."""
        gb = self._ngb.get(norm)
        return ops.norm_a_scale(gb[0], gb[1], n) if gb is not None else None

    def _static_scales(self, p: str, t: str, c: int, n_tokens: int, ctx) -> Optional[Dict[str, object]]:
        """F16X3 operand scales of the operands born inside transformer block `t` of SpatialTransformer3D `p` (c channels,
        n_tokens tokens per sample), from static bounds -- so that no input can push them out of the fp16 range and the
        CS_STATUS_F16X3_OVERFLOW detect-and-re-run cliff is gone for them (attention.py:179-245, 335-351):
            Y1 = max|g1| sqrt(c) + ||b1||_2                      >= ||LayerNorm1(x) token||_2
            |q| <= rq Y1, |k| <= rk Y1, |v| <= rv Y1 =: Bv         (to_q / to_k / to_v have no bias)
            |a| <= Bv                                             (softmax rows are convex weights)
            |t0| <= rpi sqrt(c) Egn + |b_pi|,  Egn = max|g| sqrt(n - 1) + max|b| of the block's GroupNorm (n = n_tokens c/32)
            |t1| <= ro sqrt(c) Bv + |b_o| + |t0| + max|ctx vector|  (t1 = to_out(a) + t0 + attn2's row vector)
            |x part|, |gate| of the GEGLU <= rx Y3 + |b|;  |gg| <= Bx Bg   (|gelu(g)| <= |g|)
            |t2| <= r2 sqrt(4c) Bgg + |b_2| + |t1|
        The 2-norm chains overshoot by one to three orders of magnitude; with a power-of-two scale 65000 / bound an operand's
        absolute floor is 2^-25 bound / 65000 ~ 5e-13 bound -- fp32 grade for values down to 1e-5 of the bound.  Returns None
        when the feature is off, the model is not in F16X3 mode or the context is not the one-token form."""
        st = self._tstat.get(t)
        if st is None or self.math != L.MATH_F16X3 or not ops._sw("STATIC_SCALES") or not isinstance(ctx, tuple):
            return None
        key = (t, n_tokens, id(ctx))
        cache = self.__dict__.setdefault("_sscache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        gb = self._ngb.get(p + ".norm", (1.0, 0.0))
        cmax = float(ctx[3].get(t, 0.0)) if len(ctx) > 3 else 0.0
        o = (C.c_float * 12)()
        L.check(L.load().cs_transformer_static_scales(C.byref(st), int(c), int(n_tokens), int(self.cfg["num_heads"]),
                                                      float(gb[0]), float(gb[1]), cmax, o), "cs_transformer_static_scales")
        out = dict(attn=(float(o[0]), float(o[1]), float(o[2])), a=float(o[3]), gg=float(o[4]), t2=float(o[5]),
                   bounds=dict(q=float(o[6]), k=float(o[7]), v=float(o[8]), t1=float(o[9]), gg=float(o[10]), t2=float(o[11]),
                               ctx=cmax))
        cache[key] = out
        return out

    def _slot(self):
        """a fresh 1-element slot of this forward's arena (zeroed once in forward_ndhwc) for the magnitude bound of a RAW
        residual-stream tensor (ops.range_bound / groupnorm(bound=)); None when the feature is off"""
        a = getattr(self, "_amax", None)
        if a is None or self._amax_i >= a.numel():
            return None
        self._amax_i += 1
        return a[self._amax_i - 1:self._amax_i]

    def _res(self, p: str, l: dict, x: Tensor, semb: Tensor, out_fn=None) -> Tensor:
        sd, pk = self._sd, self._packed
        nb = x.shape[0]
        rows = x.shape[1] * x.shape[2] * x.shape[3]
        # GN output goes straight to a conv: emit its fp16 hi/lo operand form where that conv runs the slab kernel
        s1 = self._nas(p + ".in_layers.0", rows * (l["cin"] // 32))
        # (bound=: the skip conv below reads x RAW -- the GroupNorm's finalize kernel leaves x's magnitude bound on the way)
        # r5: ... or its Winograd-W operand where the conv takes that route (cs_conv_wino_ok: large batches)
        vol = (nb, int(x.shape[1]), int(x.shape[2]), int(x.shape[3]))
        hn = ops.groupnorm(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], 32, 1e-5, L.ACT_SILU,
                           split16=ops.wants_split16(nb * rows, pk[p + ".in_layers.2"]), a_scale=s1,
                           bound=self._slot() if l["cin"] != l["cout"] else None,
                           wino=ops.wants_wino(*vol, pk[p + ".in_layers.2"]))
        lo, hi = self._emb_slices[p]
        embo = semb[:, lo:hi]                         # slice of the batched emb projection (row stride = total)
        # (stats=True: the conv's epilogue leaves the partial sums the next GroupNorm takes its statistics from, r4)
        h = ops.conv_gemm(hn, pk[p + ".in_layers.2"], rowvec=embo, rv_rows=rows, math=self.math, a_scale=s1, stats=True)
        s2 = self._nas(p + ".out_layers.0", rows * (l["cout"] // 32))
        hn2 = ops.groupnorm(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], 32, 1e-5, L.ACT_SILU,
                            split16=ops.wants_split16(nb * rows, pk[p + ".out_layers.3"]), a_scale=s2,
                            wino=ops.wants_wino(*vol, pk[p + ".out_layers.3"]))
        # (x_bound: the skip conv reads the RAW residual stream -- its operand scale follows the tensor's actual range)
        skip = x if l["cin"] == l["cout"] else ops.conv_gemm(x, pk[p + ".skip_connection"], math=self.math,
                                                             x_bound=getattr(x, "cs_bound", None))
        return ops.conv_gemm(hn2, pk[p + ".out_layers.3"], res=skip, math=self.math, out_fn=out_fn, a_scale=s2,
                             stats=True)

    def _res_split(self, p: str, l: dict, x: Tensor, skip: Tensor, semb: Tensor, out_fn=None) -> Tensor:
        """ResBlock of an output block whose skip half is shared by the guidance halves (see _pack): x = the
        concatenation [h | skip] at the full batch nb, `skip` = the skip tensor itself at batch nbs (nb / 2 under
        forward_cfg, nb otherwise; a view into x's right half in the latter case).
            in_layers:  conv(SiLU(GN(x)))       = conv_h(GN(x)[:, :Ks]) + conv_s(GN(x)[:, Ks:])    -- conv_s at batch nbs
        (the 1x1x1 skip_connection is left whole: its shared part is too small a GEMM to pay for a second launch)
        GroupNorm statistics are taken over the whole concatenation (one pass, as before); channels >= Ks lie in groups
        made of skip channels only, so sample n and sample n + nbs normalise them identically."""
        sd, pk = self._sd, self._packed
        ks, ch_h = self._split_info[p]
        nb, d, h, w, C = x.shape
        nbs = skip.shape[0]
        rows, cpg, off = d * h * w, C // 32, ks - ch_h
        if nb % nbs or skip.shape[-1] != C - ch_h:
            raise L.CsError("channel-split ResBlock: skip tensor does not match the concatenation")
        gam, bet = sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"]
        wh, ws = pk[p + ".in_layers.2:h"], pk[p + ".in_layers.2:s"]
        stats = ops.groupnorm_stats(x, 32, 1e-5, bound=self._slot())
        s1 = self._nas(p + ".in_layers.0", rows * cpg)
        # r5: where the halves' convs take the Winograd-W route (each launch covers nbs samples) their operands are emitted
        # in that form -- the h half once per guidance half (a position image holds whole launches)
        wn_h, wn_s = ops.wants_wino(nbs, d, h, w, wh), ops.wants_wino(nbs, d, h, w, ws)
        a_h = None if wn_h else ops.groupnorm_apply_range(x[..., :ks], stats, gam[:ks], bet[:ks], cpg, 0, L.ACT_SILU,
                                                          split16=ops.wants_split16(nbs * rows, wh), a_scale=s1)
        a_s = ops.groupnorm_apply_range(skip[..., off:], stats, gam[ks:], bet[ks:], cpg, ks, L.ACT_SILU,
                                        split16=ops.wants_split16(nbs * rows, ws), a_scale=s1, wino=wn_s)
        y_s = ops.conv_gemm(a_s, ws, math=self.math, a_scale=s1)
        lo, hi = self._emb_slices[p]
        cout = l["cout"]
        h1 = torch.empty((nb, d, h, w, cout), dtype=torch.float32, device=x.device)
        for g in range(nb // nbs):                    # one launch per guidance half: both read the shared term
            sl = slice(g * nbs, (g + 1) * nbs)
            a_g = a_h[sl] if a_h is not None else ops.groupnorm_apply_range(
                x[sl][..., :ks], stats[sl], gam[:ks], bet[:ks], cpg, 0, L.ACT_SILU, a_scale=s1, wino=wn_h)
            ops.conv_gemm(a_g, wh, rowvec=semb[sl, lo:hi], rv_rows=rows, res=y_s, out=h1[sl], math=self.math,
                          a_scale=s1)
        # (the two launches write sample ranges of h1: its GroupNorm takes its statistics from a pass over the tensor)
        sk = ops.conv_gemm(x, pk[p + ".skip_connection"], math=self.math, x_bound=getattr(x, "cs_bound", None))
        s2 = self._nas(p + ".out_layers.0", rows * (cout // 32))
        hn2 = ops.groupnorm(h1, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], 32, 1e-5, L.ACT_SILU,
                            split16=ops.wants_split16(nb * rows, pk[p + ".out_layers.3"]), a_scale=s2,
                            wino=ops.wants_wino(nb, d, h, w, pk[p + ".out_layers.3"]))
        return ops.conv_gemm(hn2, pk[p + ".out_layers.3"], res=sk, math=self.math, out_fn=out_fn, a_scale=s2,
                             stats=True)

    def _context_vectors(self, ctx: Tensor):
        """One-token context: every transformer block's cross-attention output is the per-sample
        row vector to_out(to_v(ctx)).  It depends on ctx only, so it is computed once per sampling run: the
        sampler passes the same `[uc; c]` tensor at every step (identity + version counter guard the cache)."""
        if ctx.shape[1] != 1:
            return ctx
        key = (ctx.data_ptr(), ctx._version, tuple(ctx.shape), self.math)
        if self._ctx_cache is not None and self._ctx_cache[0] == key:
            return self._ctx_cache[1]
        pk = self._packed
        nb = ctx.shape[0]
        flat = ctx.reshape(nb, -1)
        vecs = {}
        b_ctx = None
        inp, mid, out = self._blocks
        P = self.prefix
        for bp, layers in ([(f"{P}input_blocks.{i}", l) for i, l in enumerate(inp)] + [(P + "middle_block", mid)]
                           + [(f"{P}output_blocks.{i}", l) for i, l in enumerate(out)]):
            for l in layers:
                if l["kind"] == "attn":
                    t = f"{bp}.{l['idx']}.transformer_blocks.0"
                    # (r5: the context is a RAW input and to_v's output a raw intermediate: their operand scales follow a
                    # device-side magnitude bound -- max |.|, CsConvGemm.a_bound -- not the constant 16; no host read-back)
                    dyn = self.math == L.MATH_F16X3 and ops._sw("DYN_SCALE") and ops._sw("STATIC_SCALES")
                    if dyn and b_ctx is None:
                        b_ctx = flat.abs().max().reshape(1)
                    v2 = ops.linear(flat, pk[t + ".attn2.to_v"], math=self.math, x_bound=b_ctx if dyn else None)
                    vecs[t] = ops.linear(v2, pk[t + ".attn2.to_out.0"], math=self.math,
                                         x_bound=v2.abs().max().reshape(1) if dyn else None)
        # (r5: the largest |row-vector entry| per block -- it enters t1's static bound, _static_scales -- ONE read-back per
        # sampling run, beside the run's one status read-back; never inside the step loop)
        cmax = {}
        if self.math == L.MATH_F16X3 and ops._sw("STATIC_SCALES") and vecs:
            ks = list(vecs)
            host = torch.stack([vecs[k_].abs().max() for k_ in ks]).cpu().tolist()
            cmax = {k_: float(v_) for k_, v_ in zip(ks, host)}
        cached = ("ctxvec", vecs, ctx, cmax)    # keeps ctx alive so its data_ptr cannot be recycled
        self._ctx_cache = (key, cached)
        self.__dict__.pop("_sscache", None)
        return cached

    def _attnblock(self, p: str, l: dict, x: Tensor, out_fn=None) -> Tensor:
        """AttentionBlock._forward (openai_model_3d.py:360-366): GN -> qkv (Conv1d k=1) -> QKVAttentionLegacy ->
        proj_out + x.  The reference scales q and k by ch^-1/4 each; the flash kernel scales the logits by ch^-1/2."""
        sd, pk = self._sd, self._packed
        heads = self.cfg["num_heads"]
        nb, d, h, w, c = x.shape
        n = d * h * w
        xn = ops.groupnorm(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 32, 1e-5, L.ACT_NONE)
        qkv = ops.linear(xn.view(nb, n, c), pk[p + ".qkv"], math=self.math, a_scale=self._nas(p + ".norm", n * (c // 32)))
        # r6: q / k / v = Conv1d(GroupNorm(x)) + bias are bounded by the weights and the norm's affine parameters alone
        # (ops.attnblock_static_scales; cs_unet.hip::attnblock applies the same rule to the same statistics)
        amath = self.attn_math if self.attn_math is not None else self.math
        ss = None
        gb, st = self._ngb.get(p + ".norm"), self.__dict__.get("_abstat", {}).get(p)
        if self.math == L.MATH_F16X3 and gb is not None and st is not None:
            ss = ops.attnblock_static_scales(gb[0], gb[1], n * (c // 32), c, st[0], st[1], (c // heads) ** -0.5)
        a = ops.attention(qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:3 * c], heads, (c // heads) ** -0.5, math=amath,
                          scales=ss[:3] if (ss is not None and amath == L.MATH_F16X3) else None)
        dst = out_fn((nb, d, h, w, c)).view(nb, n, c) if out_fn is not None else None
        # (spatial=: the rows are nb samples of n tokens -- what the epilogue's GroupNorm partial sums are tiled by)
        out = ops.linear(a, pk[p + ".proj_out"], res=x.view(nb, n, c), math=self.math, out=dst, stats=True,
                         spatial=(nb, n, 1, 1), a_scale=ss[3] if ss is not None else None)
        return ops.attach_stats(out.view(nb, d, h, w, c), getattr(out, "cs_stats", None))

    def _attn(self, p: str, l: dict, x: Tensor, ctx, out_fn=None) -> Tensor:
        sd, pk = self._sd, self._packed
        heads = self.cfg["num_heads"]
        nb, d, h, w, c = x.shape
        n = d * h * w
        dh = c // heads
        t = p + ".transformer_blocks.0"
        xn = ops.groupnorm(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 32, 1e-6, L.ACT_NONE)
        t0 = ops.linear(xn.view(nb, n, c), pk[p + ".proj_in"], math=self.math, a_scale=self._nas(p + ".norm", n * (c // 32)))
        # LayerNorm feeding an F16X3 GEMM writes the interleaved operand pair (same bytes as fp32; no split in the K loop)
        s1 = self._nas(t + ".norm1", c)
        n1 = ops.layernorm(t0, sd[t + ".norm1.weight"], sd[t + ".norm1.bias"], pair_scale=s1)
        qkv = ops.linear(n1, pk[t + ".attn1.qkv"], math=self.math, a_scale=s1)
        # r5: static bounds of the operands born inside the block (q / k / v, the attention output, gg, t2): _static_scales
        ss = self._static_scales(p, t, c, n, ctx)
        amath = self.attn_math if self.attn_math is not None else self.math
        a = ops.attention(qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:3 * c], heads, dh ** -0.5, math=amath,
                          scales=ss["attn"] if (ss is not None and amath == L.MATH_F16X3) else None)
        if isinstance(ctx, tuple):
            # one context token: softmax over one key == 1, attn2(x) = to_out(to_v(ctx)) for every
            # query row -> a per-sample row vector folded into the attn1 output GEMM.
            t1 = ops.linear(a, pk[t + ".attn1.to_out.0"], res=t0, rowvec=ctx[1][t], rv_rows=n, math=self.math,
                            a_scale=ss["a"] if ss is not None else None)
        else:
            t1a = ops.linear(a, pk[t + ".attn1.to_out.0"], res=t0, math=self.math)
            n2 = ops.layernorm(t1a, sd[t + ".norm2.weight"], sd[t + ".norm2.bias"])
            q2 = ops.linear(n2, pk[t + ".attn2.to_q"], math=self.math, a_scale=self._nas(t + ".norm2", c))
            k2 = ops.linear(ctx, pk[t + ".attn2.to_k"], math=self.math)
            vv2 = ops.linear(ctx, pk[t + ".attn2.to_v"], math=self.math)
            a2 = ops.attention(q2, k2, vv2, heads, dh ** -0.5, math=self.attn_math if self.attn_math is not None else self.math)
            t1 = ops.linear(a2, pk[t + ".attn2.to_out.0"], res=t1a, math=self.math)
        s3 = self._nas(t + ".norm3", c)
        n3 = ops.layernorm(t1, sd[t + ".norm3.weight"], sd[t + ".norm3.bias"], pair_scale=s3)
        # r4: gg's only reader is ff.net.2 and t2's only reader is proj_out (attention.py:241-245, 349), so both producers
        # write the interleaved F16X3 operand pair where their launch can (out_pair: scale 16, the raw-activation default;
        # a value beyond the fp16 range is now flagged by the producer) -- the consumers' K loops carry no conversion.
        # Bit-identical to the fp32 hand-over (tests/test_epilogue_outputs_gpu.py).
        on = self.math == L.MATH_F16X3 and ops._sw("PAIR_EPILOGUES")
        pair_gg = (ss["gg"] if ss is not None else ops.A_SCALE) if on else None
        pair_t2 = (ss["t2"] if ss is not None else ops.A_SCALE) if on else None
        if (t + ".ff.geglu") in pk:      # GEGLU gate fused into the projection GEMM's epilogue
            gg = ops.linear(n3, pk[t + ".ff.geglu"], act=L.ACT_GEGLU, a_scale=s3, out_pair=pair_gg)   # the library picks a 224-column tile
        else:
            ff = ops.linear(n3, pk[t + ".ff.net.0.proj"], math=self.math, a_scale=s3)
            gg = ops.geglu(ff)
        # (a producer that could not emit the pair hands over fp32: the consumer then takes the same static scale itself)
        t2 = ops.linear(gg, pk[t + ".ff.net.2"], res=t1, math=self.math, out_pair=pair_t2,
                        a_scale=None if isinstance(gg, ops.Pair16) or ss is None else ss["gg"])
        dst = out_fn((nb, d, h, w, c)).view(nb, n, c) if out_fn is not None else None
        out = ops.linear(t2, pk[p + ".proj_out"], res=x.view(nb, n, c), math=self.math, out=dst, stats=True,
                         spatial=(nb, n, 1, 1), a_scale=None if isinstance(t2, ops.Pair16) or ss is None else ss["t2"])
        return ops.attach_stats(out.view(nb, d, h, w, c), getattr(out, "cs_stats", None))

    def _run(self, bp: str, layers, h: Tensor, semb: Tensor, ctx: Tensor, out_fn=None, split_skip=None) -> Tensor:
        """`out_fn(shape) -> tensor`: where the block's LAST layer writes its result (a channel slice of the
        concatenation buffer of the output block that consumes it: torch.cat([h, hs.pop()], 1) then costs no copy)."""
        pk = self._packed
        for li, l in enumerate(layers):
            p = f"{bp}.{l['idx']}"
            k = l["kind"]
            of = out_fn if li == len(layers) - 1 else None
            if k == "conv_in":
                # (x_bound: the RAW latent -- its exact max |.| from forward_ndhwc, so no x_t can leave the fp16 range, r6)
                h = ops.conv_gemm(h, pk[p], math=self.math, out_fn=of, stats=True, x_bound=getattr(h, "cs_bound", None))
            elif k == "res":
                if li == 0 and split_skip is not None and p in self._split_info:
                    h = self._res_split(p, l, h, split_skip, semb, of)
                else:
                    h = self._res(p, l, h, semb, of)
            elif k == "attn":
                h = (self._attn(p, l, h, ctx, of) if self.cfg["use_spatial_transformer"]
                     else self._attnblock(p, l, h, of))
            elif k == "down":      # dims == 3: inner two dims only (openai_model_3d.py:188); dims == 4: all three
                # (Down / Upsample read the RAW stream with no GroupNorm in front: its bound from the producers' partials)
                h = ops.conv_gemm(h, pk[p + ".op"], stride=(1, 2, 2) if self.cfg["dims"] == 3 else (2, 2, 2),
                                  math=self.math, out_fn=of, stats=True, x_bound=ops.range_bound(h, self._slot()))
            elif k == "up":        # nearest x2 folded into the conv's addressing (openai_model_3d.py:148-157)
                h = ops.conv_gemm(h, pk[p + ".conv"], up=self._up, math=self.math, out_fn=of, stats=True,
                                  x_bound=ops.range_bound(h, self._slot()))
        return h

    @torch.no_grad()
    def forward_ndhwc(self, h: Tensor, t: Tensor, ctx: Tensor, cfg_pairs: bool = False) -> Tensor:
        """x as [nb, d, h, w, cpad(4)] channels-last -> eps [nb, d, h, w, out_channels].

        cfg_pairs=True: classifier-free guidance evaluates the SAME (x, t) under two contexts
        (samplers/ddim.py:206-209 builds x_in = cat([x] * 2)).  Everything upstream of the first
        cross-attention depends on (x, t) only, so `h`, `t` are passed once (B samples), ctx holds the 2B
        contexts [uc; c], the prefix blocks run at batch B and their outputs are shared by both halves.
        Per-sample results are bit-identical to running the duplicated batch (rows never mix)."""
        if self._packed is None:
            self._pack()
        sd, pk, P = self._sd, self._packed, self.prefix
        inp, mid, out = self._blocks
        # magnitude-bound slots of this forward (one per raw-stream consumer; one memset): see _slot / ops.range_bound
        self._amax = (torch.zeros(64, dtype=torch.float32, device=h.device)
                      if self.math == L.MATH_F16X3 and ops._sw("DYN_SCALE") and ops._sw("GN_PARTS") else None)
        self._amax_i = 0
        if self._amax is not None:
            ops.absmax_bound(h, self._slot())          # conv_in's operand scale follows the latent itself (ops.absmax_bound)
        temb = ops.timestep_embedding(t, self.cfg["model_channels"])
        e1 = ops.linear(temb, pk[P + "time_embed.0"], act=L.ACT_SILU, math=self.math)
        # every consumer of `emb` is emb_layers = SiLU -> Linear (openai_model_3d.py:257-263): keep SiLU(emb)
        semb = ops.linear(e1, pk[P + "time_embed.2"], act=L.ACT_SILU, math=self.math)
        semb = ops.linear(semb, pk["emb_all"], math=self.math)          # [nb, sum(cout)] for all ResBlocks
        ctx = self._context_vectors(ctx) if ctx is not None else None
        # torch.cat([h, hs.pop()], dim=1) (openai_model_3d.py:781) without copies: output block j reads ONE buffer
        # cats[j] = [h (ch_h) | skip (ch_skip)]; the layer that produces h (middle block / previous output block) and
        # the input block that produces the skip both write straight into their channel slice.  Only the skips of the
        # context-free prefix, which one copy serves for both guidance halves, are still copied (duplicated) in.
        hs: List[Tensor] = []
        tr = self.trace
        nout = len(out)
        ch_skip = [layers[-1]["cout"] for layers in inp][::-1]               # skip channels of output block j
        ch_h = [out[j][0]["cin"] - ch_skip[j] for j in range(nout)]
        cats: List[Optional[Tensor]] = [None] * nout
        nocopy = not L.debug().concat_copy                                   # A/B switch: the copying form
        # r4: the partial sums each half's producer left for the GroupNorm of output block j (ops.ColStats or None)
        split_min_rows = self.split_min_rows
        seg_l: List[Optional[object]] = [None] * nout
        seg_r: List[Optional[object]] = [None] * nout

        def slot(j: int, left: bool):
            def fn(shape):
                want = (*shape[:-1], ch_h[j] + ch_skip[j])
                if cats[j] is None:
                    cats[j] = torch.empty(want, dtype=torch.float32, device=h.device)
                elif tuple(cats[j].shape) != want:
                    raise L.CsError(f"concat buffer {j}: {tuple(cats[j].shape)} vs {want}")
                return cats[j][..., :ch_h[j]] if left else cats[j][..., ch_h[j]:]
            return fn

        shared = cfg_pairs          # True while h still holds one copy per (x, t) pair
        for i, layers in enumerate(inp):
            if shared and any(l["kind"] == "attn" for l in layers):
                st = getattr(h, "cs_stats", None)
                h = torch.cat([h, h], dim=0)          # first context-dependent block: split into [uc; c]
                ops.attach_stats(h, st)               # (sample n reads the partials of sample n % B)
                semb = torch.cat([semb, semb], dim=0)
                shared = False
            direct = nocopy and not shared            # a skip at the full batch goes straight into its slice
            h = self._run(f"{P}input_blocks.{i}", layers, h, semb, ctx, slot(nout - 1 - i, False) if direct else None)
            hs.append(None if direct else h)
            seg_r[nout - 1 - i] = getattr(h, "cs_stats", None)
            if tr is not None:
                tr[f"input_blocks.{i}"] = h
        if shared:
            st = getattr(h, "cs_stats", None)
            h = ops.attach_stats(torch.cat([h, h], dim=0), st)
            semb = torch.cat([semb, semb], dim=0)
        h = self._run(P + "middle_block", mid, h, semb, ctx, slot(0, True) if nocopy else None)
        seg_l[0] = getattr(h, "cs_stats", None)
        if tr is not None:
            tr["middle_block"] = h
        for i, layers in enumerate(out):
            skip = hs.pop()
            if not nocopy:
                h = ops.concat_channels(h, skip)
            else:
                if skip is not None:                  # prefix skip: B-sized, feeds both guidance halves
                    right = slot(i, False)((h.shape[0], *skip.shape[1:]))
                    nbs = skip.shape[0]
                    for g in range(h.shape[0] // nbs):
                        ops.copy_rows(skip, right[g * nbs:(g + 1) * nbs])
                h = cats[i]
                cats[i] = None
                h.cs_segs = [(0, seg_l[i]), (ch_h[i], seg_r[i])]     # both halves' producers' partials (None: unavailable)
            # channel-split blocks take the skip tensor itself: the shared (B-sized) one under guidance pairs, else the
            # right half of the concatenation
            split_skip = None
            if (f"{P}output_blocks.{i}.0" in self._split_info
                    and h.shape[0] * h.shape[1] * h.shape[2] * h.shape[3] >= split_min_rows):
                split_skip = skip if skip is not None else h[..., ch_h[i]:]
            h = self._run(f"{P}output_blocks.{i}", layers, h, semb, ctx,
                          slot(i + 1, True) if nocopy and i + 1 < nout else None, split_skip=split_skip)
            if i + 1 < nout:
                seg_l[i + 1] = getattr(h, "cs_stats", None)
            if tr is not None:
                tr[f"output_blocks.{i}"] = h
        so = self._nas(P + "out.0", h.shape[1] * h.shape[2] * h.shape[3] * (h.shape[4] // 32))
        hn = ops.groupnorm(h, sd[P + "out.0.weight"], sd[P + "out.0.bias"], 32, 1e-5, L.ACT_SILU,
                           split16=ops.wants_split16(h.shape[0] * h.shape[1] * h.shape[2] * h.shape[3], pk[P + "out.2"]),
                           a_scale=so)
        return ops.conv_gemm(hn, pk[P + "out.2"], math=self.math, a_scale=so)

    @torch.no_grad()
    def forward_cfg(self, x: Tensor, t: Tensor, c_in: Tensor) -> Tensor:
        """eps for the classifier-free-guidance pair batch without duplicating (x, t):
        x (B,C,D,H,W), t (B,), c_in (2B,1,ctx) = [uc; c]  ->  (2B,C,D,H,W) = [eps_uc; eps_c]."""
        if self.conditioning_key == "concat":
            # the condition is an input channel: nothing upstream of it to share, run the duplicated batch
            return self.forward(torch.cat([x, x]), torch.cat([t, t]), c_concat=[c_in])
        if self.conditioning_key != "crossattn":
            raise NotImplementedError("forward_cfg: crossattn / concat conditioning only")
        ctx = c_in.to(dtype=torch.float32).contiguous()
        if ctx.shape[0] != 2 * x.shape[0]:
            raise ValueError("c_in must hold [uc; c] for every sample of x")
        h = ops.nchw_to_ndhwc(x.to(torch.float32), cpad=(self.cfg["in_channels"] + 3) // 4 * 4)
        eps = self.forward_ndhwc(h, t.to(torch.int64).contiguous(), ctx, cfg_pairs=True)
        return ops.ndhwc_to_nchw(eps)

    @torch.no_grad()
    def forward(self, x: Tensor, t: Tensor, c_concat: Optional[list] = None,
                c_crossattn: Optional[list] = None) -> Tensor:
        """network.py:20-42.  x: (B, C, D, H, W) fp32 on the HIP device; t: (B,) int64."""
        if t.dtype != torch.int64:
            t = t.to(torch.int64)
        cpad = (self.cfg["in_channels"] + 3) // 4 * 4
        if self.conditioning_key == "concat":          # network.py:25-27
            if not c_concat:
                raise ValueError("c_concat is required for conditioning_key='concat'")
            if self.cfg["use_spatial_transformer"]:
                raise NotImplementedError("concat conditioning with SpatialTransformer3D blocks (needs a context)")
            xc = torch.cat([x.to(torch.float32)] + [c.to(device=x.device, dtype=torch.float32) for c in c_concat], dim=1)
            if xc.shape[1] != self.cfg["in_channels"]:
                raise ValueError(f"x + c_concat have {xc.shape[1]} channels, in_channels={self.cfg['in_channels']}")
            return ops.ndhwc_to_nchw(self.forward_ndhwc(ops.nchw_to_ndhwc(xc, cpad=cpad), t.contiguous(), None))
        if self.conditioning_key != "crossattn":
            raise NotImplementedError(f"conditioning_key={self.conditioning_key!r}: 'crossattn' "
                                      "(config/sdfusion-txt2shape.yaml:5) and 'concat' "
                                      "(config/sdfusion-txt2shape_concat.yaml:5) are implemented")
        if not self.cfg["use_spatial_transformer"]:
            raise NotImplementedError("crossattn conditioning needs use_spatial_transformer=True")
        if c_crossattn is None:
            raise ValueError("c_crossattn is required for conditioning_key='crossattn'")
        ctx = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        ctx = ctx.to(dtype=torch.float32).contiguous()
        if t.dtype != torch.int64:
            t = t.to(torch.int64)
        h = ops.nchw_to_ndhwc(x.to(torch.float32), cpad=(self.cfg["in_channels"] + 3) // 4 * 4)
        eps = self.forward_ndhwc(h, t.contiguous(), ctx)
        return ops.ndhwc_to_nchw(eps)

    __call__ = forward

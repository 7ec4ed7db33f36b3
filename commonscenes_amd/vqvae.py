"""MI355X-native VQ-VAE decode side behind the reference's `VQVAE` interface.

Mirrors  model/networks/vqvae_networks/network.py:48-103   VQVAE.decode / decode_no_quant
         model/networks/vqvae_networks/quantizer.py:68-119 VectorQuantizer.forward (nearest code)
         model/networks/vqvae_networks/vqvae_modules.py:292-409 Decoder3D (+ ResnetBlock, AttnBlock, Upsample)
         model/model_utils.py:7-31 load_vqvae
Channels-last throughout; nearest x2 upsampling is folded into the following conv's address
arithmetic (no 8x larger intermediate), swish/GELU are fused into the GroupNorm apply pass, the
single-head N=4096 attention is the flash kernel (no 64 MiB score matrix), and the codebook search
runs out of LDS (no 134 MB distance matrix).  Only the decode side exists (the encoder is
training-only).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from . import lib as L
from . import ops

Tensor = torch.Tensor


def _vq_groups(c: int) -> int:          # vqvae_modules.py:13-21
    if c <= 32:
        return c // 4
    if c % 32 != 0:
        return 30
    return 32


def _dd(ddconfig) -> dict:
    g = (lambda k, d=None: ddconfig.get(k, d)) if isinstance(ddconfig, dict) else (
        lambda k, d=None: getattr(ddconfig, k, d))
    cfg = dict(ch=int(g("ch")), out_ch=int(g("out_ch")), ch_mult=tuple(g("ch_mult")),
               num_res_blocks=int(g("num_res_blocks")), z_channels=int(g("z_channels")),
               resolution=int(g("resolution")), attn_resolutions=tuple(g("attn_resolutions", ()) or ()))
    if cfg["attn_resolutions"]:
        raise NotImplementedError("attn_resolutions is empty in config/vqvae_snet.yaml; per-level attention "
                                  "is not on the path")
    return cfg


def vqvae_param_shapes(ddconfig, n_embed: int, embed_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """Decode-side state_dict entries of the reference VQVAE."""
    cfg = _dd(ddconfig)
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(p, o, i, k):
        S[p + ".weight"] = (o, i, k, k, k)
        S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,)
        S[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    ch, mult = cfg["ch"], cfg["ch_mult"]
    nres = len(mult)
    block_in = ch * mult[-1]
    D = "decoder."
    conv(D + "conv_in", block_in, cfg["z_channels"], 3)
    res(D + "mid.block_1", block_in, block_in)
    norm(D + "mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv(D + f"mid.attn_1.{n}", block_in, block_in, 1)
    res(D + "mid.block_2", block_in, block_in)
    for i_level in reversed(range(nres)):
        block_out = ch * mult[i_level]
        for i_block in range(cfg["num_res_blocks"]):
            res(f"{D}up.{i_level}.block.{i_block}", block_in, block_out)
            block_in = block_out
        if i_level != 0:
            conv(f"{D}up.{i_level}.upsample.conv", block_in, block_in, 3)
    norm(D + "norm_out", block_in)
    conv(D + "conv_out", cfg["out_ch"], block_in, 3)
    S["quantize.embedding.weight"] = (n_embed, embed_dim)
    conv("post_quant_conv", cfg["z_channels"], embed_dim, 1)
    return S


class VQVAE:
    """Decode-side drop-in for reference `VQVAE` (network.py:48-103)."""

    def __init__(self, ddconfig, n_embed: int, embed_dim: int, device: str | torch.device = "cuda"):
        self.cfg = _dd(ddconfig)
        self.ddconfig = ddconfig
        self.n_embed, self.embed_dim = int(n_embed), int(embed_dim)
        self.device = torch.device(device)
        self.shapes = vqvae_param_shapes(ddconfig, n_embed, embed_dim)
        self._sd: Dict[str, Tensor] = {}
        self._packed = None
        self.math = L.DEFAULT_MATH      # F16X3 unless CS_MATH=fp32
        self.last_indices: Optional[Tensor] = None

    # ---- nn.Module-like surface ----
    def state_dict(self):
        return OrderedDict((k, self._sd[k]) for k in self.shapes if k in self._sd)

    def load_state_dict(self, sd, strict: bool = True):
        """Accepts a full reference VQVAE state_dict; encoder.* / quant_conv.* are training-only and ignored."""
        missing = [k for k in self.shapes if k not in sd]
        unexpected = [k for k in sd if k not in self.shapes and not k.startswith(("encoder.", "quant_conv."))]
        if strict and (missing or unexpected):
            raise RuntimeError(f"VQVAE.load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for k, shp in self.shapes.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {shp}")
                self._sd[k] = sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
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
        return self

    def set_math(self, mode) -> "VQVAE":
        """'fp32' or 'f16x3' GEMM numerics (see DiffusionUNet.set_math)."""
        m = {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(mode, mode)
        if m not in (L.MATH_FP32, L.MATH_F16X3):
            raise ValueError(f"unknown math mode {mode!r}")
        if m != self.math:
            self.math = m
            self._packed = None
        return self

    # ---- packing ----
    def _pack(self):
        sd = self._sd
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise RuntimeError(f"VQVAE: weights not loaded ({len(missing)} tensors missing)")
        pk = {}
        for k in self.shapes:
            if k.endswith(".weight") and sd[k].dim() == 5:
                p = k[:-7]
                if p.startswith("decoder.mid.attn_1.") and p.split(".")[-1] in ("q", "k", "v"):
                    continue
                if p == "post_quant_conv":
                    # emit a zero 4th channel so the decoder's conv_in reads float4-aligned rows
                    wz, bz = sd[k], sd[p + ".bias"]
                    pad = (-wz.shape[0]) % 4
                    wz = torch.cat([wz, wz.new_zeros((pad, *wz.shape[1:]))], dim=0)
                    bz = torch.cat([bz, bz.new_zeros(pad)], dim=0)
                    pk[p] = ops.pack_weight(wz, bz, cin_pad=(wz.shape[1] + 3) // 4 * 4, math=self.math)
                    continue
                cin = sd[k].shape[1]
                if p == "decoder.conv_out" and ops.tapcol_ok(sd[k], self.math):
                    # vqvae_modules.py:473 (3x3x3 conv to out_ch = 1): taps as columns, see ops.py
                    pk[p] = ops.pack_weight_tapcol(sd[k], sd.get(p + ".bias"))
                    continue
                # Upsample's conv (vqvae_modules.py:35-39: nearest x2 in D, H, W) runs on the source grid
                fold = (1, 1, 1) if p.endswith(".upsample.conv") else None
                pk[p] = ops.pack_weight(sd[k], sd.get(p + ".bias"), cin_pad=(cin + 3) // 4 * 4, math=self.math,
                                        fold_up=fold)
                if fold is None and self.math == L.MATH_F16X3:      # r5: + the Winograd-W pack where the geometry allows
                    ops.pack_weight_wino(pk[p], sd[k])
        a = "decoder.mid.attn_1."
        wqkv = torch.cat([sd[a + "q.weight"], sd[a + "k.weight"], sd[a + "v.weight"]], dim=0)
        bqkv = torch.cat([sd[a + "q.bias"], sd[a + "k.bias"], sd[a + "v.bias"]], dim=0)
        pk[a + "qkv"] = ops.pack_weight(wqkv, bqkv, math=self.math)
        # r6: what bounds the attention block's q / k / v (ops.attnblock_static_scales; cs_vqvae_pack takes the same statistics)
        self._attn_stat = None
        if self.math == L.MATH_F16X3:
            c_ = int(sd[a + "q.weight"].shape[0])
            l2, _ = ops.weight_rowstats([sd[a + f"{n}.weight"].reshape(c_, -1) for n in ("q", "k", "v")])
            _, bm = ops.weight_rowstats([sd[a + f"{n}.bias"].reshape(1, -1) for n in ("q", "k", "v")])
            self._attn_stat = (l2, bm)
        # |gamma|, |beta| maxima of the Normalize layers: norm-fed GEMMs take their F16X3 operand scale from the
        # producer's bound (ops.norm_a_scale), one read-back at load time
        norms = [k[:-7] for k in sd if k.startswith("decoder.") and k.endswith(".weight") and sd[k].dim() == 1]
        if self.math == L.MATH_F16X3 and norms:
            mx = torch.stack([torch.stack([sd[n + ".weight"].abs().max(), sd[n + ".bias"].abs().max()]) for n in norms]).cpu()
            self._ngb = {n: (float(mx[i, 0]), float(mx[i, 1])) for i, n in enumerate(norms)}
        else:
            self._ngb = {}
        self._packed = pk

    # ---- building blocks ----
    def _nas(self, norm: str, n: int):
        gb = getattr(self, "_ngb", {}).get(norm)
        return ops.norm_a_scale(gb[0], gb[1], n) if gb is not None else None

    def _res(self, p: str, x: Tensor) -> Tensor:
        sd, pk = self._sd, self._packed
        c = x.shape[-1]
        m = x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3]
        rows = m // x.shape[0]
        s1 = self._nas(p + ".norm1", rows * (c // _vq_groups(c)))
        vol = tuple(int(v) for v in x.shape[:4])
        # (r5: the Winograd-W operand where the conv takes that route -- decided per sample geometry, never by the batch)
        h = ops.groupnorm(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], _vq_groups(c), 1e-6, L.ACT_SILU,
                          split16=ops.wants_split16(m, pk[p + ".conv1"]), a_scale=s1,
                          wino=ops.wants_wino(*vol, pk[p + ".conv1"]))
        # (stats="invariant", r5: the conv's epilogue leaves the partial sums norm2 takes its statistics from -- only where
        # the statistics tiles are the same for one object and for a slice of sixteen, see ops._epilogue_extras)
        h = ops.conv_gemm(h, pk[p + ".conv1"], math=self.math, a_scale=s1, stats="invariant")
        co = h.shape[-1]
        s2 = self._nas(p + ".norm2", rows * (co // _vq_groups(co)))
        h = ops.groupnorm(h, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], _vq_groups(co), 1e-6, L.ACT_SILU,
                          split16=ops.wants_split16(m, pk[p + ".conv2"]), a_scale=s2,
                          wino=ops.wants_wino(*vol, pk[p + ".conv2"]))
        skip = x if (p + ".nin_shortcut") not in pk else ops.conv_gemm(x, pk[p + ".nin_shortcut"], math=self.math)
        return ops.conv_gemm(h, pk[p + ".conv2"], res=skip, math=self.math, a_scale=s2, stats="invariant")

    def _attn(self, p: str, x: Tensor) -> Tensor:
        sd, pk = self._sd, self._packed
        nb, d, h, w, c = x.shape
        n = d * h * w
        hn = ops.groupnorm(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], _vq_groups(c), 1e-6, L.ACT_NONE)
        qkv = ops.linear(hn.view(nb, n, c), pk[p + ".qkv"], math=self.math,
                         a_scale=self._nas(p + ".norm", n * (c // _vq_groups(c))))
        # r6: q / k / v = Conv1x1(Normalize(x)) + bias are bounded by the weights and the norm's affine parameters alone: static
        # operand scales (no input can leave the fp16 range; the attention output is a convex combination of v rows)
        ss = None
        gb = getattr(self, "_ngb", {}).get(p + ".norm")
        st = getattr(self, "_attn_stat", None)
        if self.math == L.MATH_F16X3 and gb is not None and st is not None:
            ss = ops.attnblock_static_scales(gb[0], gb[1], n * (c // _vq_groups(c)), c, st[0], st[1], int(c) ** (-0.5))
        a = ops.attention(qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:], 1, int(c) ** (-0.5), math=self.math,
                          scales=ss[:3] if ss is not None else None)
        out = ops.linear(a, pk[p + ".proj_out"], res=x.view(nb, n, c), math=self.math,
                         a_scale=ss[3] if ss is not None else None)
        return out.view(nb, d, h, w, c)

    @torch.no_grad()
    def decoder_ndhwc(self, z: Tensor) -> Tensor:
        """Decoder3D.forward (vqvae_modules.py:376-409) on [nb,d,h,w,4] -> [nb,4d,4h,4w,out_ch]."""
        sd, pk = self._sd, self._packed
        D = "decoder."
        nres = len(self.cfg["ch_mult"])
        h = ops.conv_gemm(z, pk[D + "conv_in"], math=self.math, stats="invariant")
        h = self._res(D + "mid.block_1", h)
        h = self._attn(D + "mid.attn_1", h)
        h = self._res(D + "mid.block_2", h)
        for i_level in reversed(range(nres)):
            for i_block in range(self.cfg["num_res_blocks"]):
                h = self._res(f"{D}up.{i_level}.block.{i_block}", h)
            if i_level != 0:
                h = ops.conv_gemm(h, pk[f"{D}up.{i_level}.upsample.conv"], up=(1, 1, 1), math=self.math)
        c = h.shape[-1]
        so = self._nas(D + "norm_out", h.shape[1] * h.shape[2] * h.shape[3] * (c // _vq_groups(c)))
        h = ops.groupnorm(h, sd[D + "norm_out.weight"], sd[D + "norm_out.bias"], _vq_groups(c), 1e-6, L.ACT_GELU,
                          split16=ops.wants_split16(h.shape[0] * h.shape[1] * h.shape[2] * h.shape[3], pk[D + "conv_out"]),
                          a_scale=so)
        return ops.conv_gemm(h, pk[D + "conv_out"], math=self.math, a_scale=so)

    # ---- reference API ----
    @torch.no_grad()
    def quantize(self, h: Tensor) -> Tuple[Tensor, Tensor]:
        """VectorQuantizer.forward(is_voxel=True) value path: (quant NCDHW, indices)."""
        zl = ops.nchw_to_ndhwc(h.to(torch.float32), cpad=4)
        idx, zq = ops.vq_lookup(zl, self._sd["quantize.embedding.weight"])
        return ops.ndhwc_to_nchw(zq, c=self.embed_dim), idx

    # Objects decode independently; a 64^3 x 128-channel activation is 134 MB per object (4.3 GB at 32), so large
    # batches are decoded in slices of this many, which bounds the workspace.
    MAX_DECODE_BATCH = 16

    @torch.no_grad()
    def decode(self, quant: Tensor) -> Tensor:
        """network.py:90-93: post_quant_conv + decoder on an already-quantised latent (NCDHW)."""
        if self._packed is None:
            self._pack()
        if quant.shape[0] > self.MAX_DECODE_BATCH:
            return torch.cat([self.decode(quant[i:i + self.MAX_DECODE_BATCH])
                              for i in range(0, quant.shape[0], self.MAX_DECODE_BATCH)], dim=0)
        zl = ops.nchw_to_ndhwc(quant.to(torch.float32), cpad=4)
        return self._decode_cl(zl)

    def _decode_cl(self, zl: Tensor) -> Tensor:
        q4 = ops.conv_gemm(zl, self._packed["post_quant_conv"], math=self.math)   # [nb,d,h,w,4], channel 3 == 0
        return ops.ndhwc_to_nchw(self.decoder_ndhwc(q4))

    @torch.no_grad()
    def decode_no_quant(self, h: Tensor, force_not_quantize: bool = False) -> Tensor:
        """network.py:95-103: (despite the name) quantise to the nearest code, then decode."""
        if self._packed is None:
            self._pack()
        if h.shape[0] > self.MAX_DECODE_BATCH:
            outs, idxs = [], []
            for i in range(0, h.shape[0], self.MAX_DECODE_BATCH):
                outs.append(self.decode_no_quant(h[i:i + self.MAX_DECODE_BATCH], force_not_quantize))
                if not force_not_quantize:
                    idxs.append(self.last_indices)
            if idxs:
                self.last_indices = torch.cat(idxs)
            return torch.cat(outs, dim=0)
        zl = ops.nchw_to_ndhwc(h.to(torch.float32), cpad=4)
        if not force_not_quantize:
            idx, zl = ops.vq_lookup(zl, self._sd["quantize.embedding.weight"])
            self.last_indices = idx
        return self._decode_cl(zl)


def load_vqvae(vq_conf, vq_ckpt: str, opt=None, device: Optional[str] = None) -> VQVAE:
    """model/model_utils.py:7-31.  Accepts a raw state_dict or {'vqvae': state_dict}."""
    mp = vq_conf["model"]["params"] if isinstance(vq_conf, dict) else vq_conf.model.params
    g = (lambda o, k: o[k]) if isinstance(mp, dict) else getattr
    dev = device or (opt.hyper.device if opt is not None else "cuda")
    vq = VQVAE(g(mp, "ddconfig"), g(mp, "n_embed"), g(mp, "embed_dim"), device=dev)
    sd = torch.load(vq_ckpt, map_location="cpu")
    vq.load_state_dict(sd["vqvae"] if "vqvae" in sd else sd)
    return vq.eval()

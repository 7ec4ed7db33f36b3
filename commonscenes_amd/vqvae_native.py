"""VQ-VAE decode through the native whole-decode driver (csrc/cs_vqvae.hip: cs_vqvae_create / _pack / _decode).

`NativeVQVAE` has the decode-side surface of `commonscenes_amd.vqvae.VQVAE` (reference `VQVAE`,
vqvae_networks/network.py:48-103): `decode_no_quant(h, force_not_quantize=False)`, `decode(quant)`,
`load_state_dict` / `state_dict` with the reference's keys.  Same kernels in the same order as the Python
sequencer, so the two agree bit for bit (tests/test_vqvae_native_gpu.py)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .vqvae import _dd

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class NativeVQVAE:
    def __init__(self, ddconfig, n_embed: int, embed_dim: int, device: str | torch.device = "cuda",
                 math: str | int | None = None):
        self.cfg = _dd(ddconfig)
        self.n_embed, self.embed_dim = int(n_embed), int(embed_dim)
        self.device = torch.device(device)
        self.math = L.DEFAULT_MATH if math is None else {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(math, math)
        self._h = None
        self._arena: Optional[Tensor] = None
        self._sd: Dict[str, Tensor] = {}
        self._ws: Optional[Tensor] = None
        self.last_indices: Optional[Tensor] = None
        self.grid = self.cfg["resolution"] >> (len(self.cfg["ch_mult"]) - 1)
        self._create()

    def _create(self):
        cfg = self.cfg
        c = L.CsVqvaeConfig()
        c.ch, c.out_ch, c.n_mult = cfg["ch"], cfg["out_ch"], len(cfg["ch_mult"])
        for i, v in enumerate(cfg["ch_mult"]):
            c.ch_mult[i] = int(v)
        c.num_res_blocks, c.z_channels, c.resolution = cfg["num_res_blocks"], cfg["z_channels"], cfg["resolution"]
        c.n_embed, c.embed_dim, c.math = self.n_embed, self.embed_dim, self.math
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.cs_vqvae_create(C.byref(c), C.byref(h)), "cs_vqvae_create")
        self._h = h
        self.params: "OrderedDict[str, Tuple[Tuple[int, ...], int]]" = OrderedDict()
        name, shape, nd, off = C.c_char_p(), (C.c_int64 * 5)(), C.c_int(), C.c_int64()
        for i in range(lib.cs_vqvae_param_count(h)):
            L.check(lib.cs_vqvae_param_info(h, i, C.byref(name), C.byref(shape), C.byref(nd), C.byref(off)),
                    "cs_vqvae_param_info")
            self.params[name.value.decode()] = (tuple(int(shape[k]) for k in range(nd.value)), int(off.value))

    def set_math(self, mode) -> "NativeVQVAE":
        """GEMM numerics ('fp32' | 'f16x3'): a property of the plan, so switching rebuilds it and re-packs the weights
        (as NativeDiffusionUNet.set_math does) -- what rel2shape's F16X3-overflow fall-back calls."""
        m = {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(mode, mode)
        if m not in (L.MATH_FP32, L.MATH_F16X3):
            raise ValueError(f"unknown math mode {mode!r}")
        if m != self.math:
            self.math = m
            sd = {k: v.clone() for k, v in self._sd.items()}      # the entries are views into the raw buffer being replaced
            self.__del__()
            self._arena = None
            self._ws = None
            self._sd = {}
            self._create()
            if sd:
                self.load_state_dict(sd)
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().cs_vqvae_destroy(h)
            except Exception:       # interpreter shutdown
                pass

    @property
    def shapes(self) -> "OrderedDict[str, Tuple[int, ...]]":
        return OrderedDict((k, v[0]) for k, v in self.params.items())

    def state_dict(self) -> "OrderedDict[str, Tensor]":
        return OrderedDict((k, self._sd[k]) for k in self.params if k in self._sd)

    def load_state_dict(self, sd, strict: bool = False):
        """Entries outside the decode path (encoder.*, quant_conv.*) are ignored, as in vqvae.VQVAE."""
        if self.device.type != "cuda":
            raise L.CsError("NativeVQVAE: weights must be on the HIP device (no CPU path)")
        missing = [k for k in self.params if k not in sd]
        if missing:
            raise RuntimeError(f"NativeVQVAE.load_state_dict: missing {missing[:5]}")
        lib = L.load()
        raw = torch.empty(int(lib.cs_vqvae_raw_bytes(self._h)), dtype=torch.uint8, device=self.device)
        for k, (shp, off) in self.params.items():
            t = sd[k]
            if tuple(t.shape) != shp:
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {shp}")
            dst = raw[off:off + 4 * t.numel()].view(torch.float32).view(shp)
            dst.copy_(t.detach().to(device=self.device, dtype=torch.float32))
            self._sd[k] = dst
        self._raw = raw
        self._arena = torch.empty(int(lib.cs_vqvae_arena_bytes(self._h)), dtype=torch.uint8, device=self.device)
        L.check(lib.cs_vqvae_pack(self._h, raw.data_ptr(), self._arena.data_ptr(), _stream()), "cs_vqvae_pack")
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(self._sd.values())

    def _run(self, h: Tensor, quantize: bool) -> Tensor:
        if self._arena is None:
            raise RuntimeError("NativeVQVAE: weights not loaded")
        h = h.to(device=self.device, dtype=torch.float32).contiguous()
        nb = h.shape[0]
        g = self.grid
        if tuple(h.shape[1:]) != (self.embed_dim, g, g, g):
            raise ValueError(f"latent must be (B, {self.embed_dim}, {g}, {g}, {g}), got {tuple(h.shape)}")
        lib = L.load()
        need = int(lib.cs_vqvae_workspace_bytes(self._h, nb))
        if need < 0:
            L.check(need, "cs_vqvae_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        r = self.cfg["resolution"]
        out = torch.empty((nb, self.cfg["out_ch"], r, r, r), dtype=torch.float32, device=self.device)
        idx = torch.empty((nb * g ** 3,), dtype=torch.int64, device=self.device) if quantize else None
        L.check(lib.cs_vqvae_decode(self._h, self._arena.data_ptr(), h.data_ptr(), out.data_ptr(),
                                    idx.data_ptr() if idx is not None else None, nb, 1 if quantize else 0,
                                    ops.status_word(self.device).data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                    _stream()), "cs_vqvae_decode")
        if quantize:
            self.last_indices = idx
        return out

    @torch.no_grad()
    def decode(self, quant: Tensor) -> Tensor:
        """network.py:90-93: post_quant_conv + decoder on an already-quantised latent."""
        return self._run(quant, False)

    @torch.no_grad()
    def decode_no_quant(self, h: Tensor, force_not_quantize: bool = False) -> Tensor:
        """network.py:95-103: (despite the name) quantise to the nearest code, then decode."""
        return self._run(h, not force_not_quantize)

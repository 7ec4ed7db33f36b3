"""The UNet forward through the native whole-forward driver (csrc/cs_unet.hip: cs_unet_create / cs_unet_pack /
cs_unet_context / cs_unet_step) -- the SURVEY 8b "cs_unet_step" entry point, usable from any host language.

`NativeDiffusionUNet` offers the same surface as `commonscenes_amd.unet.DiffusionUNet` (reference
`DiffusionUNet`, network.py:11-42: `df(x, t, c_crossattn=[ctx])`, `.conditioning_key`, `load_state_dict` /
`state_dict` with the reference's keys) plus `forward_cfg`; it launches the same kernels in the same order, so
the two drivers agree bit for bit (tests/test_unet_native_gpu.py).  Python only owns the torch buffers.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .unet import _cfg

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class NativeDiffusionUNet:
    def __init__(self, unet_params, vq_conf=None, conditioning_key: Optional[str] = "crossattn",
                 device: str | torch.device = "cuda", math: str | int | None = None, grid: Tuple[int, int, int] = (16, 16, 16)):
        self.cfg = _cfg(unet_params)
        self.conditioning_key = conditioning_key
        self.device = torch.device(device)
        self.math = L.DEFAULT_MATH if math is None else {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(math, math)
        self.grid = tuple(grid)
        self._h = None
        self._arena: Optional[Tensor] = None
        self._sd: Dict[str, Tensor] = {}
        self._ws: Optional[Tensor] = None
        self._ws_key = None
        self._ctx_cache = None
        self._create()

    # ---- plan ---------------------------------------------------------------------------------------
    def _create(self):
        cfg = self.cfg
        c = L.CsUnetConfig()
        c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
        c.model_channels, c.num_res_blocks = cfg["model_channels"], cfg["num_res_blocks"]
        mult, ar = list(cfg["channel_mult"]), list(cfg["attention_resolutions"])
        if len(mult) > 8 or len(ar) > 8:
            raise L.CsError("channel_mult / attention_resolutions: at most 8 entries")
        c.n_mult, c.n_attn_res = len(mult), len(ar)
        for i, v in enumerate(mult):
            c.channel_mult[i] = int(v)
        for i, v in enumerate(ar):
            c.attention_resolutions[i] = int(v)
        c.num_heads, c.context_dim = cfg["num_heads"], int(cfg["context_dim"] or 0)
        c.d, c.h, c.w = self.grid
        c.math = self.math
        c.use_spatial_transformer, c.dims = int(cfg["use_spatial_transformer"]), cfg["dims"]
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.cs_unet_create(C.byref(c), C.byref(h)), "cs_unet_create")
        self._h = h
        self.params: "OrderedDict[str, Tuple[Tuple[int, ...], int]]" = OrderedDict()
        name, shape, nd, off = C.c_char_p(), (C.c_int64 * 5)(), C.c_int(), C.c_int64()
        for i in range(lib.cs_unet_param_count(h)):
            L.check(lib.cs_unet_param_info(h, i, C.byref(name), C.byref(shape), C.byref(nd), C.byref(off)),
                    "cs_unet_param_info")
            self.params[name.value.decode()] = (tuple(int(shape[k]) for k in range(nd.value)), int(off.value))
        self.ctx_floats = int(lib.cs_unet_context_floats(h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().cs_unet_destroy(h)
            except Exception:       # interpreter shutdown
                pass

    @property
    def shapes(self) -> "OrderedDict[str, Tuple[int, ...]]":
        return OrderedDict((k, v[0]) for k, v in self.params.items())

    # ---- nn.Module-like surface ---------------------------------------------------------------------------
    def state_dict(self) -> "OrderedDict[str, Tensor]":
        return OrderedDict((k, self._sd[k]) for k in self.params if k in self._sd)

    def load_state_dict(self, sd, strict: bool = True):
        if self.device.type != "cuda":
            raise L.CsError("NativeDiffusionUNet: weights must be on the HIP device (no CPU path)")
        missing = [k for k in self.params if k not in sd]
        unexpected = [k for k in sd if k not in self.params]
        if missing or (strict and unexpected):
            raise RuntimeError(f"NativeDiffusionUNet.load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        lib = L.load()
        raw = torch.empty(int(lib.cs_unet_raw_bytes(self._h)), dtype=torch.uint8, device=self.device)
        for k, (shp, off) in self.params.items():
            t = sd[k]
            if tuple(t.shape) != shp:
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {shp}")
            n = t.numel()
            dst = raw[off:off + 4 * n].view(torch.float32).view(shp)
            dst.copy_(t.detach().to(device=self.device, dtype=torch.float32))
            self._sd[k] = dst                        # views into the raw buffer (kept for state_dict())
        self._raw = raw
        self._arena = torch.empty(int(lib.cs_unet_arena_bytes(self._h)), dtype=torch.uint8, device=self.device)
        L.check(lib.cs_unet_pack(self._h, raw.data_ptr(), self._arena.data_ptr(), _stream()), "cs_unet_pack")
        self._ctx_cache = None
        return self

    def set_math(self, mode) -> "NativeDiffusionUNet":
        """GEMM numerics ('fp32' | 'f16x3'): a property of the plan, so switching rebuilds it and re-packs."""
        m = {"fp32": L.MATH_FP32, "f16x3": L.MATH_F16X3}.get(mode, mode)
        if m not in (L.MATH_FP32, L.MATH_F16X3):
            raise ValueError(f"unknown math mode {mode!r}")
        if m != self.math:
            self.math = m
            sd = dict(self._sd)
            self.__del__()
            self._create()
            self._ws_key = None
            if sd:
                self.load_state_dict(sd)
        return self

    def parameters(self):
        return iter(self._sd.values())

    def eval(self):
        return self

    def num_parameters(self) -> int:
        n = 0
        for shp, _ in self.params.values():
            k = 1
            for v in shp:
                k *= v
            n += k
        return n

    # ---- forward --------------------------------------------------------------------------------------------
    def _workspace(self, nb_x: int, cfg_pairs: bool) -> Tensor:
        key = (nb_x, cfg_pairs)
        if self._ws_key != key:
            need = int(L.load().cs_unet_workspace_bytes(self._h, nb_x, 1 if cfg_pairs else 0))
            if need < 0:
                L.check(need, "cs_unet_workspace_bytes")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def reset_run_cache(self) -> None:
        """Drop the per-run context-vector cache (see DiffusionUNet.reset_run_cache)."""
        self._ctx_cache = None

    def check_overflow(self) -> None:
        ops.check_overflow(self.device, "NativeDiffusionUNet")

    def context_vectors(self, ctx: Tensor, ws: Tensor) -> Tensor:
        """cs_unet_context, cached per conditioning tensor (the sampler passes the same [uc; c] every step)."""
        if ctx.dim() == 3:
            if ctx.shape[1] != 1:
                raise NotImplementedError("the native driver implements the shipped one-token context (SURVEY F4)")
            ctx = ctx[:, 0]
        key = (ctx.data_ptr(), ctx._version, tuple(ctx.shape))
        if self._ctx_cache is not None and self._ctx_cache[0] == key:
            return self._ctx_cache[1]
        ctx = ctx.to(torch.float32).contiguous()
        vec = torch.empty((ctx.shape[0], self.ctx_floats), dtype=torch.float32, device=self.device)
        L.check(L.load().cs_unet_context(self._h, self._arena.data_ptr(), ctx.data_ptr(), ctx.shape[0], vec.data_ptr(),
                                         ops.status_word(self.device).data_ptr(), ws.data_ptr(), ws.numel(),
                                         _stream()), "cs_unet_context")
        # r5: per transformer block the largest |entry| of its row vector -> the static bound of the block's t1 / t2
        # operands (cs_unet_set_context_bounds; unet.py::_context_vectors does the same): ONE read-back per run
        if self.math == L.MATH_F16X3 and self.cfg["use_spatial_transformer"] and not L.debug().no_static_scales:
            from .unet import unet_blocks
            inp, mid, out = unet_blocks(self.cfg)[:3]
            widths = [l["cin"] for layers in (list(inp) + [mid] + list(out)) for l in layers if l["kind"] == "attn"]
            if widths and sum(widths) != self.ctx_floats:
                # the static t1 / t2 scales would silently omit the context term and the run would rest on the overflow
                # flag alone: a layout mismatch between unet_blocks() and the driver is a bug, say so
                raise L.CsError(f"cs_unet context row vector has {self.ctx_floats} floats, the attention blocks' widths sum "
                                f"to {sum(widths)}: cannot derive the per-block context bounds")
            if widths:
                mx = torch.stack([seg.abs().max() for seg in torch.split(vec, widths, dim=1)]).cpu().tolist()
                arr = (C.c_float * len(widths))(*[float(v) for v in mx])
                L.check(L.load().cs_unet_set_context_bounds(self._h, arr, len(widths)), "cs_unet_set_context_bounds")
        self._ctx_cache = (key, vec, ctx)
        return vec

    def _step(self, x: Tensor, t: Tensor, ctx: Tensor, cfg_pairs: bool) -> Tensor:
        if self._arena is None:
            raise RuntimeError("NativeDiffusionUNet: weights not loaded")
        x = x.to(torch.float32).contiguous()
        t = t.to(torch.int64).contiguous()
        nb = x.shape[0]
        if tuple(x.shape[1:]) != (self.cfg["in_channels"], *self.grid):
            raise ValueError(f"x must be (B, {self.cfg['in_channels']}, {self.grid}), got {tuple(x.shape)}")
        ws = self._workspace(nb, cfg_pairs)
        nbo = 2 * nb if cfg_pairs else nb
        if ctx is None:                     # concat family: the condition volume is part of x, there is no context
            vec_ptr = None
        else:
            vec = self.context_vectors(ctx, ws)
            if vec.shape[0] != nbo:
                raise ValueError("context batch does not match x")
            vec_ptr = vec.data_ptr()
        out = torch.empty((nbo, self.cfg["out_channels"], *self.grid), dtype=torch.float32, device=self.device)
        L.check(L.load().cs_unet_step(self._h, self._arena.data_ptr(), x.data_ptr(), t.data_ptr(), vec_ptr,
                                      out.data_ptr(), nb, 1 if cfg_pairs else 0,
                                      ops.status_word(self.device).data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                "cs_unet_step")
        return out

    @torch.no_grad()
    def forward_cfg(self, x: Tensor, t: Tensor, c_in: Tensor) -> Tensor:
        """[eps_uc; eps_c] for the guidance pair batch without duplicating (x, t) (samplers/ddim.py:206-209)."""
        if self.conditioning_key == "concat":      # nothing upstream of the condition to share: duplicated batch
            return self.forward(torch.cat([x, x]), torch.cat([t, t]), c_concat=[c_in])
        return self._step(x, t, c_in, True)

    @torch.no_grad()
    def forward(self, x: Tensor, t: Tensor, c_concat: Optional[list] = None,
                c_crossattn: Optional[list] = None) -> Tensor:
        """network.py:20-42, crossattn branch."""
        if self.conditioning_key == "concat":          # network.py:25-27
            if not c_concat or self.cfg["use_spatial_transformer"]:
                raise NotImplementedError("concat conditioning needs c_concat and AttentionBlock blocks")
            xc = torch.cat([x.to(torch.float32)] + [c.to(device=x.device, dtype=torch.float32) for c in c_concat], dim=1)
            return self._step(xc, t, None, False)
        if self.conditioning_key != "crossattn" or not self.cfg["use_spatial_transformer"]:
            raise NotImplementedError(f"conditioning_key={self.conditioning_key!r} with this block type")
        if c_crossattn is None:
            raise ValueError("c_crossattn is required for conditioning_key='crossattn'")
        ctx = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self._step(x, t, ctx, False)

    __call__ = forward

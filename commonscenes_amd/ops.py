"""Thin torch-tensor front end over the C ABI (include/commonscenes_hip.h).

torch is used for device memory, streams and shapes only; every arithmetic op below is one call
into libcommonscenes_hip.so on the current HIP stream.  Tensors are fp32, channels-last
(`[nb, d, h, w, c]` or token matrices `[nb, n, c]`), possibly views into a wider buffer (row stride
`ld` > c) so channel concatenation needs no extra pass.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import os

import torch

from . import lib as L

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# One sticky CS_STATUS_* word per device (include/commonscenes_hip.h): every F16X3 kernel launched from this module ORs
# CS_STATUS_F16X3_OVERFLOW into it when an operand leaves the fp16 range.  Nothing reads it inside the sampling
# loop; the model classes call check_overflow() once per sampling run / decode (one scalar read-back).
_STATUS: dict = {}


def status_word(device=None) -> Tensor:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _STATUS.get(dev.index)
    if t is None:
        t = _STATUS[dev.index] = torch.zeros((1,), dtype=torch.int32, device=dev)
    return t


def read_status(device=None, reset: bool = True) -> int:
    """The device's status word (synchronises: one 4-byte read-back); cleared when `reset`."""
    t = status_word(device)
    v = int(t.item())
    if v and reset:
        t.zero_()
    return v


def clear_status(device=None) -> None:
    """Stream-ordered clear of the device's status word (no read-back, no host synchronisation)."""
    status_word(device).zero_()


def check_overflow(device=None, what: str = "F16X3 kernels") -> None:
    """Raise CsOverflowError if any F16X3 kernel since the last check met an operand beyond the fp16 range: a value a
    is carried as fp16 halves of a * a_scale, where a_scale is the layer's operand scale -- derived from the producing
    normalisation's bound (norm_a_scale: cannot overflow) or 16 for raw activations (|a| >= 65504 / 16 ~ 4094
    overflows).  That launch's output is garbage; the caller re-runs with set_math('fp32')."""
    if read_status(device) & L.STATUS_F16X3_OVERFLOW:
        raise L.CsOverflowError(f"{what}: an activation left the fp16 range of CS_MATH_F16X3 (|a| * a_scale >= 65504, "
                                "a_scale = the layer's operand scale: 16 for raw activations); "
                                "results of this run are invalid -- use set_math('fp32')")


# Index-error word per device: the gather kernels (cs_gcn_gather_cat, cs_gcn_segment_mean, cs_embedding) set it when an
# edge endpoint / embedding index is out of range (and skip that entry).  The reference's nn.Embedding / tensor indexing
# raise IndexError for the same input (VAEGAN_V2FULL.py:225-226, graph.py:146-147): check_index_errors() -- one
# read-back per encoder / decoder call -- turns the flag into that exception.
_INDEX_ERR: dict = {}


def index_err_word(device=None) -> Tensor:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _INDEX_ERR.get(dev.index)
    if t is None:
        t = _INDEX_ERR[dev.index] = torch.zeros((1,), dtype=torch.int32, device=dev)
    return t


def check_index_errors(device=None, what: str = "scene-graph indices") -> None:
    t = index_err_word(device)
    if int(t.item()):
        t.zero_()
        raise IndexError(f"{what}: index out of range (an object / predicate id beyond the embedding table, or a "
                         "triple endpoint beyond the node count)")


def _chk(t: Tensor, name: str, dtype=torch.float32) -> None:
    if not t.is_cuda:
        raise L.CsError(f"{name}: expected a HIP device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise L.CsError(f"{name}: expected {dtype}, got {t.dtype}")


def rows_ld(t: Tensor, name: str = "tensor") -> Tuple[int, int, int]:
    """View `t` as a row matrix [M, c] with row stride ld; raise unless its layout allows it."""
    if t.dim() < 1 or (t.shape[-1] != 1 and t.stride(-1) != 1):
        raise L.CsError(f"{name}: innermost dim must be contiguous")
    c = t.shape[-1]
    dims = [(s, st) for s, st in zip(t.shape[:-1], t.stride()[:-1]) if s != 1]
    if not dims:
        return 1, c, c
    ld = dims[-1][1]
    if ld < c:
        raise L.CsError(f"{name}: overlapping rows {tuple(t.shape)} {t.stride()}")
    m = 1
    expect = ld
    for s, st in reversed(dims):
        if st != expect:
            raise L.CsError(f"{name}: not a row-strided layout {tuple(t.shape)} {t.stride()}")
        expect *= s
        m *= s
    return m, c, ld


@dataclass
class PackedWeight:
    """Weights re-laid-out for the implicit GEMM: wt[tap][cin_pad][ldw], bias[cout]."""
    wt: Optional[Tensor]
    bias: Optional[Tensor]
    cout: int
    cin: int
    cin_pad: int
    ldw: int
    k: Tuple[int, int, int]
    math: int = L.MATH_FP32
    wh: Optional[Tensor] = None      # CS_MATH_F16X3: hi / lo fp16 halves, [tap][cin16/8][cout][8]
    wl: Optional[Tensor] = None
    acc_scale: float = 1.0
    # nearest-x2-upsample + 3x3x3 conv folded onto the source grid (cs_conv_gemm_up2): the doubled dims and one
    # PackedWeight per output parity class (3x2x2 / 2x2x2 kernels with pre-summed taps); wt / wh / wl are then unused
    up: Optional[Tuple[int, int, int]] = None
    classes: Optional[list] = None
    # thin-output 3x3x3 conv as "taps as columns" (pack_weight_tapcol): the pointwise pack with 27 * cout (+ pad)
    # columns; wt / wh / wl are then unused and `bias` is added by cs_tapsum27
    tapcol: Optional["PackedWeight"] = None


def pack_weight(w: Tensor, bias: Optional[Tensor] = None, cin_pad: Optional[int] = None,
                math: int = L.MATH_FP32, fold_up: Optional[Sequence[int]] = None,
                amax: Optional[float] = None) -> PackedWeight:
    """torch Conv3d (cout,cin,kd,kh,kw) or Linear (out,in) weight -> PackedWeight (device op).
    fold_up=(ud,uh,uw): the conv follows a nearest x2 upsampling of the flagged dims; its taps are pre-summed per output
    parity class (cs_fold_upsample_weight) and conv_gemm(..., up=fold_up) runs on the source grid."""
    if fold_up is not None and any(fold_up) and FOLD_UPSAMPLE:
        return _pack_weight_folded(w, bias, cin_pad, math, tuple(int(u) for u in fold_up))
    if math == L.MATH_F16X3:
        return _pack_weight_f16x3(w, bias, cin_pad, amax)
    _chk(w, "weight")
    w = w.contiguous()
    if w.dim() == 5:
        cout, cin, kd, kh, kw = w.shape
    elif w.dim() == 2:
        cout, cin = w.shape
        kd = kh = kw = 1
    else:
        raise L.CsError("weight must be 2-D (Linear) or 5-D (Conv3d)")
    taps = kd * kh * kw
    cp = cin_pad if cin_pad is not None else (cin + 3) // 4 * 4
    ldw = (cout + 3) // 4 * 4
    wt = torch.empty((taps, cp, ldw), dtype=torch.float32, device=w.device)
    L.check(L.load().cs_relayout_weight(w.data_ptr(), wt.data_ptr(), cout, cin, taps, cp, ldw, _stream()),
            "cs_relayout_weight")
    b = None
    if bias is not None:
        _chk(bias, "bias")
        b = bias.contiguous()
    return PackedWeight(wt, b, cout, cin, cp, ldw, (kd, kh, kw))


A_SCALE = 16.0      # activation pre-scale of the f16x3 mode (CsConvGemm.a_scale) for operands of unknown range; power of two


def norm_a_scale(gmax: float, bmax: float, n: int) -> float:
    """Operand scale of an F16X3 GEMM fed by a GroupNorm / LayerNorm (+ SiLU / GELU / identity) over n elements per
    statistic.  A normalised value obeys |x^| <= sqrt(n - 1), so |y| <= gmax * sqrt(n - 1) + bmax =: bound (|silu(y)|,
    |gelu(y)| <= |y|): the largest power of two 2^k with bound * 2^k <= 65000 < 65504 cannot overflow the fp16 range
    WHATEVER the input -- the producer's bound replaces the fixed guess of 16 (r3; VERDICT r2 next #7c) -- and is 16-128x
    larger for the shipped layers, so the absolute floor 2^-25 / a_scale of tiny operands (the lo half leaves fp16's
    normal range below a_scale * |a| = 2^-3) drops accordingly; a layer whose affine parameters are uniformly tiny gets a
    correspondingly huge scale and keeps full relative precision.  k is clamped to [-8, 40] (products stay below
    2^16 * 2^14 by construction, acc_scale = 2^-(k + weight exponent) stays a normal fp32).  Every step is an IEEE double operation, mirrored in csrc/cs_driver.h::norm_a_scale: both
    hosts derive the same scale."""
    import math as _m
    bound = float(gmax) * _m.sqrt(float(max(n - 1, 1))) + float(bmax)
    if not (bound > 0.0) or not _m.isfinite(bound):
        return 2.0 ** 40
    k = _m.frexp(65000.0 / bound)[1] - 1
    return 2.0 ** max(-8, min(40, k))


# CS_NO_UPFOLD=1: upsample convs take the direct form (27 taps on the doubled grid) -- A/B runs
FOLD_UPSAMPLE = not os.environ.get("CS_NO_UPFOLD")


def _pack_weight_folded(w: Tensor, bias: Optional[Tensor], cin_pad: Optional[int], math: int,
                        up: Tuple[int, int, int]) -> PackedWeight:
    _chk(w, "weight")
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise L.CsError("fold_up needs a 3x3x3 conv weight")
    lib = L.load()
    cout, cin = w.shape[:2]
    n, kd, kh, kw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    L.check(lib.cs_conv_up2_info(*up, C.byref(n), C.byref(kd), C.byref(kh), C.byref(kw)), "cs_conv_up2_info")
    wf = torch.empty((n.value, cout, cin, kd.value, kh.value, kw.value), dtype=torch.float32, device=w.device)
    L.check(lib.cs_fold_upsample_weight(w.contiguous().data_ptr(), wf.data_ptr(), cout, cin, *up, _stream()),
            "cs_fold_upsample_weight")
    classes = [pack_weight(wf[c], bias, cin_pad, math) for c in range(n.value)]
    c0 = classes[0]
    return PackedWeight(None, c0.bias, cout, cin, c0.cin_pad, c0.ldw, (3, 3, 3), math, None, None, 1.0, up, classes)


def _pack_weight_f16x3(w: Tensor, bias: Optional[Tensor], cin_pad: Optional[int],
                       amax: Optional[float] = None) -> PackedWeight:
    """fp32 weight -> (hi, lo) fp16 halves of w * 2^s with max|w| * 2^s < 2^14 (cs_pack_weight_f16x3).
    `amax`: take the power-of-two scale from this magnitude instead of max|w| -- an input-channel slice of a tensor packed
    with the WHOLE tensor's scale (both hosts do that for the channel-split convs, so they stay bit-identical)."""
    import math as _m
    _chk(w, "weight")
    w = w.contiguous()
    if w.dim() == 5:
        cout, cin, kd, kh, kw = w.shape
    elif w.dim() == 2:
        cout, cin = w.shape
        kd = kh = kw = 1
    else:
        raise L.CsError("weight must be 2-D (Linear) or 5-D (Conv3d)")
    taps = kd * kh * kw
    cp = cin_pad if cin_pad is not None else (cin + 3) // 4 * 4
    if amax is None:
        amax = float(w.abs().max().item())        # weight preparation (load time), not the sampling loop
    e = _m.frexp(amax)[1] if amax > 0 and _m.isfinite(amax) else 0
    scale = 2.0 ** (14 - e)
    kg = (cin + 15) // 16 * 2
    wh = torch.empty((taps, kg, cout, 8), dtype=torch.float16, device=w.device)
    wl = torch.empty_like(wh)
    L.check(L.load().cs_pack_weight_f16x3(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), cout, cin, taps, scale,
                                          _stream()), "cs_pack_weight_f16x3")
    b = None
    if bias is not None:
        _chk(bias, "bias")
        b = bias.contiguous()
    return PackedWeight(None, b, cout, cin, cp, cout, (kd, kh, kw), L.MATH_F16X3, wh, wl, 1.0 / (scale * A_SCALE))


# CS_NO_TAPCOL=1: thin-output convs stay on the implicit GEMM (A/B runs)
TAPCOL = not os.environ.get("CS_NO_TAPCOL")


def tapcol_ok(w: Tensor, math: int) -> bool:
    """the rule both hosts apply (csrc/cs_driver.h::add_layer_gemm): F16X3, 3x3x3, at most 4 output channels"""
    return (TAPCOL and math == L.MATH_F16X3 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[0] <= 4
            and w.shape[1] % 4 == 0)


def pack_weight_tapcol(w: Tensor, bias: Optional[Tensor] = None) -> PackedWeight:
    """Conv3d (cout <= 4, cin, 3, 3, 3) weight -> "taps as columns" pack: the pointwise F16X3 weight [ncolp][cin] with row
    o * 27 + t = w[o, :, t] (cs_pack_weight_f16x3_tapcol); conv_gemm then runs ONE 1x1x1 GEMM with 27 * cout columns and
    cs_tapsum27 adds the 27 shifted columns of every output channel (+ bias).  Same power-of-two weight scale as the
    ordinary pack (the tensor's max |w|)."""
    import math as _m
    _chk(w, "weight")
    w = w.contiguous()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3) or cout > 4:
        raise L.CsError("pack_weight_tapcol: (cout <= 4, cin, 3, 3, 3) weights only")
    amax = float(w.abs().max().item())
    e = _m.frexp(amax)[1] if amax > 0 and _m.isfinite(amax) else 0
    scale = 2.0 ** (14 - e)
    ncolp = (27 * cout + 3) // 4 * 4
    kg = (cin + 15) // 16 * 2
    wh = torch.empty((1, kg, ncolp, 8), dtype=torch.float16, device=w.device)
    wl = torch.empty_like(wh)
    L.check(L.load().cs_pack_weight_f16x3_tapcol(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), cout, cin, ncolp, scale,
                                                 _stream()), "cs_pack_weight_f16x3_tapcol")
    b = None
    if bias is not None:
        _chk(bias, "bias")
        b = bias.contiguous()
    cp = (cin + 3) // 4 * 4
    pw = PackedWeight(None, None, ncolp, cin, cp, ncolp, (1, 1, 1), L.MATH_F16X3, wh, wl, 1.0 / (scale * A_SCALE))
    return PackedWeight(None, b, cout, cin, cp, cout, (3, 3, 3), L.MATH_F16X3, None, None, pw.acc_scale, None, None, pw)


def tapcol_tile(m: int, ncolp: int) -> int:
    """tile of the taps-as-columns GEMM (mirrored in cs_driver.h): 256-row tiles once they fill the chip"""
    if os.environ.get("CS_TAPCOL_TILE"):               # tuning runs (Python host only)
        return int(os.environ["CS_TAPCOL_TILE"])
    if (m + 255) // 256 < 192:
        return 0
    return 7 if ncolp <= 64 else 6


def _conv_tapcol(x, w: PackedWeight, spatial, a_scale, out, out_fn) -> Tensor:
    xt = x.hi if isinstance(x, Split16) else x.t if isinstance(x, Pair16) else x
    if spatial is None:
        if xt.dim() != 5:
            raise L.CsError("conv needs x as [nb,d,h,w,c] or an explicit spatial=")
        nb, d, h, wd = (int(v) for v in xt.shape[:4])
    else:
        nb, d, h, wd = (int(v) for v in spatial)
    m = nb * d * h * wd
    pw = w.tapcol
    y = conv_gemm(x, pw, spatial=(nb, d, h, wd), tile=tapcol_tile(m, pw.cout), a_scale=a_scale)
    oshape = (nb, d, h, wd, w.cout)
    if out is None:
        out = out_fn(oshape) if out_fn is not None else torch.empty(oshape, dtype=torch.float32, device=xt.device)
        if tuple(out.shape) != oshape:
            out = out.view(oshape)
    _chk(out, "out")
    om, oc, ldo = rows_ld(out, "out")
    if om != m or oc != w.cout:
        raise L.CsError(f"out has shape {tuple(out.shape)}, expected {m} rows x {w.cout}")
    ym, yc, ldy = rows_ld(y, "y")
    L.check(L.load().cs_tapsum27(y.data_ptr(), _ptr(w.bias), out.data_ptr(), nb, d, h, wd, w.cout, ldy, ldo, _stream()),
            "cs_tapsum27")
    return out


def pack_geglu_weight(w: Tensor, bias: Tensor, group: int = 112) -> PackedWeight:
    """GEGLU.proj (attention.py:42) weight (2H, C) -> f16x3 pack whose output columns are interleaved per
    `2*group`-column GEMM tile as [x (group) | gate (group)], so the gate is applied in the GEMM epilogue
    (act=ACT_GEGLU) and the (.., 2H) intermediate never reaches HBM."""
    h2 = w.shape[0]
    h = h2 // 2
    if h % group:
        raise L.CsError(f"GEGLU width {h} is not a multiple of {group}")
    idx = torch.arange(h, device=w.device).view(h // group, group)
    perm = torch.cat([idx, idx + h], dim=1).reshape(-1)          # [x tile0 | gate tile0 | x tile1 | ...]
    return pack_weight(w[perm].contiguous(), bias[perm].contiguous(), math=L.MATH_F16X3)


def conv_gemm(x: Tensor, w: PackedWeight, *, spatial: Optional[Tuple[int, int, int, int]] = None,
              stride: Sequence[int] = (1, 1, 1), up: Sequence[int] = (0, 0, 0), act: int = L.ACT_NONE,
              rowvec: Optional[Tensor] = None, rv_rows: int = 1, res: Optional[Tensor] = None,
              scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
              out: Optional[Tensor] = None, tile: int = 0, math: int = L.MATH_FP32,
              splitk: Optional[int] = None, out_fn=None, a_scale: Optional[float] = None) -> Tensor:
    """Conv3d (k in {1,3}, pad k//2) / Linear.  x: [nb,d,h,w,c] (conv) or [..., c] rows (linear).

    `spatial=(nb,d,h,w)` lets a row matrix be interpreted as a volume without reshaping.
    """
    if w.tapcol is not None:
        if (tuple(stride) != (1, 1, 1) or tuple(up) != (0, 0, 0) or act != L.ACT_NONE or rowvec is not None
                or res is not None or scale is not None or tile or splitk):
            raise L.CsError("taps-as-columns weights: plain 3x3x3 conv only (no stride / up / act / residual / tile)")
        return _conv_tapcol(x, w, spatial, a_scale, out, out_fn)
    xs = xp = None
    if isinstance(x, Pair16):
        if w.math != L.MATH_F16X3 or w.classes is not None:
            raise L.CsError("a Pair16 activation needs an (unfolded) F16X3-packed weight")
        xp, x = x, x.t
        _chk(x, "x")
    elif isinstance(x, Split16):
        if w.math != L.MATH_F16X3:
            raise L.CsError("a Split16 activation needs an F16X3-packed weight")
        xs, x = x, x.hi
        _chk(x, "x", torch.float16)
        _chk(xs.lo, "x.lo", torch.float16)
    else:
        _chk(x, "x")
    m, c, lda = rows_ld(x, "x")
    if c != w.cin_pad:
        raise L.CsError(f"x has {c} channels, packed weight expects {w.cin_pad}")
    kd, kh, kw = w.k
    pointwise = kd * kh * kw == 1 and tuple(stride) == (1, 1, 1) and tuple(up) == (0, 0, 0)
    if spatial is None:
        if x.dim() == 5:
            nb, d, h, wd = x.shape[:4]
        elif pointwise:
            nb, d, h, wd = m, 1, 1, 1
        else:
            raise L.CsError("conv needs x as [nb,d,h,w,c] or an explicit spatial=")
    else:
        nb, d, h, wd = spatial
        if nb * d * h * wd != m:
            raise L.CsError("spatial does not match x rows")
    pd, ph, pw = kd // 2, kh // 2, kw // 2
    vd, vh, vw = d << up[0], h << up[1], wd << up[2]
    do = (vd + 2 * pd - kd) // stride[0] + 1
    ho = (vh + 2 * ph - kh) // stride[1] + 1
    wo = (vw + 2 * pw - kw) // stride[2] + 1
    mo = nb * do * ho * wo
    ocols = w.cout // 2 if act == L.ACT_GEGLU else w.cout
    if out is None:
        if spatial is None and x.dim() != 5:
            oshape = (*x.shape[:-1], ocols)
        else:
            oshape = (nb, do, ho, wo, ocols)
        # out_fn: the caller places the result itself, e.g. as a channel slice of a wider (concatenation) buffer
        out = out_fn(oshape) if out_fn is not None else torch.empty(oshape, dtype=torch.float32, device=x.device)
        if tuple(out.shape) != tuple(oshape):
            out = out.view(oshape)
    _chk(out, "out")
    om, oc, ldo = rows_ld(out, "out")
    if om != mo or oc != ocols:
        raise L.CsError(f"out has shape {tuple(out.shape)}, expected {mo} rows x {w.cout}")
    p = L.CsConvGemm()
    math = w.math                      # the numerics mode is a property of how the weight was packed
    folded = w.classes is not None
    # F16X3 operand scale: the producer's (a Split16 carries it; `a_scale` for fp32 activations that come from a
    # normalisation, norm_a_scale) or the default 16 for operands of unknown range
    a_sc = float(xs.a_scale) if xs is not None else float(xp.a_scale) if xp is not None else float(a_scale or A_SCALE)
    if folded and a_sc != A_SCALE:
        raise L.CsError("folded Upsample convs read raw activations: operand scale must be the default")
    if folded and (tuple(up) != tuple(w.up) or tuple(stride) != (1, 1, 1) or tile or splitk):
        raise L.CsError(f"weight was folded for up={w.up}: conv_gemm must be called with that up, stride 1, no tile / splitk")
    if folded:
        p.x, p.out = x.data_ptr(), out.data_ptr()
        if math == L.MATH_F16X3:
            p.a_scale = A_SCALE
            if xs is not None:
                p.x_lo, p.a_format = xs.lo.data_ptr(), 1
            p.status = status_word(x.device).data_ptr()
    elif math == L.MATH_F16X3:
        p.x, p.w, p.w_lo, p.out = x.data_ptr(), w.wh.data_ptr(), w.wl.data_ptr(), out.data_ptr()
        p.acc_scale = w.acc_scale * (A_SCALE / a_sc)       # w.acc_scale = 1 / (weight scale * 16); powers of two: exact
        p.a_scale = a_sc
        if xs is not None:
            p.x_lo, p.a_format = xs.lo.data_ptr(), 1
        elif xp is not None:
            p.a_format = 2
        p.status = status_word(x.device).data_ptr()
    else:
        p.x, p.w, p.out = x.data_ptr(), w.wt.data_ptr(), out.data_ptr()
    p.bias = _ptr(w.bias)
    p.scale, p.shift = _ptr(scale), _ptr(shift)
    p.rowvec = _ptr(rowvec)
    p.res = _ptr(res)
    p.nb, p.din, p.hin, p.win = nb, d, h, wd
    p.dout, p.hout, p.wout = do, ho, wo
    p.cin, p.cout = w.cin_pad, w.cout
    p.lda, p.ldw, p.ldo = lda, w.ldw, ldo
    p.ldr = 0
    if res is not None:
        _chk(res, "res")
        rm, rc, ldr = rows_ld(res, "res")
        if rm != mo or rc != w.cout:
            raise L.CsError("res shape mismatch")
        p.ldr = ldr
    p.ldrv = 0
    if rowvec is not None:
        _chk(rowvec, "rowvec")
        vm, vc, ldrv = rows_ld(rowvec, "rowvec")
        if vc != w.cout or vm * rv_rows < mo:
            raise L.CsError("rowvec shape mismatch")
        p.ldrv = ldrv
    p.kd, p.kh, p.kw = kd, kh, kw
    p.sd, p.sh, p.sw = stride
    p.pd, p.ph, p.pw = pd, ph, pw
    p.ud, p.uh, p.uw = up
    if PINGPONG_OFF and tile == 0 and _pingpong_ok(mo, w.cin, w.cout, math, pointwise, scale is not None,
                                                   rv_rows if rowvec is not None else 0):
        tile = tile_for(mo, w.cout, 0, math, act=act)          # what the library picks without tile 5
    p.act, p.rv_rows, p.math, p.tile = act, rv_rows, math, tile
    lib = L.load()
    if folded:
        return _conv_gemm_up2(lib, p, w, x, out, mo)
    if splitk is not None and splitk > 1:            # explicit split-K factor (tuning / tests); the plan is bypassed
        ws = torch.empty((splitk * mo * w.cout,), dtype=torch.float32, device=x.device)
        p.splitk, p.splitk_ws = int(splitk), ws.data_ptr()
    elif math == L.MATH_F16X3 and tile == 0 and SPLITK and splitk is None:
        # few output tiles (small batches): let the library cut the K loop into slices; the partial tiles live in a
        # scratch tensor that the stream-ordered allocator may reuse as soon as this call's kernels are queued
        sk, wsb = C.c_int32(1), C.c_int64(0)
        L.check(lib.cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(wsb)), "cs_conv_gemm_plan")
        if sk.value > 1:
            ws = torch.empty((wsb.value // 4,), dtype=torch.float32, device=x.device)
            p.splitk, p.splitk_ws = sk.value, ws.data_ptr()
    prof = GEMM_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.cs_conv_gemm(C.byref(p), _stream()), "cs_conv_gemm")
    if prof is not None:
        e1.record()
        c3 = (kd, kh, kw) == (3, 3, 3) and tuple(stride) == (1, 1, 1) and tuple(up) == (0, 0, 0)
        tl = tile_for(mo, w.cout, tile, w.math, cin=w.cin, pointwise=pointwise, bn=scale is not None, act=act,
                      rv_rows=rv_rows if rowvec is not None else 0, conv3_win=wd if c3 else 0, presplit=xs is not None)
        if p.splitk > 1 and tl in (8, 9):
            tl -= 1 if tl == 8 else 3
        if p.splitk > 1 and math == L.MATH_F16X3:
            # K-sliced launches: the 256x224 tile for slab convs (r3), the 128x224 tile otherwise (cs_conv_gemm)
            tl = 4 if (c3 and wd <= 32 and not os.environ.get("CS_SLICE_TILE2")) or tl == 4 else 2
        prof.append(dict(e0=e0, e1=e1, flops=2.0 * mo * w.cout * w.cin * kd * kh * kw, taps=kd * kh * kw,
                         m=mo, n=w.cout, k=w.cin * kd * kh * kw, tile=tl,
                         slab=slab_width(tl, (kd, kh, kw), stride, up, wd, math, xs is not None, p.splitk),
                         pre=xs is not None, pair=xp is not None))
    return out


def _conv_gemm_up2(lib, p, w: PackedWeight, x: Tensor, out: Tensor, mo: int) -> Tensor:
    """upsample + conv on the source grid: one GEMM per output parity class into a scratch tensor, then the interleave
    (cs_conv_gemm_up2); the library splits K per class where the plan says so."""
    n = len(w.classes)
    wsb = lib.cs_conv_gemm_up2_ws_bytes(C.byref(p))
    if wsb <= 0:
        raise L.CsError("cs_conv_gemm_up2: descriptor not supported (residual / row vector / BN / odd cout)")
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=x.device)
    f16 = w.math == L.MATH_F16X3
    w_arr = (C.c_void_p * n)(*[(c.wh if f16 else c.wt).data_ptr() for c in w.classes])
    lo_arr = (C.c_void_p * n)(*[(c.wl.data_ptr() if f16 else 0) for c in w.classes])
    sc_arr = (C.c_float * n)(*[c.acc_scale for c in w.classes])
    prof = GEMM_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.cs_conv_gemm_up2(C.byref(p), w_arr, lo_arr, sc_arr, ws.data_ptr(), _stream()), "cs_conv_gemm_up2")
    if prof is not None:
        e1.record()
        kd, kh, kw = w.classes[0].k
        m1 = mo // n
        # executed multiply-adds (all classes); the direct form's 27-tap count is 27 / (kd * kh * kw) times this
        prof.append(dict(e0=e0, e1=e1, flops=2.0 * mo * w.cout * w.cin * kd * kh * kw, taps=kd * kh * kw, m=m1,
                         n=w.cout, k=w.cin * kd * kh * kw, tile=tile_for(m1, w.cout, 0, w.math, cin=w.cin)))
    return out


# Split-K for GEMMs with few output tiles (cs_conv_gemm_plan decides); CS_NO_SPLITK=1 turns it off (A/B runs).
SPLITK = not os.environ.get("CS_NO_SPLITK")

# Set to a list to collect one record per GEMM launch (HIP events on the launch stream): bench.py uses this
# to measure the dominant kernel's achieved TFLOP/s inside the timed region.
GEMM_PROFILE = None


# CS_NO_PINGPONG=1: token GEMMs on the one-tile-per-workgroup kernels (A/B runs; same bits either way)
PINGPONG_OFF = bool(os.environ.get("CS_NO_PINGPONG"))


def _pingpong_ok(m: int, cin: int, cout: int, math: int, pointwise: bool, bn: bool = False, rv_rows: int = 0) -> bool:
    """cs_pw_gemm_f16x3_applicable (csrc/cs_gemm_pw.hip), alignment conditions aside.  rv_rows: rows per row-vector
    entry when the GEMM has one (0: none)."""
    return (math == L.MATH_F16X3 and pointwise and not bn and cout % 224 == 0 and (cin + 15) // 16 >= 28
            and ((m + 127) // 128) * (cout // 224) >= 384 and rv_rows % 128 == 0)


TILE512 = os.environ.get("CS_TILE512", "") == "1"      # auto-select the 512-row slab tiles (A/B runs; off: they lost)


def tile_for(m: int, cout: int, tile: int = 0, math: int = L.MATH_FP32, cin: int = 0, pointwise: bool = False,
             bn: bool = False, act: int = L.ACT_NONE, rv_rows: int = 0, conv3_win: int = 0, presplit: bool = False) -> int:
    """mirror of the tile auto-selection in cs_conv_gemm (csrc/cs_gemm.hip).  conv3_win: the line width W of a 3x3x3
    stride-1 "same" conv (0 = any other geometry); presplit: the activations arrive as the fp16 hi / lo pair."""
    if tile:
        return tile
    t = _tile_for(m, cout, math, cin, act)
    # the 512-row slab tiles (two row blocks per wave): only with CS_TILE512=1 (cs_gemm.hip), pre-split operands only
    if conv3_win and presplit and TILE512 and math == L.MATH_F16X3:
        t512 = (m + 511) // 512
        if t == 7 and cout == 64 and conv3_win <= 64 and t512 >= 512:
            return 8
        if t == 6 and conv3_win <= 32 and t512 * (cout // 128) >= 512:
            return 9
    return t


def _tile_for(m: int, cout: int, math: int, cin: int, act: int) -> int:
    # tile 5 (the persistent ping-pong kernel) is never auto-selected: cs_pw_gemm_f16x3_preferred() is false
    mt = (m + 127) // 128
    if math == L.MATH_F16X3 and cout % 224 == 0 and ((m + 255) // 256) * (cout // 224) >= 192:
        return 2 if (act == L.ACT_GEGLU and 0 < (cin + 15) // 16 <= 32) else 4
    if math == L.MATH_F16X3 and cout % 128 == 0 and ((m + 255) // 256) * (cout // 128) >= 192:
        return 6 if act != L.ACT_GEGLU else 2
    if math == L.MATH_F16X3 and (cout == 64 or cout <= 4) and (m + 255) // 256 >= 192:
        return 7 if act != L.ACT_GEGLU else 2
    if cout % 224 == 0 and mt * (cout // 224) >= 256:
        t = 2
    elif cout > 64 and mt * ((cout + 127) // 128) >= 256:
        t = 1
    else:
        t = 3
    return 2 if act == L.ACT_GEGLU else t


def wants_split16(m: int, w: "PackedWeight") -> bool:
    """Should the GroupNorm feeding the conv `w` over m output rows emit the fp16 hi / lo operand pair?  Yes where the conv
    will run the slab kernel (3x3x3 on a 256-row tile: large batches) -- there the in-loop conversion is what is left
    to remove.  Small batches (128-row / 64-row tiles, split-K) measured slightly slower with it (7.85 vs 7.58 ms per
    one-object step), 1-tap GEMMs neutral: both keep fp32 activations."""
    if not (SPLIT16_PRODUCERS and w.math == L.MATH_F16X3 and w.classes is None and w.tapcol is None
            and tuple(w.k) == (3, 3, 3) and w.cin % 8 == 0):
        return False
    if _tile_for(m, w.cout, w.math, w.cin, L.ACT_NONE) in (4, 6, 7):
        return True
    # r3: also where the conv runs the 128-row slab tile (224-column convs of medium batches, with or without K slices)
    return SPLIT16_MIN_ROWS > 0 and w.cout % 224 == 0 and m >= SPLIT16_MIN_ROWS


def slab_width(tile: int, k, stride, up, win: int, math: int, presplit: bool, splitk: int) -> int:
    """mirror of the slab dispatch in cs_conv_gemm_f16x3_dispatch (csrc/cs_gemm_f16x3.hip): 0 = per-tap gather."""
    if splitk > 1 and tile != 4:
        tile = 2                                     # the K-sliced path runs 128x224 tiles (cs_conv_gemm)
    if (math != L.MATH_F16X3 or tuple(k) != (3, 3, 3) or tuple(stride) != (1, 1, 1)
            or tuple(up) != (0, 0, 0) or tile not in (2, 4, 6, 7, 8, 9) or win > 64):
        return 0
    if win <= 32:
        return 32
    return 64 if tile in (7, 8) else 0


def linear(x: Tensor, w: PackedWeight, **kw) -> Tensor:
    return conv_gemm(x, w, **kw)


@dataclass
class Split16:
    """An activation tensor as the fp16 hi / lo pair of value * A_SCALE (F16X3 A-operand format)."""
    hi: Tensor
    lo: Tensor
    a_scale: float = 16.0      # the power of two the values were multiplied by before the split (norm_a_scale)

    @property
    def shape(self):
        return self.hi.shape

    def dim(self):
        return self.hi.dim()

    def view(self, *shape):
        return Split16(self.hi.view(*shape), self.lo.view(*shape), self.a_scale)

    def __getitem__(self, idx):
        return Split16(self.hi[idx], self.lo[idx], self.a_scale)


@dataclass
class Pair16:
    """An activation tensor as the INTERLEAVED F16X3 operand pair (CsConvGemm.a_format = 2): `t` has the shape, dtype tag
    (float32) and bytes of the fp32 tensor it replaces; per row and 16-channel chunk it holds [hi c0-7 | lo c0-7 |
    hi c8-15 | lo c8-15] fp16 halves of value * a_scale.  Only a GEMM may read it."""
    t: Tensor
    a_scale: float = 16.0

    @property
    def shape(self):
        return self.t.shape

    def dim(self):
        return self.t.dim()

    def view(self, *shape):
        if shape[-1] != self.t.shape[-1]:
            raise L.CsError("Pair16.view: the channel dimension cannot be reshaped")
        return Pair16(self.t.view(*shape), self.a_scale)


# CS_NO_PAIR16=1: LayerNorm keeps emitting fp32 (A/B runs; bit-identical either way)
PAIR16_PRODUCERS = not os.environ.get("CS_NO_PAIR16", "")


# Producer-side operand split: GroupNorm writes the fp16 hi/lo pair of y * A_SCALE (same bytes as fp32 y) and the GEMM
# DMA-loads it with a_format=1, so its K loop carries no conversion VALU.  With the per-tap gather kernels this was
# neutral (108.0 vs 107.5 ms/step in round 1: the conversion hid under the DMA-bound loop); on the slab kernel the
# conversion is what is left -- a timing-only build without it ran the conv shapes at 472-475 instead of 402-405 TF/s
# (tools/slab_whatif2.sh) -- so it is the default.  Results are bit-identical either way (y * 16 is exact).
# CS_NO_SPLIT16=1 turns it off (A/B runs).
SPLIT16_PRODUCERS = not os.environ.get("CS_NO_SPLIT16")
# ... and (r3) where it runs the 128-row slab tile, from this many rows (4 objects at 16x8x8): 7 objects 27.34 -> 26.94
# ms/step, 4 objects 17.61 -> 17.41, 16 objects 49.11 -> 48.75; below it slower (1 object 8.58 -> 8.73): the three-launch
# GroupNorm replaces the single-launch one there (profiles/r03_j_split16_small_ab.txt).  0 = never.
SPLIT16_MIN_ROWS = int(os.environ.get("CS_SPLIT16_MIN_ROWS", "8192"))


def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, act: int = L.ACT_NONE,
              out: Optional[Tensor] = None, split16: bool = False, a_scale: Optional[float] = None):
    """GroupNorm over [nb, ..., c] (stats per sample & group), fused activation.
    split16=True returns a Split16 (fp16 hi/lo pair, pre-scaled) for an F16X3 GEMM to consume."""
    _chk(x, "x")
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    rows = m // nb
    lib = L.load()
    ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device=x.device)
    stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
    if split16 and SPLIT16_PRODUCERS:
        L.check(lib.cs_groupnorm_stats(x.data_ptr(), nb, rows, c, ldx, groups, eps, ws.data_ptr(),
                                       stats.data_ptr(), _stream()), "cs_groupnorm_stats")
        yh = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        yl = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        a_sc = float(a_scale or A_SCALE)
        L.check(lib.cs_groupnorm_apply_split16(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               yh.data_ptr(), yl.data_ptr(), nb, rows, c, ldx, c, groups, act,
                                               a_sc, status_word(x.device).data_ptr(), _stream()),
                "cs_groupnorm_apply_split16")
        return Split16(yh, yl, a_sc)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    om, oc, ldy = rows_ld(out, "out")
    if om != m or oc != c:
        raise L.CsError("groupnorm out shape mismatch")
    # one launch for small tensors (one or two objects), statistics + apply otherwise: cs_groupnorm decides
    L.check(lib.cs_groupnorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), nb, rows, c, ldx, ldy,
                             groups, eps, act, ws.data_ptr(), stats.data_ptr(), _stream()), "cs_groupnorm")
    return out


def groupnorm_stats(x: Tensor, groups: int, eps: float) -> Tensor:
    """(mean, rstd) per (sample, group) of a channels-last tensor: [nb, groups, 2] fp32 (fp64 accumulation)."""
    _chk(x, "x")
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    lib = L.load()
    ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device=x.device)
    stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
    L.check(lib.cs_groupnorm_stats(x.data_ptr(), nb, m // nb, c, ldx, groups, eps, ws.data_ptr(), stats.data_ptr(),
                                   _stream()), "cs_groupnorm_stats")
    return stats


def groupnorm_apply_range(x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, cpg: int, ch0: int,
                          act: int = L.ACT_NONE, split16: bool = False, a_scale: Optional[float] = None):
    """Normalise + affine + activation of a CHANNEL RANGE: x (and gamma / beta) hold channels ch0 .. ch0 + c of a tensor
    whose statistics `stats` [nb', groups, 2] were taken over groups of cpg channels (cs_groupnorm_apply_range); sample n
    of x uses stats[n].  split16=True returns the Split16 operand pair."""
    _chk(x, "x"); _chk(stats, "stats")
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    groups = stats.shape[1]
    if stats.shape[0] < nb or not stats.is_contiguous() or gamma.numel() != c or beta.numel() != c:
        raise L.CsError("groupnorm_apply_range: stats / gamma / beta do not match x")
    lib = L.load()
    if split16 and SPLIT16_PRODUCERS:
        yh = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        yl = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        a_sc = float(a_scale or A_SCALE)
        L.check(lib.cs_groupnorm_apply_split16_range(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     yh.data_ptr(), yl.data_ptr(), nb, m // nb, c, ldx, c, groups, cpg,
                                                     ch0, act, a_sc, status_word(x.device).data_ptr(), _stream()),
                "cs_groupnorm_apply_split16_range")
        return Split16(yh, yl, a_sc)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(lib.cs_groupnorm_apply_range(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                         nb, m // nb, c, ldx, c, groups, cpg, ch0, act, _stream()),
            "cs_groupnorm_apply_range")
    return y


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, out: Optional[Tensor] = None,
              pair_scale: Optional[float] = None):
    """nn.LayerNorm over the last dim.  pair_scale=s: return the result as the Pair16 operand pair of y * s (for an F16X3
    GEMM; c % 16 == 0) instead of an fp32 tensor."""
    _chk(x, "x")
    m, c, ldx = rows_ld(x, "x")
    if pair_scale is not None and PAIR16_PRODUCERS and c % 16 == 0:
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        L.check(L.load().cs_layernorm_pair16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), m, c, ldx, c,
                                             eps, float(pair_scale), status_word(x.device).data_ptr(), _stream()),
                "cs_layernorm_pair16")
        return Pair16(y, float(pair_scale))
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _, _, ldy = rows_ld(out, "out")
    L.check(L.load().cs_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), m, c,
                                  ldx, ldy, eps, _stream()), "cs_layernorm")
    return out


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, out: Optional[Tensor] = None,
              math: int = L.MATH_FP32) -> Tensor:
    """q: [nb, nq, heads*dh] (views with wider row stride allowed), k/v: [nb, nk, heads*dh]."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n)
        if t.dim() != 3:
            raise L.CsError(f"{n} must be [nb, n, c]")
    nb, nq, cq = q.shape
    nk = k.shape[1]
    dh = cq // heads
    _, _, ldq = rows_ld(q, "q")
    _, _, ldk = rows_ld(k, "k")
    _, _, ldv = rows_ld(v, "v")
    if out is None:
        out = torch.empty((nb, nq, cq), dtype=torch.float32, device=q.device)
    _, _, ldo = rows_ld(out, "out")
    lib = L.load()
    if math == L.MATH_F16X3:
        # K / V split once per call into their LDS tile images where the library has that path (ws_bytes > 0)
        wsb = lib.cs_attn_f16x3_ws_bytes(nb, nq, nk, heads, dh)
        ws = torch.empty((wsb // 4,), dtype=torch.float32, device=q.device) if wsb > 0 else None
        L.check(lib.cs_attn_selfattn_f16x3_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nb, nq, nk, heads,
                                              dh, ldq, ldk, ldv, ldo, scale, status_word(q.device).data_ptr(), _ptr(ws),
                                              _stream()), "cs_attn_selfattn_f16x3_ws")
    elif math == L.MATH_F16:
        L.check(lib.cs_attn_selfattn_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nb, nq, nk, heads, dh,
                                         ldq, ldk, ldv, ldo, scale, status_word(q.device).data_ptr(), _stream()),
                "cs_attn_selfattn_f16")
    else:
        L.check(lib.cs_attn_selfattn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nb, nq, nk, heads, dh,
                                     ldq, ldk, ldv, ldo, scale, _stream()), "cs_attn_selfattn")
    return out


def geglu(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _chk(x, "x")
    m, c2, ldx = rows_ld(x, "x")
    h = c2 // 2
    if out is None:
        out = torch.empty((*x.shape[:-1], h), dtype=torch.float32, device=x.device)
    _, _, ldo = rows_ld(out, "out")
    L.check(L.load().cs_geglu(x.data_ptr(), out.data_ptr(), m, h, ldx, ldo, _stream()), "cs_geglu")
    return out


def copy_rows(src: Tensor, dst: Tensor) -> Tensor:
    _chk(src, "src"); _chk(dst, "dst")
    m, c, lds = rows_ld(src, "src")
    dm, dc, ldd = rows_ld(dst, "dst")
    if dm != m or dc != c:
        raise L.CsError("copy_rows shape mismatch")
    L.check(L.load().cs_copy_rows(src.data_ptr(), dst.data_ptr(), m, c, lds, ldd, _stream()), "cs_copy_rows")
    return dst


def concat_channels(a: Tensor, b: Tensor) -> Tensor:
    """torch.cat([a, b], channel) for channels-last tensors (openai_model_3d.py:781)."""
    ca, cb = a.shape[-1], b.shape[-1]
    out = torch.empty((*a.shape[:-1], ca + cb), dtype=torch.float32, device=a.device)
    copy_rows(a, out[..., :ca])
    if b.shape[0] == a.shape[0]:
        copy_rows(b, out[..., ca:])
    elif a.shape[0] % b.shape[0] == 0:           # b shared by a.shape[0] / b.shape[0] groups of samples (CFG pairs)
        nb = b.shape[0]
        for g in range(a.shape[0] // nb):
            copy_rows(b, out[g * nb:(g + 1) * nb, ..., ca:])
    else:
        raise L.CsError("concat_channels: batch mismatch")
    return out


def add_rowvec_(x: Tensor, v: Tensor, rows: int) -> Tensor:
    _chk(x, "x"); _chk(v, "v")
    m, c, ldx = rows_ld(x, "x")
    _, vc, ldv = rows_ld(v, "v")
    if vc != c:
        raise L.CsError("add_rowvec shape mismatch")
    L.check(L.load().cs_add_rowvec(x.data_ptr(), v.data_ptr(), m, c, ldx, ldv, rows, _stream()), "cs_add_rowvec")
    return x


def nchw_to_ndhwc(x: Tensor, cpad: Optional[int] = None) -> Tensor:
    """[nb, c, d, h, w] -> [nb, d, h, w, cpad] (zero padded channels)."""
    _chk(x, "x")
    x = x.contiguous()
    nb, c = x.shape[:2]
    sp = tuple(x.shape[2:])
    s = 1
    for v in sp:
        s *= v
    cp = cpad if cpad is not None else c
    y = torch.empty((nb, *sp, cp), dtype=torch.float32, device=x.device)
    L.check(L.load().cs_nchw_to_ndhwc(x.data_ptr(), y.data_ptr(), nb, c, s, cp, _stream()), "cs_nchw_to_ndhwc")
    return y


def ndhwc_to_nchw(x: Tensor, c: Optional[int] = None) -> Tensor:
    """[nb, d, h, w, ld] -> [nb, c, d, h, w] keeping the first c channels."""
    _chk(x, "x")
    m, cx, ldx = rows_ld(x, "x")
    nb = x.shape[0]
    sp = tuple(x.shape[1:-1])
    cc = c if c is not None else cx
    y = torch.empty((nb, cc, *sp), dtype=torch.float32, device=x.device)
    L.check(L.load().cs_ndhwc_to_nchw(x.data_ptr(), y.data_ptr(), nb, cc, m // nb, ldx, _stream()),
            "cs_ndhwc_to_nchw")
    return y


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    _chk(t, "t", torch.int64)
    t = t.contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    L.check(L.load().cs_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, max_period, _stream()),
            "cs_timestep_embedding")
    return out


def ddim_cfg_update(x: Tensor, eps: Tensor, a_t: float, a_prev: float, sigma_t: float,
                    sqrt_one_minus_at: float, cfg_scale: float, cfg: bool,
                    noise: Optional[Tensor] = None, want_pred_x0: bool = True,
                    out: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    _chk(x, "x"); _chk(eps, "eps")
    x = x.contiguous(); eps = eps.contiguous()
    nb = x.shape[0]
    per = x.numel() // nb
    if eps.numel() != (2 if cfg else 1) * x.numel():
        raise L.CsError("eps must hold [uc; c] halves when cfg is on")
    xp = out if out is not None else torch.empty_like(x)
    p0 = torch.empty_like(x) if want_pred_x0 else None
    if noise is not None:
        noise = noise.contiguous()
    L.check(L.load().cs_ddim_cfg_update(x.data_ptr(), eps.data_ptr(), _ptr(noise), xp.data_ptr(), _ptr(p0),
                                        nb, per, a_t, a_prev, sigma_t, sqrt_one_minus_at, cfg_scale,
                                        1 if cfg else 0, _stream()), "cs_ddim_cfg_update")
    return xp, p0


PLMS_PLAIN, PLMS_AB2, PLMS_AB3, PLMS_AB4, PLMS_EULER_AVG = 0, 1, 2, 3, 4


def plms_update(x: Tensor, eps: Tensor, hist: Sequence[Tensor], mode: int, a_t: float, a_prev: float,
                sqrt_one_minus_at: float, cfg_scale: float, cfg: bool, want_e: bool = True,
                want_pred_x0: bool = True) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """cs_plms_update: (x_prev, pred_x0, e_t).  `hist` = earlier noise predictions, newest first."""
    _chk(x, "x"); _chk(eps, "eps")
    x = x.contiguous(); eps = eps.contiguous()
    nb = x.shape[0]
    per = x.numel() // nb
    if eps.numel() != (2 if cfg else 1) * x.numel():
        raise L.CsError("eps must hold [uc; c] halves when cfg is on")
    hs = [h.contiguous() for h in hist]
    for h in hs:
        _chk(h, "hist")
        if h.numel() != x.numel():
            raise L.CsError("history entries must have x's shape")
    hp = [h.data_ptr() for h in hs] + [None] * (3 - len(hs))
    xp = torch.empty_like(x)
    p0 = torch.empty_like(x) if want_pred_x0 else None
    e = torch.empty_like(x) if want_e else None
    L.check(L.load().cs_plms_update(x.data_ptr(), eps.data_ptr(), hp[0], hp[1], hp[2], _ptr(e), xp.data_ptr(), _ptr(p0),
                                    nb, per, mode, a_t, a_prev, sqrt_one_minus_at, cfg_scale, 1 if cfg else 0,
                                    _stream()), "cs_plms_update")
    return xp, p0, e


def ddim_coefficients(a_t: float, a_prev: float, sigma_t: float, sqrt_one_minus_at: float) -> Tuple[float, ...]:
    """Host helper: the five fp32 coefficients cs_ddim_cfg_update derives from its scalar arguments."""
    buf = (C.c_float * 5)()
    L.check(L.load().cs_ddim_coefficients(a_t, a_prev, sigma_t, sqrt_one_minus_at,
                                          C.cast(buf, C.c_void_p).value), "cs_ddim_coefficients")
    return tuple(float(v) for v in buf)


def ddim_cfg_update_dev(x: Tensor, eps: Tensor, coef: Tensor, cfg_scale: float, cfg: bool,
                        noise: Optional[Tensor] = None, pred_x0: Optional[Tensor] = None,
                        out: Optional[Tensor] = None) -> Tensor:
    """ddim_cfg_update with the step coefficients (ddim_coefficients(...)) in a 5-float device tensor."""
    _chk(x, "x"); _chk(eps, "eps"); _chk(coef, "coef")
    if not (x.is_contiguous() and eps.is_contiguous()) or coef.numel() < 5:
        raise L.CsError("ddim_cfg_update_dev needs contiguous x / eps and a 5-float coefficient block")
    nb = x.shape[0]
    per = x.numel() // nb
    if eps.numel() != (2 if cfg else 1) * x.numel():
        raise L.CsError("eps must hold [uc; c] halves when cfg is on")
    xp = out if out is not None else torch.empty_like(x)
    L.check(L.load().cs_ddim_cfg_update_dev(x.data_ptr(), eps.data_ptr(), _ptr(noise), xp.data_ptr(),
                                            _ptr(pred_x0), nb, per, coef.data_ptr(), cfg_scale,
                                            1 if cfg else 0, _stream()), "cs_ddim_cfg_update_dev")
    return xp


def vq_lookup(z: Tensor, codebook: Tensor) -> Tuple[Tensor, Tensor]:
    """z: [..., ld>=edim] rows (first edim columns used); returns (idx int64 [M], zq [..., edim_pad])."""
    _chk(z, "z"); _chk(codebook, "codebook")
    m, c, ldz = rows_ld(z, "z")
    ncode, edim = codebook.shape
    idx = torch.empty((m,), dtype=torch.int64, device=z.device)
    zq = torch.zeros(z.shape, dtype=torch.float32, device=z.device)
    _, _, ldq = rows_ld(zq, "zq")
    L.check(L.load().cs_vq_argmin_lookup(z.data_ptr(), codebook.contiguous().data_ptr(), idx.data_ptr(),
                                         zq.data_ptr(), m, ncode, edim, ldz, ldq, _stream()),
            "cs_vq_argmin_lookup")
    return idx, zq


def gcn_gather_cat(obj: Tensor, pred: Tensor, edges: Tensor) -> Tensor:
    _chk(obj, "obj"); _chk(pred, "pred"); _chk(edges, "edges", torch.int64)
    obj = obj.contiguous(); pred = pred.contiguous(); edges = edges.contiguous()
    n_obj, d_obj = obj.shape
    n_tri, d_pred = pred.shape
    out = torch.empty((n_tri, 2 * d_obj + d_pred), dtype=torch.float32, device=obj.device)
    err = index_err_word(obj.device)
    L.check(L.load().cs_gcn_gather_cat(obj.data_ptr(), pred.data_ptr(), edges.data_ptr(), out.data_ptr(),
                                       n_obj, n_tri, d_obj, d_pred, err.data_ptr(), _stream()),
            "cs_gcn_gather_cat")
    return out


def gcn_segment_mean(new_t: Tensor, edges: Tensor, n_obj: int, h: int, off_o: int) -> Tensor:
    _chk(new_t, "new_t"); _chk(edges, "edges", torch.int64)
    n_tri, _, ld_t = rows_ld(new_t, "new_t")
    pooled = torch.empty((n_obj, h), dtype=torch.float32, device=new_t.device)
    err = index_err_word(new_t.device)
    L.check(L.load().cs_gcn_segment_mean(new_t.data_ptr(), edges.contiguous().data_ptr(), pooled.data_ptr(),
                                         n_obj, n_tri, h, off_o, ld_t, err.data_ptr(), _stream()),
            "cs_gcn_segment_mean")
    return pooled


def gcn_csr(edges: Tensor, n_obj: int) -> Tensor:
    """CSR-by-destination index of a scene graph's edges [T, 2] (built once, shared by every layer's pooling)."""
    _chk(edges, "edges", torch.int64)
    edges = edges.contiguous()
    n_tri = edges.shape[0]
    lib = L.load()
    csr = torch.empty((int(lib.cs_gcn_csr_ints(n_obj, n_tri)),), dtype=torch.int32, device=edges.device)
    L.check(lib.cs_gcn_csr_build(edges.data_ptr(), csr.data_ptr(), n_obj, n_tri, index_err_word(edges.device).data_ptr(),
                                 _stream()), "cs_gcn_csr_build")
    return csr


def gcn_segment_mean_csr(new_t: Tensor, csr: Tensor, n_obj: int, h: int, off_o: int) -> Tensor:
    _chk(new_t, "new_t"); _chk(csr, "csr", torch.int32)
    _, _, ld_t = rows_ld(new_t, "new_t")
    pooled = torch.empty((n_obj, h), dtype=torch.float32, device=new_t.device)
    L.check(L.load().cs_gcn_segment_mean_csr(new_t.data_ptr(), csr.data_ptr(), pooled.data_ptr(), n_obj, h, off_o, ld_t,
                                             _stream()), "cs_gcn_segment_mean_csr")
    return pooled


def embedding(table: Tensor, idx: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _chk(table, "table"); _chk(idx, "idx", torch.int64)
    n_rows, dim = table.shape
    n = idx.numel()
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=table.device)
    _, oc, ldo = rows_ld(out, "out")
    err = index_err_word(table.device)
    L.check(L.load().cs_embedding(table.contiguous().data_ptr(), idx.contiguous().data_ptr(), out.data_ptr(),
                                  n, dim, n_rows, ldo, err.data_ptr(), _stream()), "cs_embedding")
    return out


def log_softmax(x: Tensor) -> Tensor:
    _chk(x, "x")
    m, c, ldx = rows_ld(x, "x")
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _, _, ldy = rows_ld(y, "y")
    L.check(L.load().cs_log_softmax(x.data_ptr(), y.data_ptr(), m, c, ldx, ldy, _stream()), "cs_log_softmax")
    return y

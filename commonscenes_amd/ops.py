"""Thin torch-tensor front end over the C ABI (include/commonscenes_hip.h).

torch is used for device memory, streams and shapes only; every arithmetic op below is one call
into libcommonscenes_hip.so on the current HIP stream.  Tensors are fp32, channels-last
(`[nb, d, h, w, c]` or token matrices `[nb, n, c]`), possibly views into a wider buffer (row stride
`ld` > c) so channel concatenation needs no extra pass.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
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


# Split-K arrival counters (CsConvGemm.splitk_sync, ABI 15): zeroed once per (device, stream); the kernels that use them
# return them to zero, so one buffer serves every K-sliced launch queued on that stream.
_SYNC: dict = {}
SYNC_WORDS = 8192


def sync_words() -> Tensor:
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    t = _SYNC.get(key)
    if t is None:
        t = _SYNC[key] = torch.zeros((SYNC_WORDS,), dtype=torch.int32, device=torch.device("cuda", key[0]))
    return t


def check_overflow(device=None, what: str = "F16X3 kernels") -> None:
    """Raise CsOverflowError if any F16X3 kernel since the last check met an operand beyond the fp16 range: a value a
    is carried as fp16 halves of a * a_scale, where a_scale is the layer's operand scale -- derived from the producing
    normalisation's bound (norm_a_scale: cannot overflow) or 16 for raw activations (|a| >= 65504 / 16 ~ 4094
    overflows).  That launch's output is garbage; the caller re-runs with set_math('fp32').
    CS_STATUS_INTERNAL (a kernel was asked for an epilogue output on a path that cannot produce it: a planning bug, never
    a data condition) raises CsError -- no fall-back hides it.  CS_STATUS_SPLITK_TIMEOUT (a fused split-K launch was not
    resident: CUs held by another stream or masked) raises CsSplitKTimeout; the model classes answer it by re-running with
    the two-kernel split-K, which needs no co-residency."""
    st = read_status(device)
    if st & (L.STATUS_INTERNAL | L.STATUS_SPLITK_TIMEOUT):
        for t in _SYNC.values():          # the counters of an aborted hand-off may be left non-zero
            t.zero_()
    if st & L.STATUS_INTERNAL:
        raise L.CsError(f"{what}: CS_STATUS_INTERNAL -- a kernel reached an epilogue path that cannot emit what its "
                        "descriptor asked for (GroupNorm partials / operand pair); "
                        "results of this run are invalid (library planning bug)")
    if st & L.STATUS_SPLITK_TIMEOUT:
        raise L.CsSplitKTimeout(f"{what}: CS_STATUS_SPLITK_TIMEOUT -- a fused split-K launch was not resident (a slice never "
                                "reached a CU before its tile's reducers gave up); results of this run are invalid -- re-run "
                                "with the two-kernel split-K (lib.debug_set(no_fused_reduce=1); the model classes do)")
    if st & L.STATUS_F16X3_OVERFLOW:
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
    # r5: the Winograd-W packs of a 3x3x3 conv (pack_weight_wino), by variant (2 = F(2,3): four position images, 4 = F(4,3):
    # six): {variant: (hi, lo, accumulator scale)} with images [positions][9 taps][cin16/8][cout][8]; None = direct form only
    wino: Optional[dict] = None


def pack_weight(w: Tensor, bias: Optional[Tensor] = None, cin_pad: Optional[int] = None,
                math: int = L.MATH_FP32, fold_up: Optional[Sequence[int]] = None,
                amax: Optional[float] = None) -> PackedWeight:
    """torch Conv3d (cout,cin,kd,kh,kw) or Linear (out,in) weight -> PackedWeight (device op).
    fold_up=(ud,uh,uw): the conv follows a nearest x2 upsampling of the flagged dims; its taps are pre-summed per output
    parity class (cs_fold_upsample_weight) and conv_gemm(..., up=fold_up) runs on the source grid."""
    if fold_up is not None and any(fold_up) and _sw("FOLD_UPSAMPLE"):
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
    statistic: the largest power of two that keeps the producer's bound |y| <= gmax * sqrt(n - 1) + bmax inside the fp16
    range -- it cannot overflow WHATEVER the input (r3).  The rule itself is cs_norm_a_scale (csrc/cs_plan.hip): ONE
    implementation, called by this host and by the native drivers (r4; r3 kept a Python copy here)."""
    return float(L.load().cs_norm_a_scale(float(gmax), float(bmax), int(n)))


# ---- switches ---------------------------------------------------------------------------------------------------------
# r4: the CS_* A/B switches are ONE struct parsed once by the library (include/commonscenes_hip.h CsDebug, lib.debug());
# the module attributes below are views of it, kept under their r1-r3 names.  A test may still assign one
# (monkeypatch.setattr(ops, "SPLITK", False)): the assignment shadows the view for this module's own reads too.
_SWITCHES = {
    "FOLD_UPSAMPLE": lambda d: not d.no_upfold,          # CS_NO_UPFOLD=1: Upsample convs in direct form (27 taps, doubled grid)
    "TAPCOL": lambda d: not d.no_tapcol,                 # CS_NO_TAPCOL=1: thin-output convs stay on the implicit GEMM
    "SPLITK": lambda d: not d.no_splitk,                 # CS_NO_SPLITK=1: never ask cs_conv_gemm_plan for K slices
    "PAIR16_PRODUCERS": lambda d: not d.no_pair16,       # CS_NO_PAIR16=1: LayerNorm keeps emitting fp32
    "SPLIT16_PRODUCERS": lambda d: not d.no_split16,     # CS_NO_SPLIT16=1: GroupNorm emits fp32, not the operand pair
    "PAIR_EPILOGUES": lambda d: not d.no_pair_epilogue,  # CS_NO_PAIR_EPILOGUE=1: GEMM epilogues always write fp32
    "GN_PARTS": lambda d: not d.no_gn_parts,             # CS_NO_GN_PARTS=1: GroupNorm statistics from a pass over the tensor
    "DYN_SCALE": lambda d: not d.no_dyn_scale,           # CS_NO_DYN_SCALE=1: raw-activation consumers keep the fixed scale 16
    "STATIC_SCALES": lambda d: not d.no_static_scales,   # CS_NO_STATIC_SCALES=1: transformer-internal operands keep 16 + flag
}


def _sw(name: str):
    """The switch's value: the library's CsDebug view (lib.debug_override(...) / cs_debug_set reach every host at once),
    unless a module global of the same name shadows it -- `ops.SPLITK = False` / monkeypatch.setattr(ops, "SPLITK", True)
    force the switch off / ON for this module's reads (also against the environment).  Reading never writes (ADVICE r5:
    r4's version deleted a truthy override on read, so a forced-on switch was silently dropped); `reset_switches()`
    removes every shadow -- the tests' autouse fixture calls it, so a leftover override cannot outlive its test."""
    g = globals()
    if name in g:
        return bool(g[name])
    return _SWITCHES[name](L.debug())


def reset_switches() -> None:
    """drop every module-global shadow of a CsDebug switch (back to the live view)"""
    g = globals()
    for name in _SWITCHES:
        g.pop(name, None)


def __getattr__(name: str):          # PEP 562: ops.SPLITK etc. for outside readers
    if name in _SWITCHES:
        return _sw(name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")




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


def pack_weight_wino(pw: PackedWeight, w: Tensor, amax: Optional[float] = None) -> PackedWeight:
    """Add the Winograd-W pack (CsConvGemm.a_format = 3, cs_pack_weight_f16x3_wino) to the F16X3 pack `pw` of the 3x3x3 conv
    weight `w` -- where the geometry can ever take that route (a 224-column width, or the VQ decoder's 64 / multiples of
    128; cin % 8 == 0); the per-call decision is cs_conv_wino_ok (wants_wino).  One power-of-two scale for the four
    positions: max |u_q| <= 1.5 max |w|.  `amax`: the magnitude of the WHOLE tensor `w` is an input-channel slice of (the
    channel-split convs: both hosts scale their halves by the whole tensor's maximum)."""
    import math as _m
    if (pw.math != L.MATH_F16X3 or pw.classes is not None or pw.tapcol is not None or w.dim() != 5
            or tuple(w.shape[2:]) != (3, 3, 3) or not (pw.cout % 224 == 0 or pw.cout % 128 == 0 or pw.cout == 64)
            or pw.cin % 8 or pw.cin < 16):
        return pw
    w = w.contiguous()
    amax0 = float(w.abs().max().item()) if amax is None else float(amax)
    kg = (pw.cin + 15) // 16 * 2
    pw.wino = {}
    # F(2,3): max |u_q| <= 1.5 max |w|; F(4,3): <= max |w|
    for variant, grow in ((2, 1.5), (4, 1.0)):
        am = grow * amax0
        e = _m.frexp(am)[1] if am > 0 and _m.isfinite(am) else 0
        scale = 2.0 ** (14 - e)
        wh = torch.empty((variant + 2, 9, kg, pw.cout, 8), dtype=torch.float16, device=w.device)
        wl = torch.empty_like(wh)
        L.check(L.load().cs_pack_weight_f16x3_wino_v(w.data_ptr(), wh.data_ptr(), wl.data_ptr(), pw.cout, pw.cin, scale, variant,
                                                     0, 0, _stream()), "cs_pack_weight_f16x3_wino_v")
        pw.wino[variant] = (wh, wl, 1.0 / (scale * A_SCALE))
    return pw


def tapcol_ok(w: Tensor, math: int) -> bool:
    """is this (named) thin-output 3x3x3 conv run as taps-as-columns?  cs_tapcol_ok: the rule both hosts ask"""
    if not _sw("TAPCOL") or w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        return False
    return bool(L.load().cs_tapcol_ok(int(w.shape[0]), int(w.shape[1]), 3, int(math)))


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
    """tile of the taps-as-columns GEMM: cs_tapcol_tile (csrc/cs_plan.hip), the rule both hosts ask"""
    return int(L.load().cs_tapcol_tile(int(m), int(ncolp)))


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


@dataclass
class ColStats:
    """Per-(row tile, column) fp64 (sum, sum of squares) of a GEMM's output, written by its epilogue
    (CsConvGemm.gn_part): what the GroupNorm that follows needs instead of a pass over the tensor."""
    part: Tensor          # [tiles, ld, 2] float64
    nch: int              # columns (= the producer's cout)
    nb: int               # samples the producer ran
    tps: int              # statistics tiles per sample (and per class)
    ncls: int = 1         # parity classes of a folded Upsample launch (tiles ordered [class][sample][tile])
    owner: Optional[Tuple[int, int]] = None     # (data_ptr, _version) of the tensor the partials describe (attach_stats)


def _forget_stats(t: Tensor) -> None:
    for a in ("cs_stats", "cs_segs", "cs_bound"):
        if hasattr(t, a):
            delattr(t, a)


def attach_stats(t: Tensor, st) -> Tensor:
    """Remember the producer's partials on the tensor object (a plain attribute: views made later do not inherit it).
    Whatever an EARLIER launch into the same tensor object left -- partials, segments, a magnitude bound -- is dropped first:
    a caller-supplied `out=` buffer that is reused must never keep statistics of its previous contents.  The
    partials remember the tensor's (data_ptr, _version): stats_segments() refuses them after an in-place torch update."""
    _forget_stats(t)
    if st is not None:
        t.cs_stats = dataclasses.replace(st, owner=(t.data_ptr(), t._version))
    return t


def _stats_fresh(x: Tensor, st) -> bool:
    # (a view made by the sequencer shares storage offset 0 .. and the version counter with its base: data_ptr may differ
    # for a re-attached view, so only the version is compared when the pointer moved inside the same storage)
    return st is not None and (st.owner is None or st.owner[1] == x._version)


def stats_segments(x: Tensor):
    """[(first channel, ColStats)] covering x's channels in order, or None: `cs_segs` (a concatenation whose halves came
    from two producers) or `cs_stats` (one producer)."""
    if not _sw("GN_PARTS"):
        return None
    segs = getattr(x, "cs_segs", None)
    if segs is None:
        st = getattr(x, "cs_stats", None)
        segs = [(0, st)] if st is not None else None
    if segs is None or any(s is None for _, s in segs) or sum(s.nch for _, s in segs) != x.shape[-1]:
        return None
    st1 = getattr(x, "cs_stats", None)
    if st1 is not None and getattr(x, "cs_segs", None) is None and not _stats_fresh(x, st1):
        return None                          # x was updated in place after its producer left the partials
    return segs


def _seg_array(segs):
    arr = (L.CsGnSeg * len(segs))()
    for i, (ch0, st) in enumerate(segs):
        arr[i].part, arr[i].ld, arr[i].col0 = st.part.data_ptr(), int(st.part.shape[1]), 0
        arr[i].ch0, arr[i].nch = int(ch0), int(st.nch)
        arr[i].tiles_per_sample, arr[i].ncls, arr[i].nb_src = int(st.tps), int(st.ncls), int(st.nb)
    return arr


def range_bound(x: Tensor, slot: Optional[Tensor], groups: int = 32) -> Optional[Tensor]:
    """Leave an upper bound of max |x| in `slot` (a zeroed 1-element fp32 device tensor) -- the maximum over x's (sample,
    group) statistics of |mean| + std * sqrt(n - 1) (Samuelson), computed from the producers' partial sums by the tiny
    finalize kernel -- and remember it on x (`x.cs_bound`).  A consumer of the RAW tensor hands it to
    conv_gemm(x_bound=): its F16X3 operand scale then follows the tensor's actual range (CsConvGemm.a_bound) instead of
    the fixed guess 16.  For tensors that are NOT followed by a GroupNorm (the inputs of Downsample / Upsample); a
    GroupNorm over x leaves the bound as a by-product (groupnorm(..., bound=slot)).  A tensor without partials gets it from
    one statistics pass (cs_groupnorm_stats_bound).  None when the feature is off or the channels do not divide."""
    if slot is None or not _sw("DYN_SCALE") or x.shape[-1] % groups:
        return None
    segs = stats_segments(x)
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    lib = L.load()
    if segs is not None:
        L.check(lib.cs_groupnorm_finalize_parts(_seg_array(segs), len(segs), nb, m // nb, c, groups, 1e-5, None,
                                                slot.data_ptr(), _stream()), "cs_groupnorm_finalize_parts")
    else:       # no partials (e.g. a folded Upsample conv at a small batch): one statistics pass over the (small) tensor
        ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device=x.device)
        st = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
        L.check(lib.cs_groupnorm_stats_bound(x.data_ptr(), nb, m // nb, c, ldx, groups, 1e-5, ws.data_ptr(), st.data_ptr(),
                                             slot.data_ptr(), _stream()), "cs_groupnorm_stats_bound")
    x.cs_bound = slot
    return slot


def groupnorm_stats_from_parts(segs, nb: int, rows: int, c: int, groups: int, eps: float, device,
                               bound: Optional[Tensor] = None) -> Tensor:
    """(mean, rstd) [nb, groups, 2] from the producers' partials (cs_groupnorm_finalize_parts): no pass over the tensor.
    bound: a zeroed 1-element slot that receives the tensor's magnitude bound (see range_bound)."""
    arr = _seg_array(segs)
    stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=device)
    L.check(L.load().cs_groupnorm_finalize_parts(arr, len(segs), nb, rows, c, groups, eps, stats.data_ptr(), _ptr(bound),
                                                 _stream()), "cs_groupnorm_finalize_parts")
    return stats


INVARIANT_MAX_BATCH = 16      # stats="invariant": the largest batch the tile choice is compared at (VQVAE.MAX_DECODE_BATCH)


def _epilogue_extras(lib, p, x_dev, nb: int, rps: int, m_tiles_rows: int, cout: int, want_stats,
                     out_pair: Optional[float], ncls: int = 1):
    """Ask the library what this launch's epilogue can emit (cs_conv_gemm_epilogue_caps -- the one rule) and set the
    descriptor up for it.  Returns (ColStats | None, pair_taken: bool).  rps = rows per sample the statistics tiles run
    over (the source rows for a folded Upsample conv), m_tiles_rows = the rows those tiles cover in all."""
    want = bool(want_stats)
    if not (want and _sw("GN_PARTS")) and out_pair is None:
        return None, False
    rows, pair = C.c_int32(0), C.c_int32(0)
    L.check(lib.cs_conv_gemm_epilogue_caps(C.byref(p), C.byref(rows), C.byref(pair)), "cs_conv_gemm_epilogue_caps")
    if want_stats == "invariant" and rows.value > 0:
        # r5: partial sums only where the statistics tiles do not depend on the BATCH -- the launch of ONE
        # sample must pick the same statistics-tile rows, so that an object decoded alone and inside a slice of 16 sums its
        # groups over the same tiles in the same order (the VQ decoder's bit-exact batch invariance,
        # tests/test_model_gpu.py::test_vq_decode_batch_invariance)
        # (the SAME kernel variant -- tile code and statistics rows -- at one sample, at the decode slice limit of sixteen
        # and at this batch: otherwise some batch sizes would take their statistics from the partials and others from a pass
        # over the tensor, or from differently shaped lane sums)
        t0, s0 = C.c_int32(0), C.c_int32(0)
        L.check(lib.cs_conv_gemm_launch_info(C.byref(p), C.byref(t0), C.byref(s0)), "cs_conv_gemm_launch_info")
        for nb_probe in (1, INVARIANT_MAX_BATCH):
            q = L.CsConvGemm.from_buffer_copy(p)
            q.nb = nb_probe
            r1, t1, s1 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            L.check(lib.cs_conv_gemm_epilogue_caps(C.byref(q), C.byref(r1), None), "cs_conv_gemm_epilogue_caps")
            L.check(lib.cs_conv_gemm_launch_info(C.byref(q), C.byref(t1), C.byref(s1)), "cs_conv_gemm_launch_info")
            if r1.value != rows.value or t1.value != t0.value or s1.value != s0.value or nb > INVARIANT_MAX_BATCH:
                rows = C.c_int32(0)
                break
    st = None
    if want and _sw("GN_PARTS") and rows.value > 0:
        tiles = (m_tiles_rows + rows.value - 1) // rows.value
        part = torch.empty((ncls * tiles, cout, 2), dtype=torch.float64, device=x_dev)
        p.gn_part, p.gn_ld, p.gn_rows = part.data_ptr(), cout, rows.value
        st = ColStats(part, cout, nb, rps // rows.value, ncls)
    took = False
    if out_pair is not None and pair.value:
        p.out_format, p.out_scale = 2, float(out_pair)
        took = True
    return st, took


def conv_gemm(x: Tensor, w: PackedWeight, *, spatial: Optional[Tuple[int, int, int, int]] = None,
              stride: Sequence[int] = (1, 1, 1), up: Sequence[int] = (0, 0, 0), act: int = L.ACT_NONE,
              rowvec: Optional[Tensor] = None, rv_rows: int = 1, res: Optional[Tensor] = None,
              scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
              out: Optional[Tensor] = None, tile: int = 0, math: int = L.MATH_FP32,
              splitk: Optional[int] = None, out_fn=None, a_scale: Optional[float] = None,
              stats: bool = False, out_pair: Optional[float] = None, x_bound: Optional[Tensor] = None):
    """Conv3d (k in {1,3}, pad k//2) / Linear.  x: [nb,d,h,w,c] (conv) or [..., c] rows (linear).

    `spatial=(nb,d,h,w)` lets a row matrix be interpreted as a volume without reshaping.
    r4: `stats=True` -- the result feeds a GroupNorm: where the launch can (cs_conv_gemm_epilogue_caps), its epilogue
    leaves per-(row tile, column) partial sums and the returned tensor carries them as `.cs_stats` (ColStats).
    `out_pair=s` -- the result's only reader is the next F16X3 GEMM: where the launch can, it is written as the
    interleaved operand pair of out * s and a Pair16 is returned instead of a tensor.
    `x_bound=slot` (x.cs_bound, see range_bound) -- x is a RAW activation with a known magnitude bound: the kernel derives
    the operand scale from it (no fixed guess, no overflow possible).
    """
    if w.tapcol is not None:
        if (tuple(stride) != (1, 1, 1) or tuple(up) != (0, 0, 0) or act != L.ACT_NONE or rowvec is not None
                or res is not None or scale is not None or tile or splitk):
            raise L.CsError("taps-as-columns weights: plain 3x3x3 conv only (no stride / up / act / residual / tile)")
        return _conv_tapcol(x, w, spatial, a_scale, out, out_fn)
    if isinstance(x, Wino16):
        if tuple(stride) != (1, 1, 1) or tuple(up) != (0, 0, 0) or tile or splitk or x_bound is not None:
            raise L.CsError("a Wino16 activation feeds a plain 3x3x3 stride-1 conv (no stride / up / tile / splitk)")
        return _conv_wino(x, w, act, rowvec, rv_rows, res, scale, shift, out, out_fn, stats, out_pair)
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
    p.act, p.rv_rows, p.math, p.tile = act, rv_rows, math, tile
    if x_bound is not None and math == L.MATH_F16X3 and xs is None and xp is None and _sw("DYN_SCALE"):
        p.a_bound = x_bound.data_ptr()
    lib = L.load()
    if folded:
        st, _ = _epilogue_extras(lib, p, x.device, nb, d * h * wd, m, w.cout, stats, None, ncls=len(w.classes))
        return attach_stats(_conv_gemm_up2(lib, p, w, x, out, mo), st)
    if splitk is not None and splitk > 1:            # explicit split-K factor (tuning / tests); the plan is bypassed
        ws = torch.empty((splitk * mo * w.cout,), dtype=torch.float32, device=x.device)
        p.splitk, p.splitk_ws = int(splitk), ws.data_ptr()
    elif math == L.MATH_F16X3 and tile == 0 and _sw("SPLITK") and splitk is None:
        # few output tiles (small batches): let the library cut the K loop into slices; the partial tiles live in a
        # scratch tensor that the stream-ordered allocator may reuse as soon as this call's kernels are queued
        sk, wsb = C.c_int32(1), C.c_int64(0)
        L.check(lib.cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(wsb)), "cs_conv_gemm_plan")
        if sk.value > 1:
            ws = torch.empty((wsb.value // 4,), dtype=torch.float32, device=x.device)
            p.splitk, p.splitk_ws = sk.value, ws.data_ptr()
    if p.splitk > 1 and math == L.MATH_F16X3:
        # r5: arrival counters -- where the launch is resident at once the slices finish the reduce + epilogue themselves
        p.splitk_sync, p.splitk_sync_words = sync_words().data_ptr(), SYNC_WORDS
    st, paired = (None, False)
    if math == L.MATH_F16X3:
        st, paired = _epilogue_extras(lib, p, x.device, nb, do * ho * wo, mo, w.cout, stats, out_pair)
    prof = GEMM_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(lib.cs_conv_gemm(C.byref(p), _stream()), "cs_conv_gemm")
    if prof is not None:
        e1.record()
        # which kernel variant ran: asked from the library (cs_conv_gemm_launch_info), not mirrored here
        tl_, sl_ = C.c_int32(0), C.c_int32(0)
        L.check(lib.cs_conv_gemm_launch_info(C.byref(p), C.byref(tl_), C.byref(sl_)), "cs_conv_gemm_launch_info")
        tl = tl_.value
        prof.append(dict(e0=e0, e1=e1, flops=2.0 * mo * w.cout * w.cin * kd * kh * kw, taps=kd * kh * kw,
                         m=mo, n=w.cout, k=w.cin * kd * kh * kw, tile=tl,
                         slab=sl_.value, pre=xs is not None, pair=xp is not None, res=res is not None))
    if paired:
        return Pair16(out, float(out_pair))
    return attach_stats(out, st)


def _conv_wino(xw: "Wino16", w: PackedWeight, act, rowvec, rv_rows, res, scale, shift, out, out_fn, stats, out_pair):
    """3x3x3 stride-1 conv on the Winograd-W operand (CsConvGemm.a_format = 3): the library runs the four position GEMMs in
    one launch into a workspace and the output transform + epilogue in the split-K reduce kernel's place."""
    if not w.wino or xw.variant not in w.wino or w.math != L.MATH_F16X3:
        raise L.CsError("a Wino16 activation needs a weight with the Winograd-W pack of its variant (pack_weight_wino)")
    nb, d, h, wd = xw.spatial
    mo = nb * d * h * wd
    dev = xw.hi.device
    oshape = (nb, d, h, wd, w.cout)
    if out is None:
        out = out_fn(oshape) if out_fn is not None else torch.empty(oshape, dtype=torch.float32, device=dev)
        if tuple(out.shape) != tuple(oshape):
            out = out.view(oshape)
    _chk(out, "out")
    om, oc, ldo = rows_ld(out, "out")
    if om != mo or oc != w.cout:
        raise L.CsError(f"out has shape {tuple(out.shape)}, expected {mo} rows x {w.cout}")
    lib = L.load()
    p = _wino_desc(nb, d, h, wd, w)
    wh, wl, wacc = w.wino[xw.variant]
    p.x, p.x_lo, p.a_format = xw.hi.data_ptr(), xw.lo.data_ptr(), (4 if xw.variant == 4 else 3)
    p.lda = int(xw.hi.shape[-1])
    p.w, p.w_lo, p.out, p.ldo = wh.data_ptr(), wl.data_ptr(), out.data_ptr(), ldo
    p.a_scale = float(xw.a_scale)
    p.acc_scale = wacc * (A_SCALE / float(xw.a_scale))
    p.status = status_word(dev).data_ptr()
    p.bias = _ptr(w.bias)
    p.scale, p.shift = _ptr(scale), _ptr(shift)
    p.rowvec = _ptr(rowvec)
    p.res = _ptr(res)
    p.act, p.rv_rows = act, rv_rows
    if res is not None:
        _chk(res, "res")
        rm, rc, ldr = rows_ld(res, "res")
        if rm != mo or rc != w.cout:
            raise L.CsError("res shape mismatch")
        p.ldr = ldr
    if rowvec is not None:
        _chk(rowvec, "rowvec")
        vm, vc, ldrv = rows_ld(rowvec, "rowvec")
        if vc != w.cout or vm * rv_rows < mo:
            raise L.CsError("rowvec shape mismatch")
        p.ldrv = ldrv
    sk, wsb = C.c_int32(1), C.c_int64(0)
    L.check(lib.cs_conv_wino_plan(C.byref(p), C.byref(sk), C.byref(wsb)), "cs_conv_wino_plan")
    ws = torch.empty((wsb.value // 4,), dtype=torch.float32, device=dev)
    p.splitk, p.splitk_ws = sk.value, ws.data_ptr()
    st, paired = _epilogue_extras(lib, p, dev, nb, d * h * wd, mo, w.cout, stats, out_pair)
    prof = GEMM_PROFILE
    if prof is None:
        L.check(lib.cs_conv_gemm(C.byref(p), _stream()), "cs_conv_gemm")
    else:
        # per-kernel HIP events: the position GEMMs and the output transform are the two launches of cs_conv_gemm
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        L.check(lib.cs_conv_wino_positions(C.byref(p), _stream()), "cs_conv_wino_positions")
        e1.record()
        L.check(lib.cs_conv_wino_output(C.byref(p), _stream()), "cs_conv_wino_output")
        e2.record()
        # flops = what the position GEMMs EXECUTE (18 of the direct form's 27 multiply-adds per output); flops_direct = the
        # direct form's algorithmic work the pair of launches replaces; m / k = the position launch's own GEMM shape
        npos = xw.variant + 2
        # (r6: the tail plan writes one slice for the main (position, column tile) units and `slices` for the tail's: the average)
        sl_, um_, ut_ = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(lib.cs_conv_wino_plan_info(C.byref(p), C.byref(sl_), C.byref(um_), C.byref(ut_)), "cs_conv_wino_plan_info")
        eff_slices = float(sk.value)
        if um_.value > 0 and sl_.value == sk.value:
            eff_slices = (um_.value + (ut_.value - um_.value) * sk.value) / float(ut_.value)
        prof.append(dict(e0=e0, e1=e1, e2=e2, flops=2.0 * (mo // xw.variant) * npos * w.cout * w.cin * 9,
                         flops_direct=2.0 * mo * w.cout * w.cin * 27, taps=9, m=(mo // xw.variant) * npos, n=w.cout, k=w.cin * 9,
                         npos=npos,
                         tile=4 if w.cout % 224 == 0 else 6 if w.cout % 128 == 0 else 7, slab=32, pre=True, pair=False,
                         res=res is not None, wino=True, slices=eff_slices, tail_plan=bool(um_.value > 0),
                         dispatches=2 if (um_.value > 0 and sl_.value == sk.value) else 1))
    if paired:
        return Pair16(out, float(out_pair))
    return attach_stats(out, st)


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
                         n=w.cout, k=w.cin * kd * kh * kw, tile=4 if w.cout % 224 == 0 else 6, slab=32))
    return out


# Set to a list to collect one record per GEMM launch (HIP events on the launch stream): bench.py uses this
# to measure the dominant kernel's achieved TFLOP/s inside the timed region.
GEMM_PROFILE = None


def wants_split16(m: int, w: "PackedWeight") -> bool:
    """Should the GroupNorm feeding the conv `w` over m output rows emit the fp16 hi / lo operand pair?  The rule is
    cs_conv_wants_split16 (csrc/cs_plan.hip: where the conv will run the slab kernel on a 256-row tile, and on the 128-row
    slab tile of medium batches) -- ONE implementation for this host and the native drivers (r4)."""
    if not _sw("SPLIT16_PRODUCERS"):
        return False
    k = w.k[0] if tuple(w.k) == (w.k[0],) * 3 else 0
    return bool(L.load().cs_conv_wants_split16(int(m), int(w.cin), int(w.cout), int(k),
                                               int(w.classes is None and w.tapcol is None), int(w.math)))


def _wino_desc(nb: int, d: int, h: int, wd: int, w: "PackedWeight"):
    p = L.CsConvGemm()
    p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = nb, d, h, wd, d, h, wd
    p.cin, p.cout, p.lda, p.ldo, p.ldw = w.cin_pad, w.cout, w.cin_pad, w.cout, w.ldw
    p.kd = p.kh = p.kw = 3
    p.sd = p.sh = p.sw = p.pd = p.ph = p.pw = 1
    p.math, p.rv_rows = w.math, 1
    return p


def wants_wino(nb: int, d: int, h: int, wd: int, w: "PackedWeight") -> int:
    """Should the GroupNorm feeding the 3x3x3 conv `w` over an [nb, d, h, wd] volume emit the Winograd-W operand
    (groupnorm(..., wino=variant)), and which -- 0 = no (direct form), 2 = F(2,3), 4 = F(4,3)?  cs_conv_wino_ok: the ONE
    rule (csrc/cs_gemm.hip) both hosts ask."""
    if not w.wino or not _sw("SPLIT16_PRODUCERS"):
        return 0
    v = int(L.load().cs_conv_wino_ok(C.byref(_wino_desc(int(nb), int(d), int(h), int(wd), w))))
    return v if v in w.wino else (2 if v and 2 in w.wino else 0)


def linear(x: Tensor, w: PackedWeight, **kw) -> Tensor:
    return conv_gemm(x, w, **kw)


@dataclass
class Wino16:
    """An activation volume in the Winograd-W operand form of the 3x3x3 conv that reads it (CsConvGemm.a_format = 3 / 4):
    fp16 hi / lo images [variant + 2, nb, d, h, w / variant, c] of the transformed values * a_scale
    (cs_groupnorm_apply_wino_range); variant 2 = F(2,3), 4 = F(4,3)."""
    hi: Tensor
    lo: Tensor
    a_scale: float
    spatial: Tuple[int, int, int, int]          # (nb, d, h, w) of the ORIGINAL volume
    variant: int = 2


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




# Producer-side operand split: GroupNorm writes the fp16 hi/lo pair of y * a_scale (same bytes as fp32 y) and the GEMM
# DMA-loads it with a_format=1, so its K loop carries no conversion VALU.  Bit-identical either way.
# Switches: CsDebug.no_split16 / split16_min_rows (CS_NO_SPLIT16, CS_SPLIT16_MIN_ROWS).


def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, act: int = L.ACT_NONE,
              out: Optional[Tensor] = None, split16: bool = False, a_scale: Optional[float] = None,
              bound: Optional[Tensor] = None, wino: bool = False):
    """GroupNorm over [nb, ..., c] (stats per sample & group), fused activation.
    split16=True returns a Split16 (fp16 hi/lo pair, pre-scaled) for an F16X3 GEMM to consume.
    r5: wino=True (x: [nb, d, h, w, c], ask wants_wino first) returns a Wino16 -- the Winograd-W operand of the 3x3x3 conv
    that follows, at HALF the given a_scale (the transformed values are sums / differences of two activations).
    r4: when x carries its producers' partial sums (stats_segments) the statistics come from them -- one tiny launch
    instead of a pass over the tensor -- and the tensor is read once, by the apply kernel.  bound: a zeroed 1-element slot;
    on that route the finalize kernel also leaves x's magnitude bound there and x remembers it (`x.cs_bound`, range_bound)."""
    _chk(x, "x")
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    rows = m // nb
    lib = L.load()
    segs = stats_segments(x)
    # (wino: the variant wants_wino returned -- 2 / True = F(2,3), 4 = F(4,3); operand scale / 2 resp. / 16: the transformed
    # values are bounded by 2 x resp. 10 x the activation's bound)
    wino = (4 if wino == 4 else 2 if wino else 0) if _sw("SPLIT16_PRODUCERS") else 0
    if wino:
        if x.dim() != 5 or int(x.shape[3]) % wino or c % 8:
            raise L.CsError("groupnorm(wino=v) needs x as [nb, d, h, w, c] with w % v == 0 and c % 8 == 0")
        split16 = True          # (same statistics routes as the pre-split pair; only the apply kernel differs)

    def _emit(stats):
        yshape = (wino + 2, nb, int(x.shape[1]), int(x.shape[2]), int(x.shape[3]) // wino, c) if wino else x.shape
        yh = torch.empty(yshape, dtype=torch.float16, device=x.device)
        yl = torch.empty(yshape, dtype=torch.float16, device=x.device)
        if wino:
            a_sc = float(a_scale or A_SCALE) * (0.5 if wino == 2 else 0.0625)
            L.check(lib.cs_groupnorm_apply_wino_range(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                      yh.data_ptr(), yl.data_ptr(), nb, int(x.shape[1]), int(x.shape[2]),
                                                      int(x.shape[3]), c, ldx, c, groups, c // groups, 0, act, a_sc, wino,
                                                      status_word(x.device).data_ptr(), _stream()),
                    "cs_groupnorm_apply_wino_range")
            return Wino16(yh, yl, a_sc, (nb, int(x.shape[1]), int(x.shape[2]), int(x.shape[3])), wino)
        a_sc = float(a_scale or A_SCALE)
        L.check(lib.cs_groupnorm_apply_split16(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               yh.data_ptr(), yl.data_ptr(), nb, rows, c, ldx, c, groups, act,
                                               a_sc, status_word(x.device).data_ptr(), _stream()),
                "cs_groupnorm_apply_split16")
        return Split16(yh, yl, a_sc)

    if segs is not None and not (split16 and _sw("SPLIT16_PRODUCERS")):
        # one call: a single launch for small tensors, finalize + apply otherwise (cs_groupnorm_parts decides)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        om, oc, ldy = rows_ld(out, "out")
        if om != m or oc != c:
            raise L.CsError("groupnorm out shape mismatch")
        stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
        if not _sw("DYN_SCALE"):
            bound = None
        L.check(lib.cs_groupnorm_parts(x.data_ptr(), _seg_array(segs), len(segs), gamma.data_ptr(), beta.data_ptr(),
                                       out.data_ptr(), nb, rows, c, ldx, ldy, groups, eps, act, stats.data_ptr(),
                                       _ptr(bound), _stream()), "cs_groupnorm_parts")
        if bound is not None:
            x.cs_bound = bound
        return out
    if segs is not None:
        if not _sw("DYN_SCALE"):
            bound = None
        stats = groupnorm_stats_from_parts(segs, nb, rows, c, groups, eps, x.device, bound)
        if bound is not None:
            x.cs_bound = bound
        if split16 and _sw("SPLIT16_PRODUCERS"):
            return _emit(stats)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        om, oc, ldy = rows_ld(out, "out")
        if om != m or oc != c:
            raise L.CsError("groupnorm out shape mismatch")
        L.check(lib.cs_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                       nb, rows, c, ldx, ldy, groups, act, _stream()), "cs_groupnorm_apply")
        return out
    ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device=x.device)
    stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
    want_bound = bound is not None and _sw("DYN_SCALE")

    def _stats():
        if want_bound:      # no partials on x: the statistics pass leaves the magnitude bound beside (mean, rstd)
            L.check(lib.cs_groupnorm_stats_bound(x.data_ptr(), nb, rows, c, ldx, groups, eps, ws.data_ptr(), stats.data_ptr(),
                                                 bound.data_ptr(), _stream()), "cs_groupnorm_stats_bound")
            x.cs_bound = bound
        else:
            L.check(lib.cs_groupnorm_stats(x.data_ptr(), nb, rows, c, ldx, groups, eps, ws.data_ptr(),
                                           stats.data_ptr(), _stream()), "cs_groupnorm_stats")
    if split16 and _sw("SPLIT16_PRODUCERS"):
        _stats()
        return _emit(stats)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    om, oc, ldy = rows_ld(out, "out")
    if om != m or oc != c:
        raise L.CsError("groupnorm out shape mismatch")
    if want_bound:
        _stats()
        L.check(lib.cs_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                       nb, rows, c, ldx, ldy, groups, act, _stream()), "cs_groupnorm_apply")
        return out
    # one launch for small tensors (one or two objects), statistics + apply otherwise: cs_groupnorm decides
    L.check(lib.cs_groupnorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), nb, rows, c, ldx, ldy,
                             groups, eps, act, ws.data_ptr(), stats.data_ptr(), _stream()), "cs_groupnorm")
    return out


def groupnorm_stats(x: Tensor, groups: int, eps: float, bound: Optional[Tensor] = None) -> Tensor:
    """(mean, rstd) per (sample, group) of a channels-last tensor: [nb, groups, 2] fp32 (fp64 accumulation)."""
    _chk(x, "x")
    nb = x.shape[0]
    m, c, ldx = rows_ld(x, "x")
    lib = L.load()
    segs = stats_segments(x)
    if segs is not None:
        if bound is not None and _sw("DYN_SCALE"):
            x.cs_bound = bound
        else:
            bound = None
        return groupnorm_stats_from_parts(segs, nb, m // nb, c, groups, eps, x.device, bound)
    ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device=x.device)
    stats = torch.empty((nb, groups, 2), dtype=torch.float32, device=x.device)
    if bound is not None and _sw("DYN_SCALE"):
        L.check(lib.cs_groupnorm_stats_bound(x.data_ptr(), nb, m // nb, c, ldx, groups, eps, ws.data_ptr(), stats.data_ptr(),
                                             bound.data_ptr(), _stream()), "cs_groupnorm_stats_bound")
        x.cs_bound = bound
        return stats
    L.check(lib.cs_groupnorm_stats(x.data_ptr(), nb, m // nb, c, ldx, groups, eps, ws.data_ptr(), stats.data_ptr(),
                                   _stream()), "cs_groupnorm_stats")
    return stats


def groupnorm_apply_range(x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, cpg: int, ch0: int,
                          act: int = L.ACT_NONE, split16: bool = False, a_scale: Optional[float] = None, wino: bool = False):
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
    if wino and _sw("SPLIT16_PRODUCERS"):       # r5: the Winograd-W operand of the conv that reads it (x: [nb, d, h, w, c])
        v = 4 if wino == 4 else 2
        if x.dim() != 5 or int(x.shape[3]) % v or c % 8:
            raise L.CsError("groupnorm_apply_range(wino=v) needs x as [nb, d, h, w, c] with w % v == 0 and c % 8 == 0")
        d_, h_, w_ = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        yh = torch.empty((v + 2, nb, d_, h_, w_ // v, c), dtype=torch.float16, device=x.device)
        yl = torch.empty_like(yh)
        a_sc = float(a_scale or A_SCALE) * (0.5 if v == 2 else 0.0625)
        L.check(lib.cs_groupnorm_apply_wino_range(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                  yh.data_ptr(), yl.data_ptr(), nb, d_, h_, w_, c, ldx, c, groups, cpg,
                                                  ch0, act, a_sc, v, status_word(x.device).data_ptr(), _stream()),
                "cs_groupnorm_apply_wino_range")
        return Wino16(yh, yl, a_sc, (nb, d_, h_, w_), v)
    if split16 and _sw("SPLIT16_PRODUCERS"):
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
    if pair_scale is not None and _sw("PAIR16_PRODUCERS") and c % 16 == 0:
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


def bound_a_scale(bound: float) -> float:
    """cs_bound_a_scale: the largest power of two s with bound * s <= 65000 -- the F16X3 operand scale of a tensor whose
    magnitude is bounded by `bound` (ONE rule for every host)."""
    return float(L.load().cs_bound_a_scale(float(bound)))


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, out: Optional[Tensor] = None,
              math: int = L.MATH_FP32, scales: Optional[Tuple[float, float, float]] = None) -> Tensor:
    """q: [nb, nq, heads*dh] (views with wider row stride allowed), k/v: [nb, nk, heads*dh].
    scales (F16X3 only, r5): the power-of-two operand pre-scales (q * scale, k, v) from the static bounds of a transformer
    block's q / k / v instead of the constant 16 (cs_attn_selfattn_f16x3_scaled)."""
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
    if math == L.MATH_F16X3 and scales is not None:
        # (r6: the K / V tile-image path too, where the library has it for this shape: cs_attn_selfattn_f16x3_ws_scaled)
        wsb = lib.cs_attn_f16x3_ws_bytes(nb, nq, nk, heads, dh)
        if wsb > 0:
            ws = torch.empty((wsb // 4,), dtype=torch.float32, device=q.device)
            L.check(lib.cs_attn_selfattn_f16x3_ws_scaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nb, nq, nk,
                                                         heads, dh, ldq, ldk, ldv, ldo, scale, float(scales[0]),
                                                         float(scales[1]), float(scales[2]),
                                                         status_word(q.device).data_ptr(), ws.data_ptr(), _stream()),
                    "cs_attn_selfattn_f16x3_ws_scaled")
        else:
            L.check(lib.cs_attn_selfattn_f16x3_scaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nb, nq, nk,
                                                      heads, dh, ldq, ldk, ldv, ldo, scale, float(scales[0]), float(scales[1]),
                                                      float(scales[2]), status_word(q.device).data_ptr(), _stream()),
                    "cs_attn_selfattn_f16x3_scaled")
    elif math == L.MATH_F16X3:
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


def weight_rowstats(mats: Sequence[Tensor]) -> Tuple[float, float]:
    """(max row 2-norm rounded up, max |entry|) over the rows of every [rows, cols] matrix in `mats` -- cs_weight_rowstats into
    ONE zeroed slot pair (the kernel folds by atomicMax), one read-back: load time only.  The native drivers run the same
    kernel on the same tensors, so both hosts derive the same scales."""
    if not mats:
        return 0.0, 0.0
    lib = L.load()
    slot = torch.zeros((2,), dtype=torch.float32, device=mats[0].device)
    for m in mats:
        m = m.contiguous()
        L.check(lib.cs_weight_rowstats(m.data_ptr(), int(m.shape[0]), int(m.shape[1]), slot.data_ptr(), _stream()),
                "cs_weight_rowstats")
    v = slot.cpu().tolist()
    return float(v[0]), float(v[1])


def attnblock_static_scales(gmax: float, bmax: float, n: int, c: int, w_l2max: float, b_absmax: float,
                            qk_scale: float) -> Optional[Tuple[float, float, float, float]]:
    """F16X3 operand scales (q * qk_scale, k, v, attention output) of an attention block whose q / k / v are
    Conv1x1(GroupNorm(x)) + bias, from bounds that hold for every input (cs_attnblock_static_scales: ONE rule, both hosts;
    r6).  None when the feature is off (CS_NO_STATIC_SCALES) or there are no statistics."""
    if not _sw("STATIC_SCALES") or not w_l2max > 0.0:
        return None
    o = (C.c_float * 4)()
    L.check(L.load().cs_attnblock_static_scales(float(gmax), float(bmax), int(n), int(c), float(w_l2max), float(b_absmax),
                                                float(qk_scale), o), "cs_attnblock_static_scales")
    return float(o[0]), float(o[1]), float(o[2]), float(o[3])


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


def absmax_bound(x: Tensor, slot: Optional[Tensor]) -> Optional[Tensor]:
    """Leave the exact max |x| in `slot` (a zeroed 1-element fp32 device tensor) and remember it on x (`x.cs_bound`): the
    magnitude bound of a RAW tensor that nothing normalises -- the UNet's conv_in operand x_t (cs_absmax; r6).  A consumer
    hands it to conv_gemm(x_bound=).  None when the feature is off."""
    if slot is None or not _sw("DYN_SCALE"):
        return None
    _chk(x, "x")
    if not x.is_contiguous():
        raise L.CsError("absmax_bound: contiguous tensor expected")
    L.check(L.load().cs_absmax(x.data_ptr(), x.numel(), slot.data_ptr(), _stream()), "cs_absmax")
    x.cs_bound = slot
    return slot


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

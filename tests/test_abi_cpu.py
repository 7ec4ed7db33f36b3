"""CPU suite: the C-ABI library builds, loads and exports every symbol include/commonscenes_hip.h declares
(no compute calls without a GPU), and the host-side plumbing behaves."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _header_symbols():
    txt = (ROOT / "include" / "commonscenes_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from commonscenes_amd import build, lib
    path = build.build_native(verbose=False)
    assert path.exists()
    dll = lib.load()
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in the header but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(lib.SIGNATURES) == set(syms)
    assert dll.cs_abi_version() == lib.ABI_VERSION == 18


def test_struct_layout_matches_header():
    """field order/count of the ctypes mirror vs the C struct text."""
    from commonscenes_amd import lib
    txt = (ROOT / "include" / "commonscenes_hip.h").read_text()
    body = txt[txt.index("typedef struct CsConvGemm {"):txt.index("} CsConvGemm;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.splitlines()[1:]:
        line = line.strip().rstrip(";")
        if not line:
            continue
        decl = line.split(None, 2) if line.startswith("const") else line.split(None, 1)
        for n in decl[-1].split(","):
            names.append(n.strip().lstrip("*").strip().split("[")[0])      # (array fields: `a_amax[2]` -> a_amax)
    assert names == [f[0] for f in lib.CsConvGemm._fields_]
    assert ctypes.sizeof(lib.CsConvGemm) == 15 * 8 + 35 * 4 + 3 * 4 + 2 * 4


def test_missing_library_fails_loudly(tmp_path):
    from commonscenes_amd import lib
    with pytest.raises(lib.NativeLibraryMissing):
        lib.load(tmp_path / "nope.so")


def test_ops_refuse_cpu_tensors():
    from commonscenes_amd import lib, ops
    with pytest.raises(lib.CsError):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))
    with pytest.raises(lib.CsError):
        ops.geglu(torch.zeros(4, 8))


def test_rows_ld_views():
    from commonscenes_amd import lib, ops
    t = torch.zeros(2, 3, 4, 5, 24)
    assert ops.rows_ld(t) == (120, 24, 24)
    assert ops.rows_ld(t[..., 4:12]) == (120, 8, 24)
    assert ops.rows_ld(torch.zeros(7, 1, 16)) == (7, 16, 16)
    assert ops.rows_ld(torch.zeros(16)) == (1, 16, 16)
    with pytest.raises(lib.CsError):
        ops.rows_ld(t.permute(0, 4, 1, 2, 3))
    with pytest.raises(lib.CsError):
        ops.rows_ld(t[:, ::2])


def test_unet_param_table_matches_reference_counts():
    """413,540,739 parameters in 496 tensors."""
    import numpy as np
    from commonscenes_amd.unet import unet_param_shapes, unet_blocks, _cfg
    from oracle.ref_torch import UNET_FULL
    cfg = dict(UNET_FULL, dims=3, use_spatial_transformer=True)
    s = unet_param_shapes(cfg)
    assert len(s) == 496
    assert sum(int(np.prod(v)) for v in s.values()) == 413_540_739
    inp, mid, out, ch = unet_blocks(_cfg(cfg))
    assert len(inp) == 9 and len(out) == 9 and ch == 224
    assert [l["cin"] for b in out for l in b if l["kind"] == "res"] == [1344, 1344, 1120, 1120, 896, 672, 672, 448, 448]


def test_unsupported_configs_raise():
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from oracle.ref_torch import UNET_CONCAT_FULL, UNET_SMALL
    for bad in (dict(dims=2), dict(use_scale_shift_norm=True), dict(num_head_channels=32), dict(num_classes=10)):
        with pytest.raises(NotImplementedError):
            DiffusionUNet({**dict(UNET_SMALL, dims=3, use_spatial_transformer=True), **bad},
                          conditioning_key="crossattn", device="cpu")
    with pytest.raises(ValueError):          # a transformer UNet needs a context width (openai_model_3d.py:512-513)
        DiffusionUNet(dict(UNET_SMALL, dims=3, use_spatial_transformer=True, context_dim=None), device="cpu")
    # concat conditioning cannot feed SpatialTransformer3D blocks (no context), and vice versa
    df = DiffusionUNet(dict(UNET_SMALL, dims=3, use_spatial_transformer=True), conditioning_key="concat",
                       device="cpu")
    with pytest.raises(NotImplementedError):
        df(torch.zeros(1, 3, 16, 16, 16), torch.zeros(1, dtype=torch.long), c_concat=[torch.zeros(1, 1, 16, 16, 16)])
    df = DiffusionUNet(dict(UNET_CONCAT_FULL), conditioning_key="crossattn", device="cpu")
    with pytest.raises(NotImplementedError):
        df(torch.zeros(1, 3, 16, 16, 16), torch.zeros(1, dtype=torch.long), c_crossattn=[torch.zeros(1, 1, 1280)])
    # the concat family's parameter count (config/sdfusion-txt2shape_concat.yaml)
    n = 0
    for shp in unet_param_shapes(dict(UNET_CONCAT_FULL)).values():
        k = 1
        for v in shp:
            k *= v
        n += k
    assert n == 337_988_003


def test_unet_plan_is_host_only_and_lists_the_reference_state_dict():
    """cs_unet_create / param_info / *_bytes need no GPU: the plan's parameter table must be the reference
    UNet3DModel state_dict (names, shapes, order -- SURVEY App. C) and the sizes must be sane."""
    from commonscenes_amd import lib as L
    from commonscenes_amd.unet import unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    from oracle.ref_torch import UNET_FULL
    cfg = dict(UNET_FULL, dims=3, use_spatial_transformer=True)
    n = NativeDiffusionUNet(cfg, device="cpu")
    ref = unet_param_shapes(cfg)
    assert list(n.shapes.items()) == list(ref.items())
    assert n.num_parameters() == 413_540_739
    lib = L.load()
    assert lib.cs_unet_raw_bytes(n._h) >= 4 * 413_540_739
    assert lib.cs_unet_arena_bytes(n._h) > 0
    w1, w32 = lib.cs_unet_workspace_bytes(n._h, 1, 1), lib.cs_unet_workspace_bytes(n._h, 32, 1)
    assert 0 < w1 < w32 < 8 * 2 ** 30
    assert n.ctx_floats == 5 * 448 + 6 * 672
    import ctypes
    bad = L.CsUnetConfig()
    h = ctypes.c_void_p()
    assert lib.cs_unet_create(ctypes.byref(bad), ctypes.byref(h)) == L.CS_EINVAL


def test_graft_entry_build_passes():
    """the driver's "does it build" check: compiles (if stale), loads, resolves every symbol, checks the ABI."""
    import __graft_entry__ as g
    g.build()


def test_every_entry_rejects_null_arguments_without_touching_the_device():
    """Argument checks come first in every C entry: all-zero arguments (NULL pointers, zero sizes) must return
    CS_EINVAL -- no launch, no HIP call, so this runs without a GPU."""
    import ctypes as C
    from commonscenes_amd import lib
    dll = lib.load()
    skip = {"cs_abi_version", "cs_groupnorm_ws_bytes", "cs_attn_f16x3_ws_bytes", "cs_conv_gemm_up2_ws_bytes", "cs_mc_blocks_per_object", "cs_gcn_csr_ints", "cs_unet_destroy", "cs_unet_param_count", "cs_unet_raw_bytes",
            "cs_unet_arena_bytes", "cs_unet_context_floats", "cs_vqvae_destroy", "cs_vqvae_param_count",
            "cs_vqvae_raw_bytes", "cs_vqvae_arena_bytes",
            # r4: host-side rule / switch queries (csrc/cs_plan.hip): plain functions of their arguments, no status code
            "cs_debug", "cs_debug_set", "cs_norm_a_scale", "cs_bound_a_scale", "cs_conv_wants_split16", "cs_tapcol_ok", "cs_tapcol_tile",
            "cs_conv_wino_ok"}
    checked = 0
    for name, (res, args) in lib.SIGNATURES.items():
        if name in skip:
            continue
        vals = []
        for a in args:
            if a in (C.c_float, C.c_double):
                vals.append(0.0)
            elif a in (C.c_int, C.c_int32, C.c_int64, C.c_uint64):
                vals.append(0)
            else:
                vals.append(None)
        rc = getattr(dll, name)(*vals)
        assert rc == lib.CS_EINVAL, f"{name}(all zero) -> {rc}"
        checked += 1
    assert checked >= 35
    # the queries that return sizes answer 0 for a NULL plan, and destroy(NULL) is a no-op
    assert dll.cs_unet_param_count(None) == 0 and dll.cs_unet_raw_bytes(None) == 0
    dll.cs_unet_destroy(None)


def test_conv_gemm_descriptor_validation():
    """cs_conv_gemm / cs_conv_gemm_plan reject malformed descriptors before any launch."""
    import ctypes as C
    from commonscenes_amd import lib
    dll = lib.load()
    buf = (C.c_float * 64)()
    base = C.addressof(buf)
    base += (-base) % 16

    def desc(**kw):
        p = lib.CsConvGemm()
        p.x = p.w = p.out = base
        p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = 1, 1, 1, 1, 1, 1, 1
        p.cin, p.cout, p.lda, p.ldw, p.ldo = 4, 4, 4, 4, 4
        p.kd = p.kh = p.kw = p.sd = p.sh = p.sw = 1
        p.rv_rows = 1
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    for bad in (dict(cin=3), dict(lda=2), dict(ldo=2), dict(kd=0), dict(sd=0), dict(ud=5), dict(math=7),
                dict(x=base + 4), dict(scale=base), dict(rowvec=base, rv_rows=0), dict(act=lib.ACT_GEGLU),
                dict(math=lib.MATH_F16X3), dict(splitk=2), dict(tile=11, math=lib.MATH_F16X3, w_lo=base, acc_scale=1.0)):
        assert dll.cs_conv_gemm(C.byref(desc(**bad)), None) == lib.CS_EINVAL, bad
    sk, ws = C.c_int32(-1), C.c_int64(-1)
    assert dll.cs_conv_gemm_plan(C.byref(desc()), C.byref(sk), C.byref(ws)) == 0 and sk.value == 1 and ws.value == 0
    big = desc(nb=2, dout=16, hout=4, wout=4, cin=672, cout=672, kd=3, kh=3, kw=3, math=lib.MATH_F16X3)
    assert dll.cs_conv_gemm_plan(C.byref(big), C.byref(sk), C.byref(ws)) == 0
    assert sk.value == 32 and ws.value == 32 * 512 * 672 * 4
    short = desc(nb=512, dout=1, hout=1, wout=1, cin=672, cout=672, kd=1, kh=1, kw=1, math=lib.MATH_F16X3)
    assert dll.cs_conv_gemm_plan(C.byref(short), C.byref(sk), C.byref(ws)) == 0 and sk.value == 1   # 42 chunks: no split


def test_vqvae_plan_is_host_only_and_lists_the_decode_side_state_dict():
    from commonscenes_amd import lib as L
    from commonscenes_amd.vqvae import vqvae_param_shapes
    from commonscenes_amd.vqvae_native import NativeVQVAE
    from oracle.ref_torch import VQ_FULL
    n = NativeVQVAE(VQ_FULL, 8192, 3, device="cpu")
    assert list(n.shapes.items()) == list(vqvae_param_shapes(VQ_FULL, 8192, 3).items())
    lib = L.load()
    assert lib.cs_vqvae_arena_bytes(n._h) > 0
    w1, w8 = lib.cs_vqvae_workspace_bytes(n._h, 1), lib.cs_vqvae_workspace_bytes(n._h, 8)
    assert 0 < w1 < w8 < 64 * 2 ** 30


def test_epilogue_caps_is_one_host_side_rule():
    """cs_conv_gemm_epilogue_caps (r4, ABI 14): what a launch's epilogue can emit -- GroupNorm partial sums per tile of
    gn_rows rows, the interleaved operand pair -- decided on the host for all tiles of the launch; both hosts (ops.py,
    cs_driver.h) ask it, cs_conv_gemm re-checks.  No device work: runs without a GPU."""
    import ctypes as C
    from commonscenes_amd import lib
    dll = lib.load()
    buf = (C.c_float * 256)()
    base = C.addressof(buf)
    base += (-base) % 64

    def desc(**kw):
        p = lib.CsConvGemm()
        p.x = p.w = p.out = p.bias = base
        p.kd = p.kh = p.kw = p.sd = p.sh = p.sw = 1
        p.rv_rows = 1
        p.math = lib.MATH_F16X3
        for k, v in kw.items():
            setattr(p, k, v)
        p.lda = p.lda or p.cin
        p.ldo = p.ldo or p.cout
        p.ldw = p.cout
        return p

    def caps(p):
        rows, pair = C.c_int32(-1), C.c_int32(-1)
        assert dll.cs_conv_gemm_epilogue_caps(C.byref(p), C.byref(rows), C.byref(pair)) == 0
        return rows.value, pair.value

    conv = dict(kd=3, kh=3, kw=3, pd=1, ph=1, pw=1)
    # 32 objects' level-0 conv: 256x224 tiles, 16 of them per sample
    assert caps(desc(nb=64, din=16, hin=16, win=16, dout=16, hout=16, wout=16, cin=224, cout=224, **conv)) == (256, 1)
    # one object (CFG batch 2), 4^3 level: the plan cuts K, the reduce kernel emits per 16 rows
    p = desc(nb=2, din=16, hin=4, win=4, dout=16, hout=4, wout=4, cin=672, cout=672, **conv)
    sk, ws = C.c_int32(0), C.c_int64(0)
    assert dll.cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(ws)) == 0 and sk.value > 1
    p.splitk = sk.value
    assert caps(p) == (16, 1)
    # no bias / residual / row vector: the kernel takes the unfused epilogue -> nothing
    q = desc(nb=64, din=1024, hin=1, win=1, dout=1024, hout=1, wout=1, cin=448, cout=1344)
    q.bias = None
    assert caps(q) == (0, 0)
    # token GEMM seen as nb samples of n tokens: tiles of 256 rows divide the 1024-token samples, not 100-token ones
    # (r6: one-tap GEMMs up to 128 K chunks run two 128-row workgroups per CU -- rule (iv) of auto_tile; longer K loops keep the
    # 256-row tile)
    assert caps(desc(nb=64, din=1024, hin=1, win=1, dout=1024, hout=1, wout=1, cin=448, cout=448)) == (128, 1)
    assert caps(desc(nb=64, din=256, hin=1, win=1, dout=256, hout=1, wout=1, cin=2688, cout=672)) == (256, 1)
    assert caps(desc(nb=64, din=100, hin=1, win=1, dout=100, hout=1, wout=1, cin=448, cout=448)) == (0, 1)
    # rows described as M one-row samples (how a plain Linear is described): no sample structure to tile
    assert caps(desc(nb=65536, din=1, hin=1, win=1, dout=1, hout=1, wout=1, cin=448, cout=448))[0] == 0
    # a column count the tile does not divide (236 columns run 128x128 tiles): the edge tile's lanes past cout are masked;
    # 236 % 8 != 0 -> no pair layout
    assert caps(desc(nb=64, din=16, hin=16, win=16, dout=16, hout=16, wout=16, cin=224, cout=236, **conv)) == (128, 0)
    # fp32-input MFMA mode: never
    assert caps(desc(nb=64, din=16, hin=16, win=16, dout=16, hout=16, wout=16, cin=224, cout=224, math=lib.MATH_FP32, **conv)) == (0, 0)
    # cs_conv_gemm refuses what the rule does not allow, before any launch
    bad = desc(nb=64, din=100, hin=1, win=1, dout=100, hout=1, wout=1, cin=448, cout=448)
    bad.w_lo, bad.acc_scale, bad.gn_part, bad.gn_ld, bad.gn_rows = base, 1.0, base, 448, 256
    assert dll.cs_conv_gemm(C.byref(bad), None) == lib.CS_EINVAL


def test_shared_host_rules_and_debug_struct():
    """r4 "one plan, one place": the rules both hosts used to carry a copy of live in the library (csrc/cs_plan.hip) and the
    CS_* switches are ONE struct.  Host-only entries: checked without a GPU."""
    from commonscenes_amd import lib as L, ops
    dll = L.load()
    # cs_norm_a_scale: the values the GPU suite pins for the shipped layers
    assert ops.norm_a_scale(1.2, 0.1, 28672) == 256.0 and ops.norm_a_scale(1.2, 0.1, 448) == 2048.0
    assert ops.norm_a_scale(0.0, 0.0, 100) == 2.0 ** 40 and ops.norm_a_scale(1e9, 0.0, 100) == 2.0 ** -8
    # cs_conv_wants_split16: large batches (slab kernel on a 256-row tile) yes, 1 object no, medium batches from 8192 rows
    assert dll.cs_conv_wants_split16(64 * 4096, 224, 224, 3, 1, L.MATH_F16X3) == 1
    assert dll.cs_conv_wants_split16(2 * 4096, 224, 224, 3, 1, L.MATH_F16X3) == 1          # >= 8192 rows
    assert dll.cs_conv_wants_split16(2 * 1024, 448, 448, 3, 1, L.MATH_F16X3) == 0
    assert dll.cs_conv_wants_split16(64 * 4096, 224, 224, 1, 1, L.MATH_F16X3) == 0          # not a 3x3x3 conv
    assert dll.cs_conv_wants_split16(64 * 4096, 224, 224, 3, 0, L.MATH_F16X3) == 0          # folded / taps-as-columns
    assert dll.cs_conv_wants_split16(64 * 4096, 224, 224, 3, 1, L.MATH_FP32) == 0
    assert dll.cs_tapcol_ok(3, 224, 3, L.MATH_F16X3) == 1 and dll.cs_tapcol_ok(8, 224, 3, L.MATH_F16X3) == 0
    assert dll.cs_tapcol_tile(100, 84) == 0 and dll.cs_tapcol_tile(64 * 4096, 84) == 6 and dll.cs_tapcol_tile(64 * 4096, 28) == 7
    # the debug struct: defaults, override, restore; ops' module attributes are views of it
    d = L.debug()
    assert d.split16_min_rows == 8192 and d.gn_small_group == 11264 and d.no_gn_parts == 0
    assert ops.GN_PARTS is True and ops.SPLITK is True
    with L.debug_override(no_gn_parts=1, split16_min_rows=0):
        assert L.debug().no_gn_parts == 1 and ops.GN_PARTS is False
        assert dll.cs_conv_wants_split16(2 * 4096, 224, 224, 3, 1, L.MATH_F16X3) == 0
    assert L.debug().no_gn_parts == 0 and ops.GN_PARTS is True
    # cs_conv_gemm_launch_info: the dominant kernel's variant, asked not mirrored
    import ctypes as C
    p = L.CsConvGemm()
    p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = 64, 16, 16, 16, 16, 16, 16
    p.cin, p.cout, p.lda, p.ldo, p.ldw = 672, 224, 672, 224, 224
    p.kd = p.kh = p.kw = 3
    p.sd = p.sh = p.sw = p.pd = p.ph = p.pw = 1
    p.math, p.a_format, p.rv_rows = L.MATH_F16X3, 1, 1
    t, sl = C.c_int32(0), C.c_int32(0)
    assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and (t.value, sl.value) == (4, 32)
    p.kd = p.kh = p.kw = 1
    p.pd = p.ph = p.pw = 0
    p.a_format = 0
    # (r6 rule (iv): a one-tap GEMM of 42 K chunks -- 672 -> 224 at 262144 rows -- runs on the 128-row pair; 2688 channels, 168
    # chunks, keep the 256-row tile)
    assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and (t.value, sl.value) == (2, 0)
    p.cin, p.lda = 2688, 2688
    assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and (t.value, sl.value) == (4, 0)
    # r5 rules: a single partial round of 256-row tiles with a short K loop takes two 128-row workgroups per CU (the level-0
    # skip conv at the reference's mini-batch of 7: 224 tiles); the fused gate always runs on the 128-row tile
    p.nb, p.cin, p.lda = 14, 448, 448
    assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and (t.value, sl.value) == (2, 0)
    with L.debug_override(no_tok_rules=1):
        assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and t.value == 4
    p.nb, p.hin, p.win, p.hout, p.wout = 64, 4, 4, 4, 4
    p.cin, p.lda, p.cout, p.ldo, p.ldw, p.act, p.a_format = 672, 672, 5376, 2688, 5376, L.ACT_GEGLU, 2
    assert dll.cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(sl)) == 0 and (t.value, sl.value) == (2, 0)


def test_static_bound_scales_are_powers_of_two_that_keep_the_bound_in_range():
    """r5 host rules (csrc/cs_plan.hip): cs_bound_a_scale = the largest power of two s with bound * s <= 65000;
    cs_transformer_static_scales = the scales of the operands born inside a transformer block from the weights' statistics
    (attention.py:237-245): every scale is a power of two, scale * bound stays inside the fp16 range, a larger weight
    statistic never gives a larger scale, and the formula matches its restatement here."""
    import ctypes as C
    import math
    from commonscenes_amd import lib
    dll = lib.load()
    for b in (1e-30, 1e-6, 0.3, 1.0, 15.9, 16.0, 4094.0, 65000.0, 65001.0, 1.2e6, 3e10):
        s = dll.cs_bound_a_scale(b)
        assert s > 0 and math.log2(s) == int(math.log2(s))
        if 2.0 ** -24 < s < 2.0 ** 40:
            assert b * s <= 65000.0 < b * s * 2
    assert dll.cs_bound_a_scale(0.0) == 2.0 ** 40 and dll.cs_bound_a_scale(float("nan")) == 2.0 ** 40
    st = lib.CsTransformerStats()
    vals = dict(rq=0.6, rk=0.61, rv=0.59, ro=0.58, bo=0.05, rx=0.6, bx=0.05, rg=0.6, bg=0.05, r2=0.55, b2=0.02, rpi=0.58,
                bpi=0.04, g1=1.2, be1=2.1, g3=1.2, be3=2.0)
    for k, v in vals.items():
        setattr(st, k, v)
    out = (C.c_float * 12)()
    c, ntok, heads = 448, 1024, 8
    assert dll.cs_transformer_static_scales(C.byref(st), c, ntok, heads, 1.2, 0.1, 3.0, out) == 0
    rc = math.sqrt(c)
    y1 = vals["g1"] * rc + vals["be1"]
    bq, bk, bv = vals["rq"] * y1, vals["rk"] * y1, vals["rv"] * y1
    egn = 1.2 * math.sqrt(ntok * (c // 32) - 1) + 0.1
    bt1 = vals["ro"] * rc * bv + vals["bo"] + (vals["rpi"] * rc * egn + vals["bpi"]) + 3.0
    y3 = vals["g3"] * rc + vals["be3"]
    bgg = (vals["rx"] * y3 + vals["bx"]) * (vals["rg"] * y3 + vals["bg"])
    bt2 = vals["r2"] * math.sqrt(4 * c) * bgg + vals["b2"] + bt1
    for got, want in zip(list(out)[6:], (bq, bk, bv, bt1, bgg, bt2)):
        assert abs(got - want) <= 1e-5 * want
    for s, b in zip(list(out)[:6], (bq / math.sqrt(c // heads), bk, bv, bv, bgg, bt2)):
        assert math.log2(s) == int(math.log2(s)) and b * s <= 65000.0 * (1 + 1e-5) < 2 * b * s * (1 + 1e-4)
    # monotone: scaling to_v up by 1e5 scales v's (and t1's, t2's) bound and can only shrink those scales
    base = list(out)
    st.rv = vals["rv"] * 1e5
    assert dll.cs_transformer_static_scales(C.byref(st), c, ntok, heads, 1.2, 0.1, 3.0, out) == 0
    assert out[2] < base[2] and out[3] < base[3] and out[5] < base[5] and out[0] == base[0] and out[4] == base[4]
    assert out[8] * out[2] <= 65000.0 * (1 + 1e-5)
    assert dll.cs_transformer_static_scales(None, c, ntok, heads, 1.0, 0.0, 0.0, out) == lib.CS_EINVAL
    assert dll.cs_transformer_static_scales(C.byref(st), 450, ntok, 8, 1.0, 0.0, 0.0, out) == lib.CS_EINVAL


def test_winograd_route_rule_and_slice_plan_are_host_functions_of_the_geometry():
    """r5 (csrc/cs_gemm.hip: cs_conv_wino_ok / cs_conv_wino_plan; openai_model_3d.py:294-314 convs): which 3x3x3 convs take the
    Winograd-W route, in which variant, and how many K slices their position GEMMs get -- plain functions of the descriptor,
    asked by both hosts BEFORE the GroupNorm emits the operand.  No GPU needed."""
    import ctypes as C
    from commonscenes_amd import lib as L
    dll = L.load()

    def desc(nb, d, h, w, cin, cout, a_format=0):
        p = L.CsConvGemm()
        p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = nb, d, h, w, d, h, w
        p.cin, p.cout, p.lda, p.ldo, p.ldw = cin, cout, cin, cout, cout
        p.kd = p.kh = p.kw = 3
        p.sd = p.sh = p.sw = p.pd = p.ph = p.pw = 1
        p.math, p.rv_rows, p.a_format = L.MATH_F16X3, 1, a_format
        return p

    ok = lambda *a: dll.cs_conv_wino_ok(C.byref(desc(*a)))
    # the UNet's widths: F(4,3) from 2048 rows where W % 4 == 0 and M / 4 is whole 256-row tiles, F(2,3) from 1024, else direct
    assert ok(64, 16, 16, 16, 224, 224) == 4 and ok(64, 16, 4, 4, 672, 672) == 4 and ok(2, 16, 8, 8, 448, 448) == 4
    assert ok(14, 16, 4, 4, 672, 672) == 2          # 3584 rows: M / 4 = 896 is not whole tiles, M / 2 = 1792 is
    assert ok(2, 16, 4, 4, 672, 672) == 0           # 512 rows: below the threshold
    assert ok(7, 16, 4, 4, 672, 672) == 0           # M / 2 = 896: not whole tiles either
    assert ok(4, 16, 16, 6, 224, 224) == 2          # W % 4 != 0
    assert ok(64, 16, 16, 16, 224, 96) == 0         # not a width any position tile covers
    # the VQ decoder's widths: at its 16^3 level only, and the same answer at every batch
    assert ok(1, 16, 16, 16, 256, 256) == ok(16, 16, 16, 16, 256, 256) == 4 and ok(1, 32, 32, 32, 128, 128) == 0
    with L.debug_override(no_wino43=1):
        assert ok(64, 16, 16, 16, 224, 224) == 2
    with L.debug_override(no_wino=1):
        assert ok(64, 16, 16, 16, 224, 224) == 0
    p = desc(64, 16, 16, 16, 224, 224)
    p.sw = 2
    assert dll.cs_conv_wino_ok(C.byref(p)) == 0     # strided convs never

    def plan(nb, d, h, w, cin, cout, fmt):
        sk, wsb = C.c_int32(0), C.c_int64(0)
        rc = dll.cs_conv_wino_plan(C.byref(desc(nb, d, h, w, cin, cout, fmt)), C.byref(sk), C.byref(wsb))
        return rc, sk.value, wsb.value
    # 32 objects, 16x4x4 level, F(2,3): 384 position tiles -> two slices = three even rounds of 256 CUs; workspace = slices x 4
    # positions x M / 2 rows
    assert plan(64, 16, 4, 4, 672, 672, 3) == (0, 2, 2 * 4 * 8192 * 672 * 4)
    # 14 samples at the 16^3 level, F(2,3): 448 tiles -> ONE slice (the efficiency-only rule took four and lost 30 %)
    assert plan(14, 16, 16, 16, 224, 224, 3)[:2] == (0, 1)
    # F(4,3): six positions x M / 4 rows
    rc, sk, wsb = plan(64, 16, 8, 8, 448, 448, 4)
    assert rc == 0 and sk == 1 and wsb == 6 * 16384 * 448 * 4
    assert plan(14, 16, 4, 4, 672, 672, 4)[0] != 0      # F(4,3) asked where only F(2,3) is granted: refused
    assert plan(1, 16, 16, 16, 256, 256, 4)[:2] == (0, 1) and plan(16, 16, 16, 16, 256, 256, 4)[1] == 1   # decoder: never sliced

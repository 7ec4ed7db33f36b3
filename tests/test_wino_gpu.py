"""r5: Winograd F(2,3) along W for the 3x3x3 stride-1 convs (CsConvGemm.a_format = 3; every ResBlock conv,
openai_model_3d.py:294-314: GroupNorm32 + SiLU -> conv3 (+ emb row vector) -> GroupNorm32 + SiLU -> conv3 (+ skip)).

The GroupNorm emits the transformed operand (cs_groupnorm_apply_wino16), the library runs the four position GEMMs (3x3x1
kernels over (D, H, W/2), 18 instead of 27 multiply-adds per output) in one launch and the output transform + epilogue in
the split-K reduce kernel's place.  Checked against fp64 at the per-op gate, against the direct form, with every epilogue
term the ResBlock uses (bias, per-sample row vector, residual, GroupNorm partial sums), with K slices, and at the rule's
edges."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _ref_gn_silu(x, g, b, eps=1e-5):
    xd = x.double().permute(0, 4, 1, 2, 3)
    y = F.group_norm(xd, 32, g.double(), b.double(), eps)
    return F.silu(y).permute(0, 2, 3, 4, 1)


def test_groupnorm_emits_the_transformed_operand():
    from commonscenes_amd import lib as L, ops
    nb, d, h, w, c = 2, 4, 4, 8, 64
    x = _rand(nb, d, h, w, c, seed=1) * 2 + 0.3
    g, b = _rand(c, seed=2) + 1.0, _rand(c, seed=3)
    v = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, a_scale=16.0, wino=True)
    assert isinstance(v, ops.Wino16) and tuple(v.hi.shape) == (4, nb, d, h, w // 2, c) and v.a_scale == 8.0
    y = _ref_gn_silu(x, g, b)
    yp = F.pad(y, (0, 0, 1, 1))
    dj = [yp[:, :, :, j:j + w:2] for j in range(4)]
    ref = torch.stack([dj[0] - dj[2], dj[1] + dj[2], dj[2] - dj[1], dj[1] - dj[3]])
    got = (v.hi.double() + v.lo.double()) / v.a_scale
    torch.cuda.synchronize()
    assert rel_l2(got, ref) < 1e-6
    # F(4,3): six images B^T d over W / 4, at a sixteenth of the scale
    v4 = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, a_scale=16.0, wino=4)
    assert v4.variant == 4 and tuple(v4.hi.shape) == (6, nb, d, h, w // 4, c) and v4.a_scale == 1.0
    yp4 = F.pad(y, (0, 0, 1, 3))
    d6 = [yp4[:, :, :, j:j + w:4] for j in range(6)]
    BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
    ref4 = torch.stack([sum(cf * d6[j] for j, cf in enumerate(row) if cf) for row in BT])
    got4 = (v4.hi.double() + v4.lo.double()) / v4.a_scale
    torch.cuda.synchronize()
    assert rel_l2(got4, ref4) < 1e-6
    ops.check_overflow()


CASES = [
    # nb, (d, h, w), cin, cout, note
    (4, (16, 16, 16), 224, 224, "16^3 level, 224 -> 224"),
    (2, (16, 16, 16), 448, 224, "16^3 level, 448 -> 224 (concatenated input)"),
    (16, (16, 8, 8), 448, 448, "16x8x8 level"),
    (64, (16, 4, 4), 672, 672, "16x4x4 level at 32 objects: 384 position tiles -> two K slices"),
    (32, (16, 4, 4), 1344, 672, "16x4x4 level, concatenated input"),
    (8, (16, 16, 16), 128, 128, "VQ decoder width 128 (256x128 tile), never K-sliced"),
    (4, (16, 16, 16), 64, 64, "VQ decoder width 64 (256x64 tile)"),
    (2, (16, 16, 16), 256, 256, "VQ decoder 256 -> 256 at its 16^3 level"),
    (2, (4, 8, 64), 224, 224, "lines of 64 voxels: W / 2 = 32, the slab's widest line"),
]


@pytest.mark.parametrize("variant", [2, 4], ids=["F(2,3)", "F(4,3)"])
@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_winograd_conv_against_fp64_and_the_direct_form(case, variant):
    from commonscenes_amd import lib as L, ops
    nb, sp, cin, cout, note = case
    rows = sp[0] * sp[1] * sp[2]
    x = _rand(nb, *sp, cin, seed=11) * 1.5 + 0.2
    g, b = _rand(cin, seed=12) * 0.2 + 1.0, _rand(cin, seed=13) * 0.2
    wt = _rand(cout, cin, 3, 3, 3, seed=14, scale=(cin * 27) ** -0.5)
    bias = _rand(cout, seed=15)
    emb = _rand(nb, cout, seed=16)
    res = _rand(nb, *sp, cout, seed=17)
    pw = ops.pack_weight(wt, bias, math=L.MATH_F16X3)
    ops.pack_weight_wino(pw, wt)
    assert pw.wino is not None
    with L.debug_override(wino_min_rows=1024, wino43_min_rows=1024, no_wino43=int(variant == 2)):
        got = ops.wants_wino(nb, *sp, pw)
        if variant == 4 and got != 4:
            assert got == 2 and (sp[2] % 4 or (nb * rows // 4) % 256)      # the rule's reasons, nothing else
            pytest.skip("F(4,3) is not granted for this geometry")
        assert got == variant
        s1 = ops.norm_a_scale(float(g.abs().max()), float(b.abs().max()), rows * (cin // 32))
        v = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, a_scale=s1, wino=variant)
        assert v.variant == variant and v.hi.shape[0] == variant + 2
        hn = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, a_scale=s1, split16=True)
        yw = ops.conv_gemm(v, pw, rowvec=emb, rv_rows=rows, res=res, stats=True)
        yd = ops.conv_gemm(hn, pw, rowvec=emb, rv_rows=rows, res=res, stats=True)
        p = ops._wino_desc(nb, *sp, pw)
        p.a_format = 4 if variant == 4 else 3
        sk, wsb = C.c_int32(0), C.c_int64(0)
        assert L.load().cs_conv_wino_plan(C.byref(p), C.byref(sk), C.byref(wsb)) == 0
    torch.cuda.synchronize()
    a = _ref_gn_silu(x, g, b)
    ref = F.conv3d(a.permute(0, 4, 1, 2, 3), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1)
    ref = ref + emb.double()[:, None, None, None, :] + res.double()
    ew, ed = rel_l2(yw, ref), rel_l2(yd, ref)
    print(f"winograd-W {note}: rel-L2 vs fp64 {ew:.2e} (direct form {ed:.2e}), K slices {sk.value}")
    # (F(4,3): transforms with constants up to 8 and 1 / 24 -- about twice the direct form's rounding error, inside the gate)
    assert ew < 1e-6 and ew < (2 if variant == 2 else 3.5) * ed + 2e-7
    if "two K slices" in note and variant == 2:
        assert sk.value == 2 and wsb.value == 2 * 2 * nb * rows * cout * 4
    if "VQ decoder" in note:
        assert sk.value == 1
    # the GroupNorm partial sums its epilogue leaves == the direct form's route to the same statistics
    stw, std = getattr(yw, "cs_stats", None), getattr(yd, "cs_stats", None)
    assert stw is not None and std is not None
    aw = ops.groupnorm_stats_from_parts([(0, stw)], nb, rows, cout, 32, 1e-5, yw.device)
    t = yw.double().reshape(nb, rows, 32, cout // 32)
    mean, var = t.mean(dim=(1, 3)), t.var(dim=(1, 3), unbiased=False)
    assert float(((aw[..., 0].double() - mean).abs() / var.sqrt()).max()) < 2e-6
    assert float((aw[..., 1].double() * (var + 1e-5).sqrt() - 1).abs().max()) < 2e-6
    ops.check_overflow()


def test_the_rule_keeps_small_odd_and_unsupported_launches_on_the_direct_form():
    from commonscenes_amd import lib as L, ops
    wt = _rand(224, 224, 3, 3, 3, seed=21, scale=0.02)
    pw = ops.pack_weight_wino(ops.pack_weight(wt, None, math=L.MATH_F16X3), wt)
    assert ops.wants_wino(4, 16, 16, 16, pw) and ops.wants_wino(1, 16, 8, 8, pw)      # from 1024 rows (the default threshold)
    assert not ops.wants_wino(1, 16, 4, 4, pw)               # 256 rows: below it
    assert not ops.wants_wino(7, 16, 4, 4, pw)               # M / 2 = 896 rows: not whole 256-row tiles per position
    assert not ops.wants_wino(4, 16, 16, 2, pw)              # W / 2 < 2
    with L.debug_override(no_wino=1):
        assert not ops.wants_wino(4, 16, 16, 16, pw)
    with L.debug_override(wino_min_rows=16384):
        assert ops.wants_wino(4, 16, 16, 16, pw) and not ops.wants_wino(2, 16, 16, 16, pw)
    w128 = _rand(128, 128, 3, 3, 3, seed=24, scale=0.02)     # the VQ decoder's widths: at its 16^3 level only
    p128 = ops.pack_weight_wino(ops.pack_weight(w128, None, math=L.MATH_F16X3), w128)
    # (F(4,3) there too, at every batch: the decoder's route does not follow the batch)
    assert ops.wants_wino(1, 16, 16, 16, p128) == ops.wants_wino(16, 16, 16, 16, p128) == 4 and not ops.wants_wino(2, 32, 32, 32, p128)
    w96 = _rand(96, 224, 3, 3, 3, seed=22, scale=0.02)       # not a 224-column width: no Winograd pack at all
    assert ops.pack_weight_wino(ops.pack_weight(w96, None, math=L.MATH_F16X3), w96).wino is None
    # a Wino16 operand on a weight without the pack is an error, not a silent fall-back
    x = _rand(4, 16, 16, 16, 224, seed=23)
    v = ops.groupnorm(x, torch.ones(224, device="cuda"), torch.zeros(224, device="cuda"), 32, 1e-5, L.ACT_SILU, wino=True)
    with pytest.raises(L.CsError):
        ops.conv_gemm(v, ops.pack_weight(wt, None, math=L.MATH_F16X3))


@pytest.mark.parametrize("variant", [2, 4], ids=["F(2,3)", "F(4,3)"])
def test_channel_range_form_feeds_the_halves_of_a_channel_split_conv(variant):
    """unet.py::_res_split: GroupNorm statistics over the whole concatenation, the two input-channel ranges normalised
    separately (cs_groupnorm_apply_wino16_range) and convolved with the halves of the weight packed at the WHOLE tensor's
    scale: their sum is the conv of the concatenation."""
    from commonscenes_amd import lib as L, ops
    nb, sp, C, ks, cout = 4, (16, 8, 8), 448, 224, 224
    rows = sp[0] * sp[1] * sp[2]
    x = _rand(nb, *sp, C, seed=31) * 1.3
    g, b = _rand(C, seed=32) * 0.2 + 1.0, _rand(C, seed=33) * 0.2
    wt = _rand(cout, C, 3, 3, 3, seed=34, scale=(C * 27) ** -0.5)
    bias = _rand(cout, seed=35)
    am = float(wt.abs().max())
    wh = ops.pack_weight(wt[:, :ks].contiguous(), bias, math=L.MATH_F16X3, amax=am)
    ws = ops.pack_weight(wt[:, ks:].contiguous(), None, math=L.MATH_F16X3, amax=am)
    ops.pack_weight_wino(wh, wt[:, :ks].contiguous(), amax=am)
    ops.pack_weight_wino(ws, wt[:, ks:].contiguous(), amax=am)
    assert wh.wino and ws.wino and wh.wino[2][2] == ws.wino[2][2] and wh.wino[4][2] == ws.wino[4][2]
    stats = ops.groupnorm_stats(x, 32, 1e-5)
    cpg = C // 32
    s1 = ops.norm_a_scale(float(g.abs().max()), float(b.abs().max()), rows * cpg)
    a_s = ops.groupnorm_apply_range(x[..., ks:], stats, g[ks:], b[ks:], cpg, ks, L.ACT_SILU, a_scale=s1, wino=variant)
    with L.debug_override(wino43_min_rows=1024):
        y_s = ops.conv_gemm(a_s, ws)
    y = torch.empty(nb, *sp, cout, device="cuda")
    for half in range(2):                         # per sample range, as the guidance halves are
        sl = slice(2 * half, 2 * half + 2)
        a_h = ops.groupnorm_apply_range(x[sl][..., :ks], stats[sl], g[:ks], b[:ks], cpg, 0, L.ACT_SILU, a_scale=s1, wino=variant)
        assert isinstance(a_h, ops.Wino16) and a_h.variant == variant
        with L.debug_override(wino43_min_rows=1024):
            ops.conv_gemm(a_h, wh, res=y_s[sl], out=y[sl])
    torch.cuda.synchronize()
    ref = F.conv3d(_ref_gn_silu(x, g, b).permute(0, 4, 1, 2, 3), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1)
    assert rel_l2(y, ref) < 1e-6
    ops.check_overflow()


def test_tail_plan_of_a_position_launch_that_is_not_whole_rounds():
    """r6: 288 position tiles (the 16x4x4 level at 32 objects, F(4,3): 6 positions x 3 column tiles = 18 units of 16 row tiles) on
    256 CUs -- the plan is a main launch over whole rounds (16 units, unsliced) + a K-sliced tail launch over the other 2 units
    (cs_conv_wino_plan_info); CS_NO_WINO_TAIL=1 is the uniform-slices form.  Both against fp64 at the per-op gate and against
    each other (they differ in the fp32 summation order only), GroupNorm partials included -- and, because the cut is by
    (position, column tile) and never by row, a sample's result does not depend on its place in the batch: the last sample is a
    copy of the first and must come out bit-identical."""
    from commonscenes_amd import lib as L, ops
    nb, sp, cin, cout = 64, (16, 4, 4), 672, 672
    rows = sp[0] * sp[1] * sp[2]
    x = _rand(nb, *sp, cin, seed=21) * 1.5 + 0.2
    g, b = _rand(cin, seed=22) * 0.2 + 1.0, _rand(cin, seed=23) * 0.2
    wt = _rand(cout, cin, 3, 3, 3, seed=24, scale=(cin * 27) ** -0.5)
    bias, emb, res = _rand(cout, seed=25), _rand(nb, cout, seed=26), _rand(nb, *sp, cout, seed=27)
    x[-1], emb[-1], res[-1] = x[0], emb[0], res[0]
    pw = ops.pack_weight(wt, bias, math=L.MATH_F16X3)
    ops.pack_weight_wino(pw, wt)
    s1 = ops.norm_a_scale(float(g.abs().max()), float(b.abs().max()), rows * (cin // 32))
    outs, plans = {}, {}
    for tail in (1, 0):
        with L.debug_override(no_wino_tail=int(not tail)):
            assert ops.wants_wino(nb, *sp, pw) == 4
            v = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, a_scale=s1, wino=4)
            outs[tail] = ops.conv_gemm(v, pw, rowvec=emb, rv_rows=rows, res=res, stats=True)
            p = ops._wino_desc(nb, *sp, pw)
            p.a_format = 4
            sl, um, ut = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            assert L.load().cs_conv_wino_plan_info(C.byref(p), C.byref(sl), C.byref(um), C.byref(ut)) == 0
            plans[tail] = (sl.value, um.value, ut.value)
    torch.cuda.synchronize()
    ops.check_overflow()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    print(f"tail plan {plans[1]}, uniform plan {plans[0]} on {cus} CUs")
    assert plans[0][1] == 0 and plans[0][2] == 18
    if cus == 256:
        assert plans[1] == (8, 16, 18)                     # 256 workgroups for the whole K loop, then 2 units x 16 tiles x 8 slices = 256
    a = _ref_gn_silu(x, g, b)
    ref = F.conv3d(a.permute(0, 4, 1, 2, 3), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 4, 1)
    ref = ref + emb.double()[:, None, None, None, :] + res.double()
    e1, e0 = rel_l2(outs[1], ref), rel_l2(outs[0], ref)
    print(f"tail plan rel-L2 vs fp64 {e1:.2e}, uniform {e0:.2e}, tail vs uniform {rel_l2(outs[1], outs[0]):.2e}")
    # (an unsliced 6048-term fp32 chain rounds a little more than three chains of 2016: 6.8e-7 vs 4.4e-7 measured)
    assert e1 < 1e-6 and e0 < 1e-6 and rel_l2(outs[1], outs[0]) < 1.5e-6
    for o in outs.values():
        assert torch.equal(o[-1], o[0])                    # the batch position does not enter the arithmetic
        st = ops.groupnorm_stats_from_parts([(0, o.cs_stats)], nb, rows, cout, 32, 1e-5, o.device)
        t = o.double().reshape(nb, rows, 32, cout // 32)
        mean, var = t.mean(dim=(1, 3)), t.var(dim=(1, 3), unbiased=False)
        assert rel_l2(st[..., 0], mean) < 1e-5 and rel_l2(st[..., 1], (var + 1e-5).rsqrt()) < 1e-5

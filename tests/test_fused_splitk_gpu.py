"""r5 (ABI 15): the split-K reduce + epilogue INSIDE the slice kernel (CsConvGemm.splitk_sync).

The K slices of an output tile publish their partial tiles, arrive on the tile's counter, and the first R slices then each
reduce a 16-row-aligned share of the tile over all slices in slice order and apply the epilogue -- the arithmetic of the
two-kernel form (slices, then splitk_reduce[_epi]_kernel), element for element.  So every output of the fused launch -- the
fp32 tensor, the interleaved operand pair, the GroupNorm partial sums -- must equal the two-kernel form BIT FOR BIT, on
every kernel family the small-batch plan slices (3x3x3 slab convs on fp32 and pre-split operands, the strided gather
kernel, the long-K pointwise GEMM on fp32 and pair operands), launch after launch (the counters return to zero), and with
other work on the chip (hand-offs must be tested under uneven load, consumer L1 warm: MI355X_MICROARCH.md).
Reference call sites: openai_model_3d.py:294-314 (ResBlock convs), attention.py:241-245 (ff.net.2).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _run(x, pw, fused, **kw):
    from commonscenes_amd import lib as L, ops
    with L.debug_override(no_fused_reduce=0 if fused else 1):
        y = ops.conv_gemm(x, pw, **kw)
    torch.cuda.synchronize()
    return y


CASES = [
    # nb, (d, h, w), cin, cout, k, stride, residual, rowvec, explicit slices, note
    (2, (16, 4, 4), 224, 672, 3, (1, 1, 1), True, True, None, "4^3 level, one object: plan's slices, 16 reducers"),
    (2, (16, 8, 8), 448, 448, 3, (1, 1, 1), True, False, None, "8x8 level, one object"),
    (2, (16, 16, 16), 224, 224, 3, (1, 1, 1), False, True, None, "16^3 level, one object"),
    (14, (16, 4, 4), 224, 672, 3, (1, 1, 1), True, False, None, "4^3 level, seven objects: six slices, four reducers"),
    (2, (16, 8, 8), 96, 224, 3, (1, 1, 1), True, False, 5, "explicit five slices (odd count)"),
    (2, (16, 8, 8), 96, 224, 3, (1, 1, 1), False, False, 3, "three slices, no epilogue terms"),
    (2, (16, 16, 16), 224, 224, 3, (1, 2, 2), False, False, 8, "strided (Downsample) conv: gather kernel, 128-row tile"),
    (3, (10, 5, 5), 64, 224, 3, (1, 1, 1), True, False, 4, "250-row samples: ragged last tile"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_fused_reduce_equals_the_two_kernel_form_bit_for_bit(case):
    from commonscenes_amd import lib as L, ops
    nb, sp, cin, cout, k, stride, with_res, with_rv, splitk, note = case
    d, h, w = sp
    x = _rand(nb, d, h, w, cin, seed=1)
    wt = _rand(cout, cin, k, k, k, seed=2, scale=(cin * k ** 3) ** -0.5)
    pw = ops.pack_weight(wt, _rand(cout, seed=3), math=L.MATH_F16X3)
    ho, wo = (h + 2 - 3) // stride[1] + 1, (w + 2 - 3) // stride[2] + 1
    res = _rand(nb, d, ho, wo, cout, seed=4) + 0.7 if with_res else None
    rv = _rand(nb, cout, seed=5) if with_rv else None
    kw = dict(stride=stride, res=res, rowvec=rv, rv_rows=d * ho * wo)
    if splitk:
        kw["splitk"] = splitk
    for stats in (False, True):
        y0 = _run(x, pw, False, stats=stats, **kw)
        for rep in range(3):                                   # launch after launch: the counters must be back at zero
            y1 = _run(x, pw, True, stats=stats, **kw)
            assert torch.equal(y1, y0), (note, stats, rep)
        if stats and getattr(y0, "cs_stats", None) is not None:
            assert getattr(y1, "cs_stats", None) is not None
            assert torch.equal(y1.cs_stats.part, y0.cs_stats.part), note
            assert y1.cs_stats.tps == y0.cs_stats.tps
    assert int(ops.sync_words().abs().sum().item()) == 0, "arrival counters must return to zero"
    ops.check_overflow()


def test_fused_reduce_on_presplit_and_pair_operands_and_pair_outputs():
    """(a) the GroupNorm -> conv pre-split pair (a_format = 1) on the K-sliced slab kernel; (b) the long-K pointwise GEMM
    ff.net.2 (2688 -> 672, attention.py:245) reading the interleaved pair and writing the interleaved pair for proj_out."""
    from commonscenes_amd import lib as L, ops
    # (a)
    nb, d, h, w, c, cout = 2, 16, 16, 16, 224, 224
    x = _rand(nb, d, h, w, c, seed=21)
    gam, bet = _rand(c, seed=22) + 1.0, _rand(c, seed=23)
    pw = ops.pack_weight(_rand(cout, c, 3, 3, 3, seed=24, scale=(c * 27) ** -0.5), _rand(cout, seed=25), math=L.MATH_F16X3)
    a = ops.groupnorm(x, gam, bet, 32, 1e-5, L.ACT_SILU, split16=True)
    assert isinstance(a, ops.Split16)
    y0 = _run(a, pw, False, splitk=8, stats=True)
    y1 = _run(a, pw, True, splitk=8, stats=True)
    assert torch.equal(y1, y0) and torch.equal(y1.cs_stats.part, y0.cs_stats.part)
    # (b)
    m, cin, cout = 512, 2688, 672
    t1 = _rand(m, cout, seed=26)
    gg = _rand(m, cin, seed=27)
    w2 = ops.pack_weight(_rand(cout, cin, seed=28, scale=cin ** -0.5), _rand(cout, seed=29), math=L.MATH_F16X3)
    wprev = ops.pack_weight(_rand(cin, 64, seed=30, scale=0.125), _rand(cin, seed=31), math=L.MATH_F16X3)
    src = _rand(m, 64, seed=32)
    ggp = ops.linear(src, wprev, out_pair=16.0)                 # a Pair16 operand from a pair-emitting epilogue
    outs = []
    for fused in (False, True):
        o = _run(ggp if isinstance(ggp, ops.Pair16) else gg, w2, fused, res=t1, out_pair=16.0)
        outs.append(o)
    a0, a1 = outs
    assert type(a0) is type(a1)
    t0 = a0.t if isinstance(a0, ops.Pair16) else a0
    t1_ = a1.t if isinstance(a1, ops.Pair16) else a1
    assert torch.equal(t0.view(torch.int32), t1_.view(torch.int32))
    assert int(ops.sync_words().abs().sum().item()) == 0
    ops.check_overflow()


def test_fused_reduce_under_load_and_with_a_warm_l1():
    """Hand-offs fail under UNEVEN load with the consumer's L1 warm, not on an idle chip: run the fused launches back to
    back with streaming kernels in between (which leave the workspace's previous contents in L1 / L2 lines) into the SAME
    workspace addresses with DIFFERENT inputs, and check every word against the two-kernel form each time."""
    from commonscenes_amd import lib as L, ops
    nb, d, h, w, cin, cout = 2, 16, 4, 4, 448, 672
    pw = ops.pack_weight(_rand(cout, cin, 3, 3, 3, seed=41, scale=(cin * 27) ** -0.5), _rand(cout, seed=42), math=L.MATH_F16X3)
    big = _rand(64, 1024, 448, seed=43)
    xs = [_rand(nb, d, h, w, cin, seed=50 + i) for i in range(6)]
    refs = [_run(x, pw, False, stats=True) for x in xs]
    for rnd in range(4):
        for x, r in zip(xs, refs):
            with L.debug_override(no_fused_reduce=0):
                _ = ops.layernorm(big, torch.ones(448, device="cuda"), torch.zeros(448, device="cuda"))   # uneven load in front
                y = ops.conv_gemm(x, pw, stats=True)
                _ = big * 1.0001
            assert torch.equal(y, r), rnd
            assert torch.equal(y.cs_stats.part, r.cs_stats.part)
    torch.cuda.synchronize()
    assert int(ops.sync_words().abs().sum().item()) == 0
    ops.check_overflow()


def test_one_object_unet_step_is_unchanged_and_launches_no_reduce_kernel():
    """The reduced UNet, one object under guidance (CFG batch 2): eps with the fused reduce == eps with the two-kernel form,
    bit for bit, on both hosts."""
    from commonscenes_amd import configs as K, lib as L, synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    cfg = K.UNET_CROSSATTN
    df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math("f16x3")
    df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device="cuda"))
    x = synth.gaussian_like("fz:x", (1, 3, 16, 16, 16)).cuda()
    t = torch.full((1,), 501, dtype=torch.int64, device="cuda")
    c = synth.gaussian_like("fz:c", (2, 1, 1280)).cuda()
    with L.debug_override(no_fused_reduce=1):
        e0 = df.forward_cfg(x, t, c).clone()
    e1 = df.forward_cfg(x, t, c).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(e0).all()
    assert torch.equal(e0, e1)
    df.check_overflow()


@pytest.mark.parametrize("case", [
    (2, 16, 4, 4, 672, 672, (0, 1, 1)),      # one object's 4^3 -> 8x8 Upsample conv (openai_model_3d.py:150-153): 24 tiles x 10 slices
    (2, 16, 8, 8, 448, 448, (0, 1, 1)),      # 8x8 -> 16^3: 64 tiles x 4 slices
    (1, 16, 4, 4, 96, 224, (0, 1, 1)),       # prefix-sized, ragged K slices
    (1, 8, 8, 8, 64, 128, (1, 1, 1)),        # eight classes, two depth taps, 128-column tiles (the decoder's geometry)
], ids=["4x4-672", "8x8-448", "small-224", "3d-128"])
def test_small_folded_upsample_conv_as_one_sliced_class_batch(case):
    """r5: at one or two objects the folded Upsample conv's parity classes are tiny GEMMs; all classes x K slices run as ONE
    launch of the four-tap slab kernel (partial tiles [class][slice]) + ONE reduce-scatter into the doubled grid, instead
    of a K-sliced launch and a reduce per class and the interleave pass.  Same folded weights, another K partition:
    against the fp64 direct form of nearest x2 + conv (vqvae_modules.py:35-39 / openai_model_3d.py:150-153) and against
    the per-class route (CS_NO_UP2_BATCH=1)."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    from conftest import rel_l2
    nb, d, h, w, cin, cout, up = case
    x = _rand(nb, d, h, w, cin, seed=61).cpu()
    wt = _rand(cout, cin, 3, 3, 3, seed=62, scale=(cin * 27) ** -0.5).cpu()
    b = _rand(cout, seed=63).cpu()
    ref = R.conv_ndhwc(x.double(), wt.double(), b.double(), (1, 1, 1), up)
    pk = ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3, fold_up=up)
    xd = x.cuda()
    prof = ops.GEMM_PROFILE = []
    try:
        out = ops.conv_gemm(xd, pk, up=up)
    finally:
        ops.GEMM_PROFILE = None
    out2 = ops.conv_gemm(xd, pk, up=up)
    with L.debug_override(no_up2_batch=1):
        per_class = ops.conv_gemm(xd, pk, up=up)
    torch.cuda.synchronize()
    assert out.shape == ref.shape and torch.equal(out, out2)
    e, ep = rel_l2(out, ref), rel_l2(per_class, ref)
    print(f"small folded Upsample conv {case}: sliced class batch {e:.2e}, per class {ep:.2e}")
    assert e < 1e-6 and e < 3 * ep + 3e-7
    assert rel_l2(out, per_class) < 1e-6
    # placement into a channel slice of a wider buffer; NaNs of a neighbouring sample do not leak through masked taps
    wide = torch.full((*out.shape[:-1], cout + 8), 7.0, device="cuda")
    ops.conv_gemm(xd, pk, up=up, out=wide[..., 4:4 + cout])
    torch.cuda.synchronize()
    assert torch.equal(wide[..., 4:4 + cout], out) and bool((wide[..., :4] == 7).all()) and bool((wide[..., -4:] == 7).all())
    if nb > 1:
        xn = xd.clone()
        xn[1] = float("nan")
        on = ops.conv_gemm(xn, pk, up=up)
        torch.cuda.synchronize()
        assert torch.equal(on[0], out[0]) and bool(torch.isnan(on[1]).all())
    ops.read_status()

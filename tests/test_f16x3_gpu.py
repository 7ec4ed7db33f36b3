"""CS_MATH_F16X3 (fp32 operands as fp16 hi/lo pairs on the fp16 MFMA) against fp64 evaluations, the fp32-MFMA
path and the reference goldens.  The mode must sit at fp32 rounding level, not at fp16 level."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def gate(k_terms: int) -> float:
    """SURVEY 8d's per-op gate is 1e-6 rel-L2 against fp64.  It holds for every op measured here (r3 record:
    profiles/r03_parity_per_op.txt) EXCEPT contractions of >= 10 000 terms: the fp32 accumulation chain itself is then
    at 0.9-1e-6 (fp32-input MFMA kernel, bit-equal to an fma chain: 9.4e-7 at K = 18 144) and F16X3 at 1.5e-6 -- the gate
    there is 2e-6 and "no worse than a few x the fp32 chain"."""
    return 1e-6 if k_terms < 10000 else 2e-6


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


CASES = [
    # nb, d, h, w, cin, cout, k, stride, up, tile
    (2, 4, 8, 8, 32, 224, 3, (1, 1, 1), (0, 0, 0), 2),
    (1, 16, 16, 16, 4, 224, 3, (1, 1, 1), (0, 0, 0), 0),
    (2, 4, 8, 8, 48, 48, 3, (1, 2, 2), (0, 0, 0), 0),
    (2, 4, 4, 4, 32, 32, 3, (1, 1, 1), (0, 1, 1), 0),
    (1, 4, 4, 4, 16, 24, 3, (1, 1, 1), (1, 1, 1), 1),
    (2, 4, 4, 4, 96, 3, 3, (1, 1, 1), (0, 0, 0), 0),
    (3, 4, 4, 4, 40, 72, 1, (1, 1, 1), (0, 0, 0), 1),
    (1, 5, 7, 3, 20, 36, 3, (1, 1, 1), (0, 0, 0), 3),
    (1, 8, 8, 8, 672, 224, 3, (1, 1, 1), (0, 0, 0), 2),      # K = 18144
    (2, 4, 8, 8, 32, 224, 3, (1, 1, 1), (0, 0, 0), 4),       # 256-row, 8-wave tile
    (1, 5, 7, 3, 20, 236, 3, (1, 1, 1), (0, 0, 0), 4),       # ragged M and N on it (scalar epilogue on the edge tile)
    (3, 4, 8, 8, 48, 448, 3, (1, 2, 2), (0, 0, 0), 4),
    (1, 4, 8, 8, 64, 224, 3, (1, 1, 1), (0, 1, 1), 4),
]


@pytest.mark.parametrize("case", CASES)
def test_f16x3_conv_is_fp32_grade(case):
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, d, h, w, cin, cout, k, stride, up, tile = case
    cin_real = 3 if cin == 4 else cin
    x = _rand(nb, d, h, w, cin, seed=1)
    if cin_real != cin:
        x[..., cin_real:] = 0
    wt = _rand(cout, cin_real, k, k, k, seed=2, scale=(cin_real * k ** 3) ** -0.5)
    b = _rand(cout, seed=3)
    ref = R.conv_ndhwc(x[..., :cin_real].double(), wt.double(), b.double(), stride, up)
    xd = x.cuda()
    o16 = ops.conv_gemm(xd, ops.pack_weight(wt.cuda(), b.cuda(), cin_pad=cin, math=L.MATH_F16X3), stride=stride,
                        up=up, tile=tile)
    o32 = ops.conv_gemm(xd, ops.pack_weight(wt.cuda(), b.cuda(), cin_pad=cin), stride=stride, up=up, tile=tile)
    torch.cuda.synchronize()
    e16, e32 = rel_l2(o16, ref), rel_l2(o32, ref)
    assert e16 < gate(cin_real * k ** 3), (e16, e32)
    assert e16 < 4 * e32 + 3e-7, (e16, e32)          # no worse than a few x the fp32 fma chain's own rounding


SLAB_CASES = [
    # nb, d, h, w, cin, cout, tile, epilogue
    (2, 16, 16, 16, 32, 224, 4, "plain"),          # level-0 geometry (W = 16)
    (3, 16, 8, 8, 48, 448, 4, "res"),              # level 1 (W = 8), residual epilogue
    (5, 16, 4, 4, 64, 672, 4, "rowvec"),           # level 2 (W = 4): a 256-row tile spans four samples
    (1, 5, 7, 3, 20, 224, 4, "plain"),             # ragged M (105 rows), cin not a multiple of 16, W = 3
    (1, 3, 5, 64, 16, 64, 7, "plain"),             # the decoder's widest line (W = 64): the wide-slab variant
    (1, 4, 9, 32, 16, 128, 6, "rowvec"),           # W = 32, 256x128 tile
    (2, 6, 9, 11, 24, 64, 7, "res"),               # 256x64 tile, odd extents
    (1, 1, 1, 1, 16, 224, 4, "plain"),             # a single voxel: every tap but the centre is padding
]


@pytest.mark.parametrize("case", SLAB_CASES)
def test_f16x3_slab_conv_is_bit_identical_to_the_gather_path(case):
    """3x3x3 stride-1 convs on the 256-row tiles stage their A operand as one slab per (kd, channel chunk) shared by
    the nine (kh, kw) taps (cs_gemm_f16x3.hip, SLAB).  The chunk order is unchanged, so the result must equal the
    per-tap gather path (the 128x224 / 64x64 tiles, which have no slab variant) bit for bit -- and be fp32-grade."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, d, h, w, cin, cout, tile, epi = case
    x = _rand(nb, d, h, w, cin, seed=31)
    wt = _rand(cout, cin, 3, 3, 3, seed=32, scale=(cin * 27) ** -0.5)
    b = _rand(cout, seed=33)
    kw, rkw = {}, {}
    if epi == "res":
        r = _rand(nb, d, h, w, cout, seed=34)
        kw, rkw = dict(res=r.cuda(), act=L.ACT_SILU), dict(res=r.double(), act="silu")
    elif epi == "rowvec":
        rv = _rand(nb, cout, seed=35)
        kw, rkw = dict(rowvec=rv.cuda(), rv_rows=d * h * w), dict(rowvec=rv.double())
    ref = R.conv_ndhwc(x.double(), wt.double(), b.double(), **rkw)
    pk = ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3)
    xd = x.cuda()
    out_slab = ops.conv_gemm(xd, pk, tile=tile, **kw)
    out_gather = ops.conv_gemm(xd, pk, tile=2 if cout % 224 == 0 else 3, **kw)
    again = ops.conv_gemm(xd, pk, tile=tile, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_slab, out_gather)
    assert torch.equal(out_slab, again)
    assert rel_l2(out_slab, ref) < gate(cin * 27)


def test_f16x3_four_way_k_split_of_the_three_column_tile_convs(monkeypatch):
    """4^3-level convs at large batch (cout = 672: three column tiles, 64 x 3 = 192 workgroups of 256x224 at 32 objects)
    are cut into four K slices by the plan (cs_gemm.hip::split4_large) and keep the slab kernel.  Against the unsplit
    slab kernel; and the rule does not look at the batch size, so a shard's rows equal the whole batch's bit for bit."""
    from commonscenes_amd import lib as L, ops
    import ctypes as C
    nb, cin, cout = 64, 256, 672
    x = _rand(2 * nb, 16, 4, 4, cin, seed=61).cuda()
    wt = _rand(cout, cin, 3, 3, 3, seed=62, scale=(cin * 27) ** -0.5)
    b, rv = _rand(cout, seed=63), _rand(2 * nb, cout, seed=64).cuda()
    pk = ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3)
    p = L.CsConvGemm()
    p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = nb, 16, 4, 4, 16, 4, 4
    p.cin, p.cout, p.kd, p.kh, p.kw, p.sd, p.sh, p.sw, p.pd, p.ph, p.pw = cin, cout, 3, 3, 3, 1, 1, 1, 1, 1, 1
    p.math = L.MATH_F16X3
    sk, wsb = C.c_int32(0), C.c_int64(0)
    assert L.load().cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(wsb)) == 0 and sk.value == 4
    p.nb = 8 * nb
    assert L.load().cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(wsb)) == 0 and sk.value == 4
    kw = dict(rv_rows=256, act=L.ACT_SILU)
    half = ops.conv_gemm(x[:nb], pk, rowvec=rv[:nb], **kw)
    whole = ops.conv_gemm(x, pk, rowvec=rv, **kw)
    with L.debug_override(no_splitk=1):
        unsplit = ops.conv_gemm(x[:nb], pk, rowvec=rv[:nb], **kw)
    torch.cuda.synchronize()
    assert torch.equal(whole[:nb], half)
    assert not torch.equal(half, unsplit) and rel_l2(half, unsplit) < 1e-6      # another summation partition, same grade


def test_f16x3_slab_conv_keeps_samples_independent():
    """the slab holds rows of neighbouring samples (a 256-row tile at the 4^3 level spans four of them): a NaN in one
    sample must not leak into the others through the zeroed-by-mask taps."""
    from commonscenes_amd import lib as L, ops
    x = _rand(6, 4, 4, 4, 32, seed=41)
    wt = _rand(224, 32, 3, 3, 3, seed=42, scale=(32 * 27) ** -0.5)
    pk = ops.pack_weight(wt.cuda(), None, math=L.MATH_F16X3)
    clean = ops.conv_gemm(x.cuda(), pk, tile=4)
    x2 = x.clone()
    x2[2] = float("nan")
    dirty = ops.conv_gemm(x2.cuda(), pk, tile=4)
    torch.cuda.synchronize()
    keep = [0, 1, 3, 4, 5]
    assert torch.equal(dirty[keep], clean[keep])
    assert torch.isnan(dirty[2]).all()


@pytest.mark.parametrize("case", [
    # nb, d, h, w (source grid), cin, cout, up, math
    (2, 6, 5, 7, 24, 224, (0, 1, 1), "f16x3"),       # the UNet's Upsample (dims = 3: H and W doubled)
    (2, 6, 5, 7, 24, 224, (0, 1, 1), "fp32"),
    (1, 4, 5, 3, 16, 64, (1, 1, 1), "f16x3"),        # the VQ decoder's / dims = 4 (all three doubled)
    (3, 2, 2, 2, 20, 36, (1, 1, 1), "fp32"),         # cin not a multiple of 16, cout of 4 only
    (1, 1, 1, 1, 8, 8, (1, 1, 1), "f16x3"),          # a single source voxel
    (1, 16, 8, 8, 32, 448, (0, 1, 1), "f16x3"),      # enough rows for the 256-row tile and the slab-less 12-tap path
])
def test_upsample_conv_folded_onto_the_source_grid(case):
    """nearest x2 + 3x3x3 conv == per output parity class a conv with two pre-summed taps per doubled dim on the source
    grid (cs_conv_gemm_up2).  Against the fp64 direct form, next to the direct kernel path (27 taps, upsampling as
    addressing) it replaces."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, d, h, w, cin, cout, up, math = case
    m = L.MATH_F16X3 if math == "f16x3" else L.MATH_FP32
    x = _rand(nb, d, h, w, cin, seed=51)
    wt = _rand(cout, cin, 3, 3, 3, seed=52, scale=(cin * 27) ** -0.5)
    b = _rand(cout, seed=53)
    ref = R.conv_ndhwc(x.double(), wt.double(), b.double(), (1, 1, 1), up)
    xd = x.cuda()
    pk = ops.pack_weight(wt.cuda(), b.cuda(), math=m, fold_up=up)
    assert pk.classes is not None and len(pk.classes) == 2 ** sum(up)
    out = ops.conv_gemm(xd, pk, up=up)
    out2 = ops.conv_gemm(xd, pk, up=up)
    direct = ops.conv_gemm(xd, ops.pack_weight(wt.cuda(), b.cuda(), math=m), up=up)
    torch.cuda.synchronize()
    assert out.shape == direct.shape == ref.shape
    assert torch.equal(out, out2)
    e, ed = rel_l2(out, ref), rel_l2(direct, ref)
    print(f"upsample conv {case}: folded {e:.2e}, direct {ed:.2e}")
    assert e < 1e-6 and e < 3 * ed + 3e-7
    # placement into a wider (concatenation) buffer: only the slice's columns are written
    wide = torch.full((*out.shape[:-1], cout + 8), 7.0, device="cuda")
    ops.conv_gemm(xd, pk, up=up, out=wide[..., 4:4 + cout])
    torch.cuda.synchronize()
    assert torch.equal(wide[..., 4:4 + cout], out) and bool((wide[..., :4] == 7).all()) and bool((wide[..., -4:] == 7).all())
    with pytest.raises(L.CsError):
        ops.conv_gemm(xd, pk)                      # a folded weight only serves the upsampling it was folded for


def test_f16x3_wide_dynamic_range():
    """weights spanning 1e-5..1, activations with outliers and tiny values: the split keeps absolute accuracy."""
    from commonscenes_amd import lib as L, ops
    m, k, n = 256, 512, 224
    x = _rand(m, k, seed=5)
    x[:, ::7] *= 1e-3
    x[3, 5] = 300.0
    x[9, 100] = -700.0
    wt = _rand(n, k, seed=6, scale=k ** -0.5)
    wt[::3] *= 1e-4
    ref = x.double() @ wt.double().t()
    out = ops.linear(x.cuda(), ops.pack_weight(wt.cuda(), math=L.MATH_F16X3))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    # row-wise: error relative to the row's operand magnitudes (what a dot product can promise)
    err = (out.cpu().double() - ref).abs()
    bound = (x.double().abs() @ wt.double().abs().t()) * 1e-6
    assert bool((err <= bound + 1e-12).all())


def test_f16x3_epilogue_matches_fp32_contract():
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, d, h, w, cin, cout = 2, 4, 4, 4, 32, 48
    x = _rand(nb, d, h, w, cin, seed=4)
    wt = _rand(cout, cin, 3, 3, 3, seed=5, scale=(cin * 27) ** -0.5)
    b, rv, res = _rand(cout, seed=6), _rand(nb, cout, seed=7), _rand(nb, d, h, w, cout, seed=8)
    ref = R.conv_ndhwc(x, wt, b, rowvec=rv, res=res, act="silu")
    out = ops.conv_gemm(x.cuda(), ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3), rowvec=rv.cuda(),
                        rv_rows=d * h * w, res=res.cuda(), act=L.ACT_SILU)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 1e-6


def _g(name):
    p = GOLDEN / f"{name}.npz"
    if not p.exists():
        pytest.skip(f"{p.name} not generated")
    return {k: v for k, v in np.load(p).items()}


def _unet(small):
    from commonscenes_amd import synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from oracle.ref_torch import UNET_FULL, UNET_SMALL
    cfg = dict(UNET_SMALL if small else UNET_FULL, dims=3, use_spatial_transformer=True)
    df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math("f16x3")
    df.load_state_dict(synth.synth_state_dict(unet_param_shapes(cfg), device="cuda"))
    return df


@pytest.mark.parametrize("small", [True, False])
def test_f16x3_unet_vs_reference_golden(small):
    g = _g("unet_small" if small else "unet_full")
    df = _unet(small)
    cu = lambda a: torch.from_numpy(a).cuda()
    eps = df(cu(g["x"]), cu(g["t"]), c_crossattn=[cu(g["ctx"])])
    torch.cuda.synchronize()
    e16 = rel_l2(eps, torch.from_numpy(g["eps"]))
    df.set_math("fp32")
    eps32 = df(cu(g["x"]), cu(g["t"]), c_crossattn=[cu(g["ctx"])])
    torch.cuda.synchronize()
    e32 = rel_l2(eps32, torch.from_numpy(g["eps"]))
    print(f"UNet {'small' if small else 'full'}: rel-L2 vs reference  f16x3 {e16:.3e}   fp32 {e32:.3e}")
    assert e16 < 1e-5 and e32 < 1e-5


def test_f16x3_ddim_full_vs_reference_golden():
    from commonscenes_amd.ddim import DDIMSampler
    from oracle.ref_torch import DIFFUSION, register_schedule
    g = _g("ddim_full")
    df = _unet(False)
    sch = register_schedule(**DIFFUSION)

    class M:
        num_timesteps = 1000
        device = torch.device("cuda")
        alphas_cumprod = sch["alphas_cumprod"]

        def apply_model(self, x, t, c):
            return df(x, t, c_crossattn=[c])

    cu = lambda a: torch.from_numpy(a).cuda()
    k, S = int(g["steps"]), int(g["S"])
    x, _ = DDIMSampler(M()).sample(S=S, batch_size=g["x_T"].shape[0], shape=(3, 16, 16, 16), conditioning=cu(g["c"]),
                                   x_T=cu(g["x_T"]), verbose=False, unconditional_guidance_scale=float(g["scale"]),
                                   unconditional_conditioning=cu(g["uc"]), eta=0.0, max_steps=k)
    torch.cuda.synchronize()
    assert rel_l2(x, torch.from_numpy(g["x"][k - 1])) < 1e-4


def test_f16x3_vq_decode_vs_reference_golden():
    from commonscenes_amd import synth
    from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes
    from oracle.ref_torch import VQ_FULL
    g = _g("vq_decode")
    vq = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math("f16x3")
    vq.load_state_dict(synth.synth_state_dict(vqvae_param_shapes(VQ_FULL, 8192, 3), device="cuda"))
    dec = vq.decode_no_quant(torch.from_numpy(g["latent"]).cuda())
    torch.cuda.synchronize()
    assert int((vq.last_indices.cpu().numpy() != g["indices"]).sum()) == 0
    assert rel_l2(dec, torch.from_numpy(g["dec"])) < 1e-4


@pytest.mark.parametrize("nb,nq,nk,heads,dh", [(2, 256, 256, 8, 84), (1, 1024, 1024, 8, 56), (1, 512, 512, 1, 256),
                                               (2, 200, 72, 4, 32), (1, 130, 3, 2, 12)])
def test_f16x3_attention_is_fp32_grade(nb, nq, nk, heads, dh):
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    c = heads * dh
    qkv = _rand(nb, max(nq, nk), 3 * c, seed=27)
    q, k, v = qkv[:, :nq, 0:c].contiguous(), qkv[:, :nk, c:2 * c].contiguous(), qkv[:, :nk, 2 * c:].contiguous()
    ref = R.attention(q.double(), k.double(), v.double(), heads, dh ** -0.5)
    o16 = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, dh ** -0.5, math=L.MATH_F16X3)
    o32 = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, dh ** -0.5)
    torch.cuda.synchronize()
    e16, e32 = rel_l2(o16, ref), rel_l2(o32, ref)
    assert e16 < 1e-6, (e16, e32)


def test_f16x3_attention_spiked_logits():
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, n, heads, dh = 1, 256, 2, 56
    c = heads * dh
    q, k, v = _rand(nb, n, c, seed=28), _rand(nb, n, c, seed=29), _rand(nb, n, c, seed=30)
    k[:, 200] = q[:, 17] * 40.0
    k[:, 3] = -q[:, 17] * 40.0
    ref = R.attention(q.double(), k.double(), v.double(), heads, dh ** -0.5)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, dh ** -0.5, math=L.MATH_F16X3)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-6


@pytest.mark.parametrize("tile", [2, 4])
def test_f16x3_fused_geglu_projection(tile):
    """ff.net.0.proj + GEGLU gate in one GEMM (act=ACT_GEGLU) == Linear -> chunk -> x * gelu(gate)."""
    from commonscenes_amd import lib as L, ops
    m, c, h = 300, 448, 1792
    x = _rand(m, c, seed=51)
    w = _rand(2 * h, c, seed=52, scale=c ** -0.5)
    b = _rand(2 * h, seed=53) * 0.1
    y = x.double() @ w.double().t() + b.double()
    a, g = y.chunk(2, dim=-1)
    ref = a * torch.nn.functional.gelu(g)
    out = ops.linear(x.cuda(), ops.pack_geglu_weight(w.cuda(), b.cuda()), act=L.ACT_GEGLU, tile=tile)
    torch.cuda.synchronize()
    assert out.shape == (m, h)
    assert rel_l2(out, ref) < 1e-6


@pytest.mark.parametrize("shape,cin,cout,tile", [((2, 4, 8, 8), 32, 224, 2), ((1, 5, 7, 3), 24, 36, 3),
                                                 ((2, 4, 4, 4), 64, 128, 1), ((1, 8, 8, 8), 672, 224, 2)])
def test_f16x3_presplit_activation_path(shape, cin, cout, tile, monkeypatch):
    """GroupNorm+SiLU emitting the fp16 hi/lo pair (cs_groupnorm_apply_split16) -> conv reading it (a_format=1)
    == fp64 GroupNorm -> SiLU -> conv."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    monkeypatch.setattr(ops, "SPLIT16_PRODUCERS", True)
    nb, d, h, w = shape
    x = _rand(nb, d, h, w, cin, seed=61) * 1.5 + 0.3
    g = _rand(cin, seed=62) * 0.2 + 1.0
    b = _rand(cin, seed=63) * 0.1
    wt = _rand(cout, cin, 3, 3, 3, seed=64, scale=(cin * 27) ** -0.5)
    bias = _rand(cout, seed=65)
    groups = 8 if cin % 32 else 32
    ref = R.conv_ndhwc(R.groupnorm_ndhwc(x.double(), g.double(), b.double(), groups, 1e-5, "silu"), wt.double(),
                       bias.double())
    hn = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), groups, 1e-5, L.ACT_SILU, split16=True)
    assert isinstance(hn, ops.Split16) and hn.hi.dtype == torch.float16
    out = ops.conv_gemm(hn, ops.pack_weight(wt.cuda(), bias.cuda(), math=L.MATH_F16X3), tile=tile)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 1e-6


@pytest.mark.parametrize("shape,cin,cout,tile", [((2, 16, 16, 16), 32, 224, 4), ((5, 16, 4, 4), 64, 672, 4),
                                                 ((1, 5, 7, 3), 24, 224, 4), ((1, 3, 5, 64), 16, 64, 7),
                                                 ((1, 4, 9, 32), 16, 128, 6), ((6, 4, 4, 4), 32, 224, 4),
                                                 # r3: the 512-row slab tiles (two row blocks per wave)
                                                 ((1, 3, 5, 64), 16, 64, 8), ((2, 6, 9, 11), 24, 64, 8),
                                                 ((3, 8, 16, 16), 32, 64, 8), ((1, 5, 7, 3), 16, 64, 8),
                                                 ((6, 4, 4, 4), 32, 64, 8), ((1, 4, 9, 32), 16, 128, 9),
                                                 ((2, 8, 16, 16), 32, 256, 9)])
def test_f16x3_presplit_slab_path_is_bit_identical(shape, cin, cout, tile, monkeypatch):
    """the slab kernel fed by the split16 GroupNorm producer (a_format = 1: no conversion in the K loop, masked taps read
    a zero block) equals the same conv on the fp32 GroupNorm output bit for bit (y * 16 is exact), for every slab tile;
    a NaN sample stays confined to itself."""
    from commonscenes_amd import lib as L, ops
    nb, d, h, w = shape
    x = _rand(nb, d, h, w, cin, seed=66) * 1.5 + 0.3
    if nb == 6:
        x[2] = float("nan")
    g, b = _rand(cin, seed=67) * 0.2 + 1.0, _rand(cin, seed=68) * 0.1
    wt = _rand(cout, cin, 3, 3, 3, seed=69, scale=(cin * 27) ** -0.5)
    pk = ops.pack_weight(wt.cuda(), _rand(cout, seed=70).cuda(), math=L.MATH_F16X3)
    groups = 8 if cin % 32 else 32
    monkeypatch.setattr(ops, "SPLIT16_PRODUCERS", True)
    hn16 = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), groups, 1e-5, L.ACT_SILU, split16=True)
    assert isinstance(hn16, ops.Split16)
    hn32 = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), groups, 1e-5, L.ACT_SILU)
    o16 = ops.conv_gemm(hn16, pk, tile=tile)
    o32 = ops.conv_gemm(hn32, pk, tile=tile)
    og = ops.conv_gemm(hn16, pk, tile=2 if cout % 224 == 0 else 3)        # pre-split, per-tap gather tiles
    torch.cuda.synchronize()
    if nb == 6:
        keep = [0, 1, 3, 4, 5]
        assert torch.isnan(o16[2]).all() and torch.isfinite(o16[keep]).all()
        assert torch.equal(o16[keep], o32[keep]) and torch.equal(o16[keep], og[keep])
    else:
        assert torch.isfinite(o16).all()
        assert torch.equal(o16, o32) and torch.equal(o16, og)


@pytest.mark.parametrize("nb,shape,cin,cout,k,extras", [
    (2, (16, 4, 4), 672, 672, 3, True),      # one object's 256-voxel level: 12 output tiles -> 32 K slices
    (2, (16, 8, 8), 448, 448, 3, False),
    (1, (4, 8, 8), 224, 224, 3, True),
    (3, (1, 1, 170), 2688, 672, 1, True),    # linear, ragged M (r4: 1-tap K loops under 128 chunks are not split)
])
def test_f16x3_splitk_matches_fp64_and_unsplit(nb, shape, cin, cout, k, extras):
    """cs_conv_gemm_plan proposes K slices for GEMMs with few output tiles; the sliced result (partials summed in
    slice order + epilogue in the reduce kernel) stays fp32-grade and within rounding of the unsplit kernel."""
    import ctypes as C
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    d, h, w = shape
    x = _rand(nb, d, h, w, cin, seed=71)
    wt = _rand(cout, cin, k, k, k, seed=72, scale=(cin * k ** 3) ** -0.5)
    b = _rand(cout, seed=73)
    m = nb * d * h * w
    res = _rand(nb, d, h, w, cout, seed=74) if extras else None
    rv = _rand(nb, cout, seed=75) if extras else None
    ref = R.conv_ndhwc(x.double(), wt.double(), b.double())
    if extras:
        ref = torch.nn.functional.silu(ref + rv.double().view(nb, 1, 1, 1, cout)) + res.double()
    pw = ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3)
    kw = dict(act=L.ACT_SILU, rowvec=rv.cuda(), rv_rows=d * h * w, res=res.cuda()) if extras else {}
    xd = x.cuda()
    assert ops.SPLITK
    split = ops.conv_gemm(xd, pw, **kw)
    with L.debug_override(no_splitk=1):          # (the CsDebug view: reaches both hosts, restored on exit -- ADVICE r4)
        whole = ops.conv_gemm(xd, pw, **kw)
    assert ops.SPLITK
    torch.cuda.synchronize()
    # the plan really split this shape
    p = L.CsConvGemm()
    p.nb, p.dout, p.hout, p.wout, p.cin, p.cout, p.kd, p.kh, p.kw, p.math = nb, d, h, w, cin, cout, k, k, k, L.MATH_F16X3
    sk, wsb = C.c_int32(), C.c_int64()
    assert L.load().cs_conv_gemm_plan(C.byref(p), C.byref(sk), C.byref(wsb)) == 0
    assert sk.value > 1 and wsb.value == sk.value * m * cout * 4
    assert torch.isfinite(split).all()
    assert rel_l2(split, ref) < gate(cin * k ** 3)
    assert rel_l2(split, whole) < 1.5e-6        # two fp32 summation orders over K up to 18144
    # deterministic: same bits on a second run
    again = ops.conv_gemm(xd, pw, **kw)
    torch.cuda.synchronize()
    assert torch.equal(split, again)


def test_plain_fp16_attention_option_is_reported_not_gated():
    """cs_attn_selfattn_f16 (BASELINE configs[4] "fp16 MFMA attention"): opt-in, reduced precision.  The error is
    REPORTED ("fp16-attention mode: report only") with a loose sanity bound; the default path is untouched."""
    from commonscenes_amd import lib as L, ops, synth
    from oracle import ref_ops as R
    heads, dh, n = 8, 56, 1024
    q, k, v = (_rand(2, n, heads * dh, seed=s) for s in (81, 82, 83))
    ref = R.attention(q.double(), k.double(), v.double(), heads, dh ** -0.5)
    o16 = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, dh ** -0.5, math=L.MATH_F16)
    o3 = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, dh ** -0.5, math=L.MATH_F16X3)
    torch.cuda.synchronize()
    e16, e3 = rel_l2(o16, ref), rel_l2(o3, ref)
    print(f"attention rel-L2 vs fp64: plain fp16 {e16:.2e}, F16X3 {e3:.2e}")
    assert e3 < 1e-6 and 1e-5 < e16 < 2e-3
    # whole UNet with the option on: report the drift against the reference golden
    g = _g("unet_small")
    df = _unet(True)
    cu = lambda a: torch.from_numpy(a).cuda()
    df.set_attention_math("f16")
    eps = df(cu(g["x"]), cu(g["t"]), c_crossattn=[cu(g["ctx"])])
    df.set_attention_math(None)
    base = df(cu(g["x"]), cu(g["t"]), c_crossattn=[cu(g["ctx"])])
    torch.cuda.synchronize()
    ea, eb = rel_l2(eps, torch.from_numpy(g["eps"])), rel_l2(base, torch.from_numpy(g["eps"]))
    print(f"UNet(small) rel-L2 vs reference: fp16 attention {ea:.2e}, default {eb:.2e}")
    assert eb < 1e-5 and ea < 5e-3


def test_f16x3_conv_on_a_tensor_larger_than_4_gib():
    """The activation descriptors are per-workgroup windows, so the tensor itself may exceed the 4 GiB a single
    buffer descriptor spans (e.g. 200+ objects per batch at the 16^3 x 672-channel level).  Checked on the tail of
    a 4.6 GiB tensor against the same conv run on that tail alone (bit-identical: samples never mix)."""
    from commonscenes_amd import lib as L, ops, synth
    nb, d, h, w, cin, cout = 420, 16, 16, 16, 672, 224
    x = torch.empty((nb, d, h, w, cin), dtype=torch.float32, device="cuda")
    assert x.numel() * 4 > 2 ** 32
    for i in range(0, nb, 60):               # fill in slices (the generator materialises fp64 temporaries)
        x[i:i + 60] = synth.tensor_device(f"big:{i}", (min(60, nb - i), d, h, w, cin), 1.0)
    wt = _rand(cout, cin, 3, 3, 3, seed=91, scale=(cin * 27) ** -0.5).cuda()
    b = _rand(cout, seed=92).cuda()
    pw = ops.pack_weight(wt, b, math=L.MATH_F16X3)
    big = ops.conv_gemm(x, pw)
    tail = ops.conv_gemm(x[-3:].contiguous(), pw)
    head = ops.conv_gemm(x[:2].contiguous(), pw)
    torch.cuda.synchronize()
    assert torch.isfinite(big[-3:]).all() and float(big[-1].abs().max()) > 0
    assert rel_l2(big[-3:], tail) < 1e-6 and rel_l2(big[:2], head) < 1e-6


def test_removed_pingpong_tile_code_is_refused():
    """tile code 5 (r2's persistent ping-pong GEMM, removed in r6) is refused, never silently mapped to another kernel"""
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    x = synth.tensor_device("ppo:x", (64, 1024, 448), 1.0)
    pw = ops.pack_weight(synth.tensor_device("ppo:w", (448, 448), 0.05), None, math=L.MATH_F16X3)
    with pytest.raises(L.CsError):
        ops.linear(x, pw, tile=5)


@pytest.mark.parametrize("shape,cin,cout,tile", [((2, 4, 12, 64), 32, 64, 8), ((3, 8, 8, 16), 48, 128, 9)])
def test_f16x3_512_row_slab_tiles_with_residual_epilogue(shape, cin, cout, tile, monkeypatch):
    """tiles 8 / 9 (512 x 64 / 512 x 128: two row blocks per wave over the same slab; the VQ-VAE decoder's 64^3 and 32^3
    levels, r3) with the pipelined residual epilogue: bit for bit against the per-tap gather tile, fp32-grade against
    fp64, deterministic."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    nb, d, h, w = shape
    x = _rand(nb, d, h, w, cin, seed=91) * 1.5 + 0.3
    g, b = _rand(cin, seed=92) * 0.2 + 1.0, _rand(cin, seed=93) * 0.1
    wt = _rand(cout, cin, 3, 3, 3, seed=94, scale=(cin * 27) ** -0.5)
    bias, res = _rand(cout, seed=95), _rand(nb, d, h, w, cout, seed=96)
    pk = ops.pack_weight(wt.cuda(), bias.cuda(), math=L.MATH_F16X3)
    monkeypatch.setattr(ops, "SPLIT16_PRODUCERS", True)
    hn = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 8 if cin % 32 else 32, 1e-6, L.ACT_SILU, split16=True)
    prof = ops.GEMM_PROFILE = []
    try:
        o = ops.conv_gemm(hn, pk, res=res.cuda(), tile=tile)
    finally:
        ops.GEMM_PROFILE = None
    o2 = ops.conv_gemm(hn, pk, res=res.cuda(), tile=tile)
    og = ops.conv_gemm(hn, pk, res=res.cuda(), tile=3)
    torch.cuda.synchronize()
    assert prof[0]["tile"] == tile and prof[0]["slab"] in (32, 64) and prof[0]["pre"]
    assert torch.equal(o, o2) and torch.equal(o, og)
    ref = R.conv_ndhwc(R.groupnorm_ndhwc(x.double(), g.double(), b.double(), 8 if cin % 32 else 32, 1e-6, "silu"),
                       wt.double(), bias.double(), res=res.double())
    assert rel_l2(o, ref) < 1e-6


@pytest.mark.parametrize("m,c,n,tile,kind", [(300, 448, 448, 4, "res"), (300, 448, 1344, 2, ""), (130, 96, 64, 3, ""),
                                              (257, 64, 128, 1, "res"), (520, 448, 896, 2, "geglu"), (300, 672, 224, 0, ""),
                                              (512, 672, 672, 0, "splitk"), (300, 64, 128, 6, ""), (300, 64, 64, 7, "")])
def test_layernorm_pair16_feeds_the_gemm_bit_identically(m, c, n, tile, kind):
    """cs_layernorm_pair16 (ABI 12) writes LayerNorm's output as the INTERLEAVED F16X3 operand pair -- per row and
    16-channel chunk [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15] -- and the GEMM reads it with a_format = 2: no conversion in
    its K loop, the same 64-byte gather pieces.  Must equal fp32 LayerNorm -> in-loop split at the same operand scale bit
    for bit (the split is the same arithmetic, done once by the producer), on every gather tile, with the residual /
    GEGLU epilogues and under split-K; and stay fp32-grade against fp64."""
    from commonscenes_amd import lib as L, ops
    x = _rand(m, c, seed=101) * 2.0 + 0.5
    g, b = _rand(c, seed=102) * 0.2 + 1.0, _rand(c, seed=103) * 0.1
    w = _rand(2 * n if kind == "geglu" else n, c, seed=104, scale=c ** -0.5)
    bias = _rand(w.shape[0], seed=105) * 0.1
    res = _rand(m, n, seed=106)
    sc = ops.norm_a_scale(float(g.abs().max()), float(b.abs().max()), c)
    y32 = ops.layernorm(x.cuda(), g.cuda(), b.cuda())
    yp = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), pair_scale=sc)
    assert isinstance(yp, ops.Pair16) and yp.t.shape == y32.shape and yp.a_scale == sc
    # the pair holds hi + lo of y * scale: reassemble it on the host
    raw = yp.t.view(torch.float16).view(m, c // 16, 4, 8).float()
    back = torch.stack([raw[:, :, 0] + raw[:, :, 1], raw[:, :, 2] + raw[:, :, 3]], dim=2).reshape(m, c) / sc
    assert float((back - y32).abs().max()) <= 2.0 ** -21 * float(y32.abs().max())
    if kind == "geglu":
        pk, kw = ops.pack_geglu_weight(w.cuda(), bias.cuda()), dict(act=L.ACT_GEGLU)
    else:
        pk = ops.pack_weight(w.cuda(), bias.cuda(), math=L.MATH_F16X3)
        kw = dict(res=res.cuda()) if kind == "res" else (dict(splitk=4, tile=2) if kind == "splitk" else {})
    if tile and kind != "splitk":
        kw["tile"] = tile
    o_pair = ops.linear(yp, pk, **kw)
    o_f32 = ops.linear(y32, pk, a_scale=sc, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(o_pair).all() and torch.equal(o_pair, o_f32)
    mu = x.double().mean(1, keepdim=True)
    yr = (x.double() - mu) / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref = yr @ w.double().t() + bias.double()
    if kind == "geglu":
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * torch.nn.functional.gelu(g_)
    elif kind == "res":
        ref = ref + res.double()
    assert rel_l2(o_pair, ref) < 1e-6


def test_folded_upsample_convs_on_the_four_tap_slab_path(tmp_path):
    """r3: the 3x2x2 / 2x2x2 kernels of the folded Upsample convs run the slab path with four taps per kd (TPK = 4: slab of
    BM + W + 1 rows starting at -pad, four-stage weight ring) instead of the per-tap gather.  Same chunk order => bit for
    bit equal to the gather path (CS_NO_SLAB4=1, separate process: the switch is read once per process), incl. the
    W = 4 level where a tile spans samples (a NaN sample stays confined), and fp32-grade against the direct 27-tap form.
    The slab kernel's classes also store straight into the doubled grid (scattered-store epilogue; no scratch tensor, no
    interleave pass) and run as ONE launch over all parity classes: equal to the interleave route (CS_NO_UP2_DIRECT=1) and
    to one launch per class (CS_NO_UP2_BATCH=1) bit for bit, incl. a ragged last tile, an output that is a channel slice of a
    wider buffer and a 4.4 GiB output."""
    import os, subprocess, sys
    from pathlib import Path
    from commonscenes_amd import lib as L, ops, synth
    root = Path(__file__).resolve().parent
    outs = []
    # arms: (slab kernel, ONE launch over all classes, each storing straight into the doubled grid) | per-tap gather +
    # interleave | slab + interleave | slab, scattered store, one launch per class
    for i, arm in enumerate(({}, {"CS_NO_SLAB4": "1"}, {"CS_NO_UP2_DIRECT": "1"}, {"CS_NO_UP2_BATCH": "1"})):
        f = tmp_path / f"slab4_{i}.pt"
        env = dict(os.environ, **arm)
        r = subprocess.run([sys.executable, str(root / "_slab4_worker.py"), str(f)], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(f))
    a = outs[0]
    for b in outs[1:]:
        for k in a:
            if k.endswith(":tile"):
                assert int(a[k]) in (4, 6), (k, int(a[k]))      # the 256-row tiles: the ones with a four-tap slab variant
                continue
            if k.endswith(":pad"):
                assert torch.equal(a[k], torch.full((4,), 7.0)), a[k]   # the scattered store stays inside its columns
                continue
            if k == "unet_hw_w4":
                assert torch.isnan(a[k][1]).all() and torch.isfinite(a[k][0]).all() and torch.isfinite(a[k][2:]).all()
                assert torch.equal(a[k][0], b[k][0]) and torch.equal(a[k][2:], b[k][2:])
            else:
                assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k
    # against the direct form (27 taps on the doubled grid) in this process
    x = synth.tensor_device("s4:x:unet_hw", (48, 16, 8, 8, 48), 1.0)
    w = synth.tensor_device("s4:w:unet_hw", (224, 48, 3, 3, 3), (48 * 27) ** -0.5)
    bb = synth.tensor_device("s4:b:unet_hw", (224,), 0.1)
    direct = ops.conv_gemm(x, ops.pack_weight(w, bb, math=L.MATH_F16X3), up=(0, 1, 1))
    torch.cuda.synchronize()
    assert rel_l2(a["unet_hw"], direct) < 1e-6


@pytest.mark.parametrize("nb,shape,cin,cout", [
    (2, (4, 6, 8), 64, 1),          # the VQ decoder's conv_out shape class
    (3, (5, 7, 3), 224, 3),         # the UNet's eps head, ragged volume
    (1, (4, 4, 4), 32, 4),
    (1, (32, 40, 40), 64, 1),       # enough rows for the 256-row tile (ops.tapcol_tile)
    (2, (16, 40, 40), 224, 3),
])
def test_thin_output_conv_as_tap_columns(nb, shape, cin, cout):
    """3x3x3 convs with <= 4 output channels run as ONE pointwise GEMM with 27 * cout columns + cs_tapsum27 (ABI 13;
    openai_model_3d.py:733-737, vqvae_modules.py:473).  Gates: fp64 conv at the per-op gate, and no worse than the
    implicit-GEMM form of the same conv; borders (zero padding) and sample boundaries are where a wrong tap offset or a
    missing mask shows, hence the ragged / multi-sample volumes."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    d, h, w = shape
    x = _rand(nb, d, h, w, cin, seed=11)
    wt = _rand(cout, cin, 3, 3, 3, seed=12, scale=(cin * 27) ** -0.5)
    b = _rand(cout, seed=13)
    ref = R.conv_ndhwc(x.double(), wt.double(), b.double(), (1, 1, 1), (0, 0, 0))
    assert ops.tapcol_ok(wt, L.MATH_F16X3)
    pk = ops.pack_weight_tapcol(wt.cuda(), b.cuda())
    assert pk.tapcol is not None and pk.tapcol.cout == (27 * cout + 3) // 4 * 4 and pk.cout == cout
    o_tc = ops.conv_gemm(x.cuda(), pk)
    o_ig = ops.conv_gemm(x.cuda(), ops.pack_weight(wt.cuda(), b.cuda(), math=L.MATH_F16X3))
    torch.cuda.synchronize()
    assert o_tc.shape == o_ig.shape == (nb, d, h, w, cout)
    e_tc, e_ig = rel_l2(o_tc, ref), rel_l2(o_ig, ref)
    print(f"thin conv cin={cin} cout={cout}: taps-as-columns {e_tc:.2e}, implicit GEMM {e_ig:.2e}")
    assert e_tc < gate(27 * cin) and e_tc < 2 * e_ig + 1e-7
    assert (o_tc - ref.float().cuda()).abs().max().item() < 2e-5 * ref.abs().max().item()
    # an operand scale from the producer and a strided (channel-slice) output go through the same route
    buf = torch.zeros((nb, d, h, w, 8), device="cuda")
    o2 = ops.conv_gemm(x.cuda(), pk, a_scale=4.0, out=buf[..., 4:4 + cout])
    assert rel_l2(o2, ref) < gate(27 * cin) and float(buf[..., :4].abs().max()) == 0.0
    with pytest.raises(L.CsError):
        ops.conv_gemm(x.cuda(), pk, stride=(1, 2, 2))


@pytest.mark.parametrize("nb,n,heads,dh", [
    (1, 4096, 1, 256),     # the VQ decoder's mid attention
    (2, 1100, 1, 256),     # ragged last key tile (32-key tiles), ragged query tile
    (1, 1030, 2, 132),     # dh < 256 on the 256-wide variant: zero padding of the images; two heads
])
def test_attention_on_presplit_tile_images_is_bit_identical(nb, n, heads, dh):
    """cs_attn_selfattn_f16x3_ws (ABI 13): K / V split into their LDS tile images once per call, streamed by LDS-DMA --
    the same operand values through the same MFMA sequence as the plain entry (in-kernel split): equal bit for bit, and
    fp32-grade against fp64; the overflow report moves to the pre-pass."""
    from commonscenes_amd import lib as L, ops
    from oracle import ref_ops as R
    lib = L.load()
    c = heads * dh
    assert lib.cs_attn_f16x3_ws_bytes(nb, n, n, heads, dh) > 0
    qkv = _rand(nb, n, 3 * c, seed=41).cuda()                 # the fused q | k | v buffer, as the hosts pass it
    q, k, v = qkv[..., 0:c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    ops.read_status()
    new = ops.attention(q, k, v, heads, dh ** -0.5, math=L.MATH_F16X3)
    old = torch.empty_like(new)
    L.check(lib.cs_attn_selfattn_f16x3(q.data_ptr(), k.data_ptr(), v.data_ptr(), old.data_ptr(), nb, n, n, heads, dh,
                                       3 * c, 3 * c, 3 * c, c, dh ** -0.5, ops.status_word().data_ptr(), None),
            "cs_attn_selfattn_f16x3")
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    assert torch.equal(new, old)
    if n <= 1100:
        ref = R.attention(q.double().cpu(), k.double().cpu(), v.double().cpu(), heads, dh ** -0.5)
        assert rel_l2(new, ref) < 1e-6
    # a key beyond the fp16 range of the scaled operands (|k| * 16 >= 65504) is reported by the pre-pass
    k2 = k.clone()
    k2[0, n // 2, 3] = 5000.0
    ops.attention(q, k2, v, heads, dh ** -0.5, math=L.MATH_F16X3)
    assert ops.read_status() & L.STATUS_F16X3_OVERFLOW

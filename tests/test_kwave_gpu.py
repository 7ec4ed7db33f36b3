"""r5: the K-wave kernel (csrc/cs_gemm_kw.hip, tile code 10) -- small one-tap GEMMs with the K loop cut across the four waves
of a workgroup (attention.py:179-245 token GEMMs, openai_model_3d.py:307-313 skip_connection, the time-embedding Linears
at one or two objects).

Same operand values and scales as every F16X3 tile, the K sum partitioned into four contiguous ranges: against fp64 at
the per-op gate, against the one-chain 64x64 tile within fp32 summation order; the single epilogue's extras -- GroupNorm
partial sums (64-row statistics tiles) and the interleaved operand pair -- against a pass over the tensor / the fp32
hand-over (bit for bit: same values converted the same way); auto-selection only where every tile is resident at once."""
import ctypes as C

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _tile_of(x_rows, cin, cout, a_format=0):
    from commonscenes_amd import lib as L
    p = L.CsConvGemm()
    p.nb, p.din, p.hin, p.win, p.dout, p.hout, p.wout = x_rows, 1, 1, 1, 1, 1, 1
    p.cin, p.cout, p.lda, p.ldo, p.ldw = cin, cout, cin, cout, cout
    p.kd = p.kh = p.kw = p.sd = p.sh = p.sw = 1
    p.math, p.a_format = L.MATH_F16X3, a_format
    p.acc_scale, p.a_scale = 1.0, 16.0
    some = 4096
    p.x = p.w = p.w_lo = p.out = some
    t, s = C.c_int32(0), C.c_int32(0)
    assert L.load().cs_conv_gemm_launch_info(C.byref(p), C.byref(t), C.byref(s)) == 0
    return t.value


CASES = [
    # rows, cin, cout, residual, rowvec rows (0 = none), act, note
    (512, 672, 672, True, 256, 0, "level-2 token GEMM with residual + per-sample row vector (attn1.to_out)"),
    (512, 672, 2016, False, 0, 0, "q|k|v: 256 tiles, the resident limit"),
    (2048, 448, 448, True, 0, 0, "level-1 token GEMM"),
    (2, 896, 896, False, 0, 2, "time_embed.2 with SiLU: two rows"),
    (2, 896, 8064, False, 0, 0, "all emb_layers in one GEMM"),
    (8192, 448, 224, False, 0, 0, "level-0 skip_connection: 512 tiles, two rounds of the chip"),
    (32768, 448, 224, False, 0, 0, "2048 tiles -> NOT auto-selected (beyond four rounds), explicit tile 10 still right"),
    (500, 200, 100, True, 0, 0, "ragged M, K and N (cin % 16 != 0, cout % 64 != 0)"),
    (64, 128, 64, False, 0, 0, "eight chunks: two per wave"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_kwave_gemm_against_fp64_and_the_one_chain_tile(case):
    from commonscenes_amd import lib as L, ops
    m, cin, cout, with_res, rvr, act, note = case
    x = _rand(m, cin, seed=1)
    wt = _rand(cout, cin, seed=2, scale=cin ** -0.5)
    b = _rand(cout, seed=3)
    pw = ops.pack_weight(wt, b, math=L.MATH_F16X3)
    res = _rand(m, cout, seed=4) if with_res else None
    rv = _rand((m + rvr - 1) // rvr, cout, seed=5) if rvr else None
    kw = dict(res=res, rowvec=rv, rv_rows=rvr or 1, act=act)
    ref = x.double() @ wt.double().t() + b.double()
    if rv is not None:
        ref = ref + rv.double().repeat_interleave(rvr, dim=0)[:m]
    if act == L.ACT_SILU:
        ref = torch.nn.functional.silu(ref)
    if res is not None:
        ref = ref + res.double()
    y10 = ops.linear(x, pw, tile=10, **kw)
    y3 = ops.linear(x, pw, tile=3, **kw)
    ya = ops.linear(x, pw, **kw)
    torch.cuda.synchronize()
    e10, e3 = rel_l2(y10, ref), rel_l2(y3, ref)
    print(f"K-wave {note}: rel-L2 vs fp64 {e10:.2e} (one-chain tile {e3:.2e})")
    assert e10 < 1e-6 and e10 < 2 * e3 + 2e-7
    assert rel_l2(y10, y3) < 1e-6
    auto = _tile_of(m, cin, cout)
    tiles = ((m + 63) // 64) * ((cout + 63) // 64)
    assert (auto == 10) == (tiles <= 1024 and cin >= 128 and cin % 4 == 0 and cout % 224 == 0), (auto, tiles)
    if auto == 10:
        assert torch.equal(ya, y10)
    with L.debug_override(no_kwave=1):
        assert _tile_of(m, cin, cout) != 10
    ops.check_overflow()


def test_kwave_epilogue_outputs_partials_and_pairs():
    from commonscenes_amd import lib as L, ops
    nb, n, c = 2, 256, 672
    x = _rand(nb * n, c, seed=11)
    pw = ops.pack_weight(_rand(c, c, seed=12, scale=c ** -0.5), _rand(c, seed=13), math=L.MATH_F16X3)
    res = _rand(nb * n, c, seed=14) + 0.5
    y = ops.linear(x, pw, res=res, stats=True, spatial=(nb, n, 1, 1))
    y0 = ops.linear(x, pw, res=res, spatial=(nb, n, 1, 1))
    torch.cuda.synchronize()
    assert _tile_of(nb * n, c, c) == 10
    assert torch.equal(y, y0)
    st = getattr(y, "cs_stats", None)
    assert st is not None and st.tps == n // 64 and st.nch == c
    a = ops.groupnorm_stats_from_parts([(0, st)], nb, n, c, 32, 1e-6, y.device)
    t = y0.double().reshape(nb, n, 32, c // 32)
    mean, var = t.mean(dim=(1, 3)), t.var(dim=(1, 3), unbiased=False)
    refst = torch.stack([mean, 1.0 / (var + 1e-6).sqrt()], dim=-1)
    dm = ((a[..., 0].double() - refst[..., 0]).abs() * refst[..., 1]).max()
    dr = ((a[..., 1].double() - refst[..., 1]).abs() / refst[..., 1]).max()
    assert float(dm) < 2e-7 and float(dr) < 2e-7, (float(dm), float(dr))
    # pair output: the consumer fed with the pair == the consumer fed with the fp32 tensor (it converts the same values)
    w2 = ops.pack_weight(_rand(c, c, seed=15, scale=c ** -0.5), _rand(c, seed=16), math=L.MATH_F16X3)
    yp = ops.linear(x, pw, res=res, out_pair=16.0)
    assert isinstance(yp, ops.Pair16)
    z_pair = ops.linear(yp, w2)
    z_f32 = ops.linear(y0, w2)
    torch.cuda.synchronize()
    assert torch.equal(z_pair.reshape(-1, c), z_f32.reshape(-1, c))      # (y0 carries the 5-d shape `spatial=` gave it)
    # a pair operand in: LayerNorm's interleaved pair feeds the K-wave kernel like it feeds the tile kernels
    gam, bet = _rand(c, seed=17) + 1.0, _rand(c, seed=18)
    ln_pair = ops.layernorm(x, gam, bet, pair_scale=64.0)
    ln_f32 = ops.layernorm(x, gam, bet)
    q_pair = ops.linear(ln_pair, w2, a_scale=64.0)
    q_f32 = ops.linear(ln_f32, w2, a_scale=64.0)
    q_t3 = ops.linear(ln_pair, w2, a_scale=64.0, tile=3)
    torch.cuda.synchronize()
    assert _tile_of(nb * n, c, c, a_format=2) == 10
    assert torch.equal(q_pair, q_f32) and rel_l2(q_pair, q_t3) < 1e-6
    ops.check_overflow()


def test_kwave_flags_an_operand_beyond_the_fp16_range_and_follows_a_magnitude_bound():
    from commonscenes_amd import lib as L, ops
    m, c = 512, 448
    x = _rand(m, c, seed=21)
    wt = _rand(c, c, seed=22, scale=c ** -0.5)
    pw = ops.pack_weight(wt, None, math=L.MATH_F16X3)
    xb = x.clone()
    xb[7, 5] = 5000.0                                   # * 16 > 65504
    ops.clear_status()
    ops.linear(xb, pw)
    with pytest.raises(L.CsOverflowError):
        ops.check_overflow()
    # with the tensor's magnitude bound the same input is fine: the operand scale follows the range (CsConvGemm.a_bound)
    slot = torch.tensor([5000.0], device="cuda")
    y = ops.linear(xb, pw, x_bound=slot)
    torch.cuda.synchronize()
    ops.check_overflow()
    assert rel_l2(y, xb.double() @ wt.double().t()) < 1e-6


@pytest.mark.parametrize("n,c,heads", [(1024, 448, 8), (256, 672, 8)])
def test_small_batch_attention_is_batch_invariant_and_takes_operand_scales(n, c, heads):
    """The UNet's self-attention (attention.py:179-218): the per-query arithmetic does not depend on the workgroup shape --
    the rows of a 2-sample call (four-wave workgroups) equal the same samples' rows inside a 32-sample call (eight-wave
    workgroups) bit for bit -- and the r5 entry with caller-given operand scales (cs_attn_selfattn_f16x3_scaled) is the
    plain entry at the default scales and agrees with it within the split's rounding at other powers of two."""
    from commonscenes_amd import lib as L, ops
    big = _rand(32, n, 3 * c, seed=41)
    small = big[5:7].contiguous()
    scale = (c // heads) ** -0.5
    a_big = ops.attention(big[..., :c], big[..., c:2 * c], big[..., 2 * c:], heads, scale, math=L.MATH_F16X3)
    a_small = ops.attention(small[..., :c], small[..., c:2 * c], small[..., 2 * c:], heads, scale, math=L.MATH_F16X3)
    a_sc = ops.attention(small[..., :c], small[..., c:2 * c], small[..., 2 * c:], heads, scale, math=L.MATH_F16X3,
                         scales=(16.0, 16.0, 16.0))
    torch.cuda.synchronize()
    assert torch.equal(a_small, a_big[5:7])
    assert torch.equal(a_sc, a_small)              # the scaled entry at the default scales is the plain entry
    # fp64 softmax attention on one (sample, head)
    q, k, v = (small[0, :, i * c:i * c + c // heads].double() for i in range(3))
    ref = torch.softmax(q @ k.t() * scale, dim=-1) @ v
    assert rel_l2(a_small[0, :, :c // heads], ref) < 1e-6
    # other power-of-two operand scales: the same result up to the hi / lo split's rounding
    a_s2 = ops.attention(small[..., :c], small[..., c:2 * c], small[..., 2 * c:], heads, scale, math=L.MATH_F16X3,
                         scales=(256.0, 64.0, 1.0))
    torch.cuda.synchronize()
    assert rel_l2(a_s2, a_small) < 1e-6
    ops.check_overflow()

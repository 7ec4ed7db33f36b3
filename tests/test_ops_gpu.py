"""Per-op parity: every C-ABI kernel entry vs its CPU restatement (oracle/ref_ops.py).

Tolerances (SURVEY 8d "parity gates"): per-op 1e-6 rel-L2 for fp32 contractions / norms,
bit-exact for integer/index outputs and pure data movement.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-6      # SURVEY 8d per-op gate (r3: every op of this file measured <= 9.2e-7, profiles/r03_parity_per_op.txt)


def _dev(t):
    return t.cuda()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(scope="module")
def ops():
    from commonscenes_amd import ops as o
    return o


@pytest.fixture(scope="module")
def R():
    from oracle import ref_ops
    return ref_ops


def test_library_loads():
    from commonscenes_amd import lib
    assert lib.load().cs_abi_version() == lib.ABI_VERSION == 18


# ---- implicit-GEMM conv / linear ----------------------------------------------------------------
CONV_CASES = [
    # nb, d, h, w, cin, cout, k, stride, up, tile
    (2, 4, 6, 6, 32, 64, 3, (1, 1, 1), (0, 0, 0), 0),
    (2, 4, 8, 8, 32, 224, 3, (1, 1, 1), (0, 0, 0), 2),
    (1, 16, 16, 16, 4, 224, 3, (1, 1, 1), (0, 0, 0), 0),      # UNet conv_in (cin 3 padded to 4)
    (2, 4, 8, 8, 48, 48, 3, (1, 2, 2), (0, 0, 0), 0),          # Downsample stride (1,2,2)
    (2, 4, 4, 4, 32, 32, 3, (1, 1, 1), (0, 1, 1), 0),          # UNet Upsample (D,2H,2W) + conv
    (1, 4, 4, 4, 16, 24, 3, (1, 1, 1), (1, 1, 1), 1),          # VQ decoder Upsample x2 + conv
    (2, 4, 4, 4, 96, 3, 3, (1, 1, 1), (0, 0, 0), 0),           # UNet out conv (cout 3)
    (1, 8, 8, 8, 64, 1, 3, (1, 1, 1), (0, 0, 0), 0),           # decoder conv_out (cout 1)
    (3, 4, 4, 4, 40, 72, 1, (1, 1, 1), (0, 0, 0), 1),          # 1x1x1 conv
    (1, 5, 7, 3, 20, 36, 3, (1, 1, 1), (0, 0, 0), 3),          # ragged extents, M tail
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_gemm(ops, R, case):
    nb, d, h, w, cin, cout, k, stride, up, tile = case
    cin_real = 3 if cin == 4 else cin
    x = _rand(nb, d, h, w, cin, seed=1)
    if cin_real != cin:
        x[..., cin_real:] = 0
    wt = _rand(cout, cin_real, k, k, k, seed=2, scale=(cin_real * k ** 3) ** -0.5)
    b = _rand(cout, seed=3)
    ref = R.conv_ndhwc(x[..., :cin_real], wt, b, stride, up)
    pw = ops.pack_weight(_dev(wt), _dev(b), cin_pad=cin)
    out = ops.conv_gemm(_dev(x), pw, stride=stride, up=up, tile=tile)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < TOL


def test_conv_epilogue_and_strided_io(ops, R):
    """bias + per-sample row vector + residual, input/outputs living in wider buffers."""
    nb, d, h, w, cin, cout = 2, 4, 4, 4, 32, 48
    x = _rand(nb, d, h, w, cin, seed=4)
    wt = _rand(cout, cin, 3, 3, 3, seed=5, scale=(cin * 27) ** -0.5)
    b = _rand(cout, seed=6)
    rv = _rand(nb, cout, seed=7)
    res = _rand(nb, d, h, w, cout, seed=8)
    ref = R.conv_ndhwc(x, wt, b, rowvec=rv, res=res)
    wide_in = torch.zeros(nb, d, h, w, cin + 16).cuda()
    wide_in[..., 8:8 + cin] = x.cuda()
    wide_out = torch.full((nb, d, h, w, cout + 12), 7.0).cuda()
    pw = ops.pack_weight(_dev(wt), _dev(b))
    ops.conv_gemm(wide_in[..., 8:8 + cin], pw, rowvec=_dev(rv), rv_rows=d * h * w, res=_dev(res),
                  out=wide_out[..., 4:4 + cout])
    torch.cuda.synchronize()
    assert rel_l2(wide_out[..., 4:4 + cout], ref) < TOL
    assert torch.all(wide_out[..., :4] == 7.0) and torch.all(wide_out[..., 4 + cout:] == 7.0)


@pytest.mark.parametrize("act", ["relu", "silu", "gelu"])
def test_linear_scale_shift_act(ops, R, act):
    from commonscenes_amd import lib as L
    m, k, n = 77, 160, 100
    x = _rand(m, k, seed=9)
    wt = _rand(n, k, seed=10, scale=k ** -0.5)
    b = _rand(n, seed=11)
    sc = _rand(n, seed=12).abs() + 0.5
    sh = _rand(n, seed=13)
    ref = R.conv_ndhwc(x[:, None, None, None, :], wt, b, act=act, scale=sc, shift=sh)[:, 0, 0, 0, :]
    pw = ops.pack_weight(_dev(wt), _dev(b))
    code = {"relu": L.ACT_RELU, "silu": L.ACT_SILU, "gelu": L.ACT_GELU}[act]
    out = ops.linear(_dev(x), pw, act=code, scale=_dev(sc), shift=_dev(sh))
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches operand / output transposes (guide rule 16)."""
    n = 96
    eye = torch.eye(n)
    wt = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 100.0   # (out,in): y = x @ wt.T
    pw = ops.pack_weight(_dev(wt))
    out = ops.linear(_dev(eye), pw)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), wt.t().contiguous())


def test_conv_named_entries(ops, R):
    """cs_conv3d_3x3x3_s111 / _s122 / cs_gemm_tokens flat-argument entries."""
    import ctypes as C
    from commonscenes_amd import lib as L
    lib = L.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb, d, h, w, cin, cout = 1, 4, 8, 8, 16, 32
    x = _rand(nb, d, h, w, cin, seed=14)
    wt = _rand(cout, cin, 3, 3, 3, seed=15, scale=0.05)
    b = _rand(cout, seed=16)
    pw = ops.pack_weight(_dev(wt), _dev(b))
    xd = _dev(x)
    o1 = torch.empty(nb, d, h, w, cout, device="cuda")
    L.check(lib.cs_conv3d_3x3x3_s111(xd.data_ptr(), pw.wt.data_ptr(), pw.bias.data_ptr(), o1.data_ptr(),
                                     nb, d, h, w, cin, cout, s), "s111")
    o2 = torch.empty(nb, d, h // 2, w // 2, cout, device="cuda")
    L.check(lib.cs_conv3d_3x3x3_s122(xd.data_ptr(), pw.wt.data_ptr(), pw.bias.data_ptr(), o2.data_ptr(),
                                     nb, d, h, w, cin, cout, s), "s122")
    torch.cuda.synchronize()
    assert rel_l2(o1, R.conv_ndhwc(x, wt, b)) < TOL
    assert rel_l2(o2, R.conv_ndhwc(x, wt, b, stride=(1, 2, 2))) < TOL
    m, k, n = 50, 64, 40
    a = _rand(m, k, seed=17)
    w2 = _rand(n, k, seed=18, scale=0.1)
    r2 = _rand(m, n, seed=19)
    pw2 = ops.pack_weight(_dev(w2))
    o3 = torch.empty(m, n, device="cuda")
    ad, rd = _dev(a), _dev(r2)          # keep the device buffers alive across the raw-pointer call
    L.check(lib.cs_gemm_tokens(ad.data_ptr(), pw2.wt.data_ptr(), None, rd.data_ptr(), o3.data_ptr(),
                               m, k, n, L.ACT_NONE, s), "gemm_tokens")
    torch.cuda.synchronize()
    assert rel_l2(o3, a @ w2.t() + r2) < TOL


def test_conv_gemm_rejects_bad_args(ops):
    import ctypes as C
    from commonscenes_amd import lib as L
    p = L.CsConvGemm()
    assert L.load().cs_conv_gemm(C.byref(p), None) == L.CS_EINVAL
    x = torch.zeros(2, 6, device="cuda")      # cin not a multiple of 4
    with pytest.raises(L.CsError):
        ops.linear(x, ops.pack_weight(torch.zeros(8, 6, device="cuda"), cin_pad=6))
    with pytest.raises(L.CsError):
        ops.linear(torch.zeros(2, 8), ops.pack_weight(torch.zeros(8, 8, device="cuda")))  # CPU tensor


# ---- norms ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,groups,rows,act", [(224, 32, (4, 4, 4), "silu"), (1120, 32, (2, 4, 4), "silu"),
                                                (1344, 32, (2, 2, 4), None), (64, 32, (8, 8, 8), "gelu"),
                                                (256, 32, (4, 4, 4), "swish"), (672, 32, (16, 4, 4), None),
                                                (448, 32, (16, 16, 16), "silu")])       # 3 x 7.3 MB: the three-launch path
def test_groupnorm(ops, R, c, groups, rows, act):
    from commonscenes_amd import lib as L
    nb = 3
    x = _rand(nb, *rows, c, seed=20) * 2.0 + 0.7          # non-zero mean stresses the variance formula
    g = _rand(c, seed=21) * 0.2 + 1.0
    b = _rand(c, seed=22) * 0.1
    eps = 1e-5 if act == "silu" else 1e-6
    # fp64 evaluation of the same formula: the fp32 CPU GroupNorm itself carries ~3e-6 of cancellation
    # error when |mean| >> std, more than the kernel under test (statistics accumulated in fp64)
    ref = R.groupnorm_ndhwc(x.double(), g.double(), b.double(), groups, eps, act)
    ref32 = R.groupnorm_ndhwc(x, g, b, groups, eps, act)
    assert rel_l2(ref32, ref) < 1e-5
    code = {None: L.ACT_NONE, "silu": L.ACT_SILU, "swish": L.ACT_SILU, "gelu": L.ACT_GELU}[act]
    out = ops.groupnorm(_dev(x), _dev(g), _dev(b), groups, eps, code)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("shape", [(2, 16, 16, 16, 224), (2, 8, 8, 8, 1344), (2, 4, 4, 4, 672), (1, 16, 16, 16, 64)])
def test_groupnorm_single_launch_equals_three_launch_path(ops, shape):
    """cs_groupnorm takes the one-launch kernel for tensors of one or two objects; it must agree with the
    statistics + apply launches it replaces (fp64 sums in a different fixed order: the fp32 statistics are expected to
    round identically, the gate allows one ulp)."""
    from commonscenes_amd import lib as L
    lib = L.load()
    nb, c, groups = shape[0], shape[-1], 32
    rows = shape[1] * shape[2] * shape[3]
    assert nb * rows * c * 4 <= 16 << 20                                            # (some shapes take the single launch)
    x = _dev(_rand(*shape, seed=27) * 1.7 + 0.4)
    g, b = _dev(_rand(c, seed=28) * 0.2 + 1.0), _dev(_rand(c, seed=29) * 0.1)
    ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, groups) // 8, dtype=torch.float64, device="cuda")
    st1, st3 = torch.empty(nb, groups, 2, device="cuda"), torch.empty(nb, groups, 2, device="cuda")
    y1, y3 = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.cs_groupnorm(x.data_ptr(), g.data_ptr(), b.data_ptr(), y1.data_ptr(), nb, rows, c, c, c, groups, 1e-5,
                             L.ACT_SILU, ws.data_ptr(), st1.data_ptr(), s), "cs_groupnorm")
    L.check(lib.cs_groupnorm_stats(x.data_ptr(), nb, rows, c, c, groups, 1e-5, ws.data_ptr(), st3.data_ptr(), s), "stats")
    L.check(lib.cs_groupnorm_apply(x.data_ptr(), st3.data_ptr(), g.data_ptr(), b.data_ptr(), y3.data_ptr(), nb, rows, c,
                                   c, c, groups, L.ACT_SILU, s), "apply")
    torch.cuda.synchronize()
    assert float(((st1 - st3).abs() / st3.abs().clamp_min(1e-30)).max()) <= 1.2e-7
    print(f"groupnorm {shape}: stats bit-equal {torch.equal(st1, st3)}, output bit-equal {torch.equal(y1, y3)}")
    assert rel_l2(y1, y3) < 2e-7
    y1b = torch.empty_like(x)
    L.check(lib.cs_groupnorm(x.data_ptr(), g.data_ptr(), b.data_ptr(), y1b.data_ptr(), nb, rows, c, c, c, groups, 1e-5,
                             L.ACT_SILU, ws.data_ptr(), st1.data_ptr(), s), "cs_groupnorm")
    torch.cuda.synchronize()
    assert torch.equal(y1, y1b)                                                     # reproducible run to run


def test_groupnorm_large_rows(ops, R):
    """rows > 64*256 exercises the split-count cap (VQ decoder 32^3 / 64^3 grids)."""
    from commonscenes_amd import lib as L
    x = _rand(1, 32, 32, 32, 64, seed=23) + 3.0
    g = torch.ones(64)
    b = torch.zeros(64)
    ref = R.groupnorm_ndhwc(x.double(), g.double(), b.double(), 32, 1e-6, None)
    out = ops.groupnorm(_dev(x), _dev(g), _dev(b), 32, 1e-6, L.ACT_NONE)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("c", [448, 672, 64, 1280])
def test_layernorm(ops, c):
    m = 300
    x = _rand(m, c, seed=24) * 1.5 + 0.3
    g = _rand(c, seed=25) * 0.2 + 1.0
    b = _rand(c, seed=26) * 0.1
    ref = torch.nn.functional.layer_norm(x, (c,), g, b, 1e-5)
    out = ops.layernorm(_dev(x), _dev(g), _dev(b), 1e-5)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


# ---- attention -----------------------------------------------------------------------------------
@pytest.mark.parametrize("nb,nq,nk,heads,dh", [(2, 256, 256, 8, 84), (1, 1024, 1024, 8, 56), (1, 512, 512, 1, 256),
                                               (2, 200, 72, 4, 32), (1, 130, 3, 2, 12)])
def test_attention(ops, R, nb, nq, nk, heads, dh):
    c = heads * dh
    qkv = _rand(nb, max(nq, nk), 3 * c, seed=27)
    q = qkv[:, :nq, 0:c]
    k = qkv[:, :nk, c:2 * c]
    v = qkv[:, :nk, 2 * c:]
    scale = dh ** -0.5
    ref = R.attention(q, k, v, heads, scale)
    if nq == nk:
        qd = _dev(qkv)          # fused qkv buffer, strided views
        out = ops.attention(qd[..., 0:c], qd[..., c:2 * c], qd[..., 2 * c:], heads, scale)
    else:
        out = ops.attention(_dev(q.contiguous()), _dev(k.contiguous()), _dev(v.contiguous()), heads, scale)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


def test_attention_spiked_logits(ops, R):
    """one dominant key per query forces large running-max jumps in the online softmax."""
    nb, n, heads, dh = 1, 256, 2, 56
    c = heads * dh
    q = _rand(nb, n, c, seed=28)
    k = _rand(nb, n, c, seed=29)
    v = _rand(nb, n, c, seed=30)
    k[:, 200] = q[:, 17] * 40.0
    k[:, 3] = -q[:, 17] * 40.0
    ref = R.attention(q, k, v, heads, dh ** -0.5)
    out = ops.attention(_dev(q), _dev(k), _dev(v), heads, dh ** -0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < TOL


# ---- elementwise / layout ------------------------------------------------------------------------
def test_geglu(ops, R):
    x = _rand(123, 2 * 448, seed=31) * 2
    out = ops.geglu(_dev(x))
    torch.cuda.synchronize()
    assert rel_l2(out, R.geglu(x)) < TOL


def test_concat_and_rowvec(ops):
    a = _rand(2, 3, 4, 4, 32, seed=32)
    b = _rand(2, 3, 4, 4, 20, seed=33)
    out = ops.concat_channels(_dev(a), _dev(b))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.cat([a, b], dim=-1))
    x = _rand(2, 48, 32, seed=34)
    v = _rand(2, 32, seed=35)
    y = ops.add_rowvec_(_dev(x.clone()), _dev(v), 48)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x + v[:, None, :])


def test_layout_roundtrip(ops):
    x = _rand(3, 3, 4, 5, 6, seed=36)
    y = ops.nchw_to_ndhwc(_dev(x), cpad=4)
    torch.cuda.synchronize()
    assert y.shape == (3, 4, 5, 6, 4)
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 4, 1))
    assert torch.all(y[..., 3] == 0)
    z = ops.ndhwc_to_nchw(y, c=3)
    torch.cuda.synchronize()
    assert torch.equal(z.cpu(), x)


def test_timestep_embedding(ops):
    from oracle.ref_torch import timestep_embedding
    t = torch.tensor([1, 11, 501, 991, 999, 0], dtype=torch.int64)
    for dim in (224, 32):
        ref = timestep_embedding(t, dim)
        out = ops.timestep_embedding(_dev(t), dim)
        torch.cuda.synchronize()
        # |t*f| reaches ~1e3: fp32 range reduction differs by a few ulp of the argument
        assert (out.cpu() - ref).abs().max() < 2e-4
        assert rel_l2(out, ref) < 5e-5


@pytest.mark.parametrize("cfg", [True, False])
def test_ddim_update(ops, R, cfg):
    x = _rand(5, 3, 16, 16, 16, seed=37)
    eps = _rand(10 if cfg else 5, 3, 16, 16, 16, seed=38)
    a_t, a_prev = float(np.float32(0.4321)), float(np.float32(0.4567))
    s1m = float(np.sqrt(np.float32(1.0) - np.float32(a_t)))
    ref_x, ref_p = R.ddim_update(x, eps, a_t, a_prev, 0.0, s1m, 3.0, cfg)
    xp, p0 = ops.ddim_cfg_update(_dev(x), _dev(eps), a_t, a_prev, 0.0, s1m, 3.0, cfg)
    torch.cuda.synchronize()
    assert rel_l2(xp, ref_x) < 1e-6
    assert rel_l2(p0, ref_p) < 1e-6


# ---- VQ -------------------------------------------------------------------------------------------
def test_vq_lookup(ops, R):
    z = _rand(2, 16, 16, 16, 4, seed=39)
    z[..., 3] = 0
    cb = (_rand(8192, 3, seed=40) * 0.9)
    idx_ref, zq_ref, d = R.vq(z[..., :3].reshape(-1, 3), cb)
    idx, zq = ops.vq_lookup(_dev(z), _dev(cb))
    torch.cuda.synchronize()
    idx = idx.cpu()
    flips = (idx != idx_ref).nonzero().flatten()
    # a flip is only acceptable on an fp32 near-tie of the distance expression
    for r in flips.tolist():
        assert abs(float(d[r, idx[r]] - d[r, idx_ref[r]])) <= 4e-7 * max(1.0, float(d[r].abs().min()))
    assert len(flips) <= 2
    assert torch.equal(zq.cpu().reshape(-1, 4)[:, :3], cb[idx])


# ---- GCN -------------------------------------------------------------------------------------------
def test_gcn_gather_and_pool(ops, R):
    n_obj, n_tri, d_obj, d_pred, H = 10, 37, 64, 48, 32
    g = torch.Generator().manual_seed(41)
    edges = torch.randint(0, n_obj, (n_tri, 2), generator=g)
    edges[:, 0][edges[:, 0] == 9] = 0        # node 9 never a subject; node 8 appears nowhere
    edges[edges == 8] = 1
    obj = _rand(n_obj, d_obj, seed=42)
    pred = _rand(n_tri, d_pred, seed=43)
    cat = ops.gcn_gather_cat(_dev(obj), _dev(pred), _dev(edges))
    torch.cuda.synchronize()
    ref = torch.cat([obj[edges[:, 0]], pred, obj[edges[:, 1]]], dim=1)
    assert torch.equal(cat.cpu(), ref)                       # pure indexing: bit exact
    new_t = _rand(n_tri, 2 * H + 40, seed=44)
    pooled = ops.gcn_segment_mean(_dev(new_t), _dev(edges), n_obj, H, H + 40)
    torch.cuda.synchronize()
    assert torch.equal(pooled.cpu(), R.gcn_pool(new_t, edges, n_obj, H, H + 40))   # same summation order
    assert torch.all(pooled[8] == 0)
    # the CSR-by-destination path (built once per graph, shared by the layers): same order, same bits
    csr = ops.gcn_csr(_dev(edges), n_obj)
    pooled2 = ops.gcn_segment_mean_csr(_dev(new_t), csr, n_obj, H, H + 40)
    torch.cuda.synchronize()
    assert torch.equal(pooled2, pooled)
    c = csr.cpu()
    assert int(c[3 * n_obj]) == 2 * n_tri                                            # every edge listed twice
    assert int(c[n_obj + 9]) == 0 and int(c[n_obj + 8]) == 0 and int(c[2 * n_obj + 8]) == 0


def test_gcn_csr_pooling_at_c4_graph_size(ops, R):
    """BASELINE configs[3]'s 258-node graph (~1300 triples): CSR pooling == the reference's scatter_add order, bit for
    bit, at H = 256 (the shipped GCN width)."""
    from commonscenes_amd import synth
    g = synth.random_scene_graph(256, seed=3)
    O, T = g["objs"].shape[0], g["triples"].shape[0]
    edges = torch.stack([g["triples"][:, 0], g["triples"][:, 2]], dim=1).contiguous()
    H, Dout = 256, 640
    new_t = _rand(T, 2 * H + Dout, seed=7)
    csr = ops.gcn_csr(_dev(edges), O)
    a = ops.gcn_segment_mean_csr(_dev(new_t), csr, O, H, H + Dout)
    b = ops.gcn_segment_mean(_dev(new_t), _dev(edges), O, H, H + Dout)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a.cpu(), R.gcn_pool(new_t, edges, O, H, H + Dout))


def test_embedding(ops):
    tab = _rand(20, 64, seed=45)
    idx = torch.tensor([3, 0, 19, 3], dtype=torch.int64)
    out = ops.embedding(_dev(tab), _dev(idx))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), tab[idx])


@pytest.mark.parametrize("b,n,m", [(3, 1000, 777), (1, 5, 3000), (2, 2048, 2048), (1, 1, 1)])
def test_chamfer_nm_distance_bit_exact(b, n, m):
    """cs_chamfer_nm_distance == the oracle's fp32 brute force, distances and first-minimum indices bit for bit
    (ragged sizes, duplicated target points -> ties keep the lowest index)."""
    from commonscenes_amd.chamfer import chamferDist, nm_distance
    from oracle import ref_ops as R
    g = torch.Generator().manual_seed(b * 1000 + n + m)
    x1 = torch.randn(b, n, 3, generator=g)
    x2 = torch.randn(b, m, 3, generator=g)
    if m > 10:
        x2[:, m // 2] = x2[:, 3]              # an exact duplicate: the tie must resolve to index 3
        x1[:, 0] = x2[:, 3]
    d, i = nm_distance(x1.cuda(), x2.cuda())
    torch.cuda.synchronize()
    rd, ri = R.chamfer_nm(x1.numpy(), x2.numpy())
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)
    if m > 10:
        assert int(i[0, 0]) == 3 and float(d[0, 0]) == 0.0
    d1, d2 = chamferDist()(x1.cuda(), x2.cuda())
    rd2, _ = R.chamfer_nm(x2.numpy(), x1.numpy())
    assert np.array_equal(d1.cpu().numpy(), rd) and np.array_equal(d2.cpu().numpy(), rd2)


def test_chamfer_self_distance_is_zero_at_full_size():
    from commonscenes_amd.chamfer import chamferDist
    x = torch.randn(4, 20000, 3, generator=torch.Generator().manual_seed(5)).cuda()
    cd = chamferDist()
    d1, d2 = cd(x, x)
    torch.cuda.synchronize()
    assert float(d1.abs().max()) == 0.0 and float(d2.abs().max()) == 0.0
    assert torch.equal(cd.idx1.long(), torch.arange(20000, device="cuda").expand(4, -1))


def test_ddim_update_with_noise_term():
    """eta > 0: x_prev = sqrt(a_prev) * pred_x0 + sqrt(1 - a_prev - sigma^2) * e + sigma * noise (ddim.py:234-243)."""
    from commonscenes_amd import ops
    x = _rand(3, 3, 16, 16, 16, seed=93)
    eps = _rand(6, 3, 16, 16, 16, seed=94)
    noise = _rand(3, 3, 16, 16, 16, seed=95)
    a_t, a_prev, sigma = 0.4321, 0.4567, 0.125
    s1m = float(np.sqrt(np.float32(1.0) - np.float32(a_t)))
    e = eps[:3].double() + 3.0 * (eps[3:].double() - eps[:3].double())
    p0 = (x.double() - s1m * e) / np.sqrt(a_t)
    ref = np.sqrt(a_prev) * p0 + np.sqrt(1.0 - a_prev - sigma ** 2) * e + sigma * noise.double()
    xp, pp = ops.ddim_cfg_update(x.cuda(), eps.cuda(), a_t, a_prev, sigma, s1m, 3.0, True, noise=noise.cuda())
    torch.cuda.synchronize()
    assert rel_l2(xp, ref) < 1e-6 and rel_l2(pp, p0) < 1e-6


def test_sampler_eta_positive_runs_and_is_seed_deterministic():
    """DDIMSampler.sample(eta=0.5): the stochastic branch (sigma_t > 0, torch RNG on the device) -- finite, different
    from the deterministic trajectory, reproducible under torch.manual_seed."""
    import test_model_gpu as T
    from commonscenes_amd import synth
    from commonscenes_amd.ddim import DDIMSampler
    m = T._SamplerModel(T._unet(True))
    x_T = synth.gaussian_like("eta:x", (2, 3, 16, 16, 16)).cuda()
    c = synth.gaussian_like("eta:c", (2, 1, 1280)).cuda()
    uc = synth.gaussian_like("eta:uc", (2, 1, 1280)).cuda()
    kw = dict(S=50, batch_size=2, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T, verbose=False,
              unconditional_guidance_scale=3.0, unconditional_conditioning=uc, max_steps=3)
    outs = []
    for eta, seed in ((0.5, 7), (0.5, 7), (0.0, 7)):
        torch.manual_seed(seed)
        x, _ = DDIMSampler(m).sample(eta=eta, **kw)
        torch.cuda.synchronize()
        outs.append(x)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


def test_groupnorm_apply_range_feeds_channel_ranges_from_one_statistics_pass():
    """cs_groupnorm_apply_range / _split16_range (ABI 11): statistics over a concatenation [h | skip] (672 = 448 + 224
    channels, 21-channel groups: group 21 straddles the seam), then the channels [0, 464) and [464, 672) normalised into
    separate operand tensors -- the second one from the skip tensor itself at HALF the batch (the guidance halves share
    it).  Each must equal the same channels of the one-call GroupNorm bit for bit."""
    from commonscenes_amd import lib as L, ops, synth
    nb, nbs, d, C, ch_h, ks = 4, 2, 4, 672, 448, 464
    h = synth.tensor_device("gr:h", (nb, d, d, d, ch_h), 1.0)
    skip = synth.tensor_device("gr:s", (nbs, d, d, d, C - ch_h), 2.0)
    cat = torch.cat([h, skip.repeat(2, 1, 1, 1, 1)], dim=-1).contiguous()
    g, b = synth.tensor_device("gr:g", (C,), 0.3) + 1.0, synth.tensor_device("gr:b", (C,), 0.2)
    stats = ops.groupnorm_stats(cat, 32, 1e-5)
    cpg = C // 32
    whole = ops.groupnorm_apply_range(cat, stats, g, b, cpg, 0, L.ACT_SILU)      # all channels: the two-pass GroupNorm
    assert rel_l2(ops.groupnorm(cat, g, b, 32, 1e-5, L.ACT_SILU), whole) < 1e-6  # (the one-call entry may take the
                                                                                 # single-launch kernel: other sum order)
    a_h = ops.groupnorm_apply_range(cat[..., :ks], stats, g[:ks], b[:ks], cpg, 0, L.ACT_SILU)
    a_s = ops.groupnorm_apply_range(skip[..., ks - ch_h:], stats, g[ks:], b[ks:], cpg, ks, L.ACT_SILU)
    torch.cuda.synchronize()
    assert a_h.shape == (nb, d, d, d, ks) and a_s.shape == (nbs, d, d, d, C - ks)
    assert torch.equal(a_h, whole[..., :ks])
    assert torch.equal(a_s, whole[:nbs, ..., ks:]) and torch.equal(a_s, whole[nbs:, ..., ks:])
    # the pre-split operand pair carries the same values (y * 16 split into fp16 hi + lo)
    p_s = ops.groupnorm_apply_range(skip[..., ks - ch_h:], stats, g[ks:], b[ks:], cpg, ks, L.ACT_SILU, split16=True)
    assert isinstance(p_s, ops.Split16)
    back = (p_s.hi.float() + p_s.lo.float()) / ops.A_SCALE
    assert float((back - a_s).abs().max()) <= 2.0 ** -20 * float(a_s.abs().max())
    # against the oracle's GroupNorm (fp64) on the concatenation
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(
        cat.double().cpu().permute(0, 4, 1, 2, 3), 32, g.double().cpu(), b.double().cpu(), 1e-5)).permute(0, 2, 3, 4, 1)
    assert rel_l2(a_h, ref[..., :ks]) < 1e-6 and rel_l2(a_s, ref[:nbs, ..., ks:]) < 1e-6

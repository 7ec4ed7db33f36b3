"""cs_unet_step (the native whole-forward driver, csrc/cs_unet.hip) on the MI355X:
  * against the golden vectors generated from the reference UNet3DModel (gate: 1e-5 rel-L2);
  * bit for bit against the Python-sequenced driver (commonscenes_amd.unet.DiffusionUNet), which launches the
    same kernels -- both math modes, plain and classifier-free-guidance-pair batches;
  * workspace accounting (exact size works, one byte less is refused with CS_ENOMEM, nothing is launched)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def _g(name):
    p = GOLDEN / f"{name}.npz"
    if not p.exists():
        pytest.skip(f"{p.name} not generated")
    return {k: v for k, v in np.load(p).items()}


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cfg(small):
    from oracle.ref_torch import UNET_FULL, UNET_SMALL
    return dict(UNET_SMALL if small else UNET_FULL, dims=3, use_spatial_transformer=True)


def _pair(small, math):
    from commonscenes_amd import synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    cfg = _cfg(small)
    sd = synth.synth_state_dict(unet_param_shapes(cfg), device="cuda")
    py = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math(math)
    py.load_state_dict(sd)
    nat = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device="cuda", math=math)
    nat.load_state_dict(sd)
    return py, nat


@pytest.mark.parametrize("small", [True, False])
def test_native_unet_vs_reference_golden(small):
    g = _g("unet_small" if small else "unet_full")
    for math in ("fp32", "f16x3"):
        _, nat = _pair(small, math)
        eps = nat(_cu(g["x"]), _cu(g["t"]), c_crossattn=[_cu(g["ctx"])])
        torch.cuda.synchronize()
        assert rel_l2(eps, torch.from_numpy(g["eps"])) < 1e-5, math
        del nat
        torch.cuda.empty_cache()


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_native_driver_equals_python_driver_bitwise(math):
    from commonscenes_amd import synth
    py, nat = _pair(True, math)
    B = 3
    x = synth.gaussian_like("nat:x", (B, 3, 16, 16, 16)).cuda()
    t = torch.tensor([981, 501, 1], dtype=torch.long).cuda()
    c = synth.gaussian_like("nat:c", (B, 1, 1280)).cuda()
    uc = synth.gaussian_like("nat:uc", (B, 1, 1280)).cuda()
    a = py(x, t, c_crossattn=[c])
    b = nat(x, t, c_crossattn=[c])
    torch.cuda.synchronize()
    assert torch.isfinite(b).all() and torch.equal(a, b)
    c_in = torch.cat([uc, c])
    a2 = py.forward_cfg(x, t, c_in)
    b2 = nat.forward_cfg(x, t, c_in)
    torch.cuda.synchronize()
    assert b2.shape == (2 * B, 3, 16, 16, 16) and torch.equal(a2, b2)
    # the guidance-pair evaluation is the duplicated batch, per sample
    b3 = nat(torch.cat([x, x]), torch.cat([t, t]), c_crossattn=[c_in])
    torch.cuda.synchronize()
    assert torch.equal(b2, b3)
    # state_dict round trip speaks the reference's keys
    sd = nat.state_dict()
    assert list(sd.keys()) == list(py.state_dict().keys())
    assert all(torch.equal(sd[k], v) for k, v in py.state_dict().items())


def test_native_driver_workspace_is_exact_and_checked():
    from commonscenes_amd import lib as L, synth
    _, nat = _pair(True, "f16x3")
    lib = L.load()
    need = int(lib.cs_unet_workspace_bytes(nat._h, 2, 1))
    assert need > 0
    x = synth.gaussian_like("ws:x", (2, 3, 16, 16, 16)).cuda()
    t = torch.tensor([11, 11], dtype=torch.long).cuda()
    ctx = synth.gaussian_like("ws:c", (4, 1280)).cuda()
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    vec = torch.empty((4, nat.ctx_floats), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.cs_unet_context(nat._h, nat._arena.data_ptr(), ctx.data_ptr(), 4, vec.data_ptr(), None, ws.data_ptr(), need, s) == 0
    out = torch.full((4, 3, 16, 16, 16), float("nan"), device="cuda")
    rc = lib.cs_unet_step(nat._h, nat._arena.data_ptr(), x.data_ptr(), t.data_ptr(), vec.data_ptr(), out.data_ptr(),
                          2, 1, None, ws.data_ptr(), need - 256, s)
    assert rc == L.CS_ENOMEM
    rc = lib.cs_unet_step(nat._h, nat._arena.data_ptr(), x.data_ptr(), t.data_ptr(), vec.data_ptr(), out.data_ptr(),
                          2, 1, None, ws.data_ptr(), need, s)
    torch.cuda.synchronize()
    assert rc == 0 and torch.isfinite(out).all()
    assert torch.equal(out, nat.forward_cfg(x, t, ctx.view(4, 1, 1280)))


def test_sampler_over_native_driver_equals_python_driver():
    """DDIMSampler.sample (CFG scale 3) driven through cs_unet_step == driven through the Python sequencer."""
    from commonscenes_amd import synth
    from commonscenes_amd.ddim import DDIMSampler
    from oracle.ref_torch import DIFFUSION, register_schedule
    py, nat = _pair(True, "f16x3")
    sch = register_schedule(**DIFFUSION)

    def model(df):
        class M:
            num_timesteps = 1000
            device = torch.device("cuda")
            alphas_cumprod = sch["alphas_cumprod"]

            def apply_model(self, x, t, c):
                return df(x, t, c_crossattn=[c])

            def apply_model_cfg(self, x, t, c_in):
                return df.forward_cfg(x, t, c_in)
        return M()

    x_T = synth.gaussian_like("ns:x", (2, 3, 16, 16, 16)).cuda()
    c = synth.gaussian_like("ns:c", (2, 1, 1280)).cuda()
    uc = synth.gaussian_like("ns:uc", (2, 1, 1280)).cuda()
    outs = []
    for df in (py, nat):
        x, _ = DDIMSampler(model(df)).sample(50, 2, (3, 16, 16, 16), conditioning=c, x_T=x_T, verbose=False,
                                             unconditional_guidance_scale=3.0, unconditional_conditioning=uc,
                                             eta=0.0, max_steps=3)
        torch.cuda.synchronize()
        outs.append(x)
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_native_driver_concat_family(math):
    """cs_unet_step on the concat family (dims=4, AttentionBlock, condition volume as the 4th input channel):
    reference golden at 1e-5 and bit-equality with the Python sequencer."""
    from commonscenes_amd import synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    from oracle.ref_torch import UNET_CONCAT_SMALL
    g = _g("unet_concat_small")
    cfg = dict(UNET_CONCAT_SMALL)
    sd = synth.synth_state_dict(unet_param_shapes(cfg), device="cuda")
    py = DiffusionUNet(cfg, conditioning_key="concat", device="cuda").set_math(math)
    py.load_state_dict(sd)
    nat = NativeDiffusionUNet(cfg, conditioning_key="concat", device="cuda", math=math)
    nat.load_state_dict(sd)
    x, t, c = _cu(g["x"]), _cu(g["t"]), _cu(g["c"])
    a = py(x, t, c_concat=[c])
    b = nat(x, t, c_concat=[c])
    torch.cuda.synchronize()
    assert rel_l2(b, torch.from_numpy(g["eps"])) < 1e-5
    assert torch.equal(a, b)
    c2 = torch.cat([c.flip(0), c])
    assert torch.equal(py.forward_cfg(x, t, c2), nat.forward_cfg(x, t, c2))


def test_product_model_with_native_drivers_equals_python_drivers(tmp_path, monkeypatch):
    """CS_UNET_DRIVER=native: Sg2ScVAEModel.sample(gen_shape=True) through cs_unet_step + cs_vqvae_decode equals the
    Python-sequenced run bit for bit (8 objects, mini-batches 7 + 1, 2 DDIM steps)."""
    import test_model_gpu as T
    g = T._g("e2e_small")
    outs = []
    for drv in ("python", "native"):
        monkeypatch.setenv("CS_UNET_DRIVER", drv)
        m = T._scene(tmp_path)
        assert type(m.Diff.df).__name__ == ("NativeDiffusionUNet" if drv == "native" else "DiffusionUNet")
        assert type(m.Diff.vqvae).__name__ == ("NativeVQVAE" if drv == "native" else "VQVAE")
        O = g["objs"].shape[0]
        dec_sdfs = torch.zeros(O, 1, 4, 4, 4)
        dec_sdfs[torch.from_numpy(g["dec_sdfs_nonzero"])] = 1.0
        m.Diff.mini_B = 7
        boxes, gen = m.sample(None, np.zeros(64), np.eye(64), torch.from_numpy(g["objs"]), torch.from_numpy(g["triples"]),
                              dec_sdfs, torch.from_numpy(g["text_feats"]), torch.from_numpy(g["rel_feats"]),
                              gen_shape=True, z=torch.from_numpy(g["z"]), x_T=torch.from_numpy(g["x_T"]), ddim_steps=2)
        torch.cuda.synchronize()
        outs.append(gen)
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])
    sub, ref = outs[1][:, :, ::2, ::2, ::2], torch.from_numpy(g["gen_sdf_sub"])
    assert sorted(rel_l2(sub[i], ref[i]) for i in range(8))[5] < 1e-4


def test_native_driver_equals_python_driver_at_the_benchmark_batch():
    """r6: the two hosts at BASELINE configs[2]'s own size -- 32 objects under classifier-free guidance = UNet batch 64 at the
    shipped width, the PRODUCT's channel-split threshold (65536 rows) -- where the plan differs from every small-batch test:
    Winograd F(4,3) position launches with the main + K-sliced tail plan at the 16x4x4 level, unsliced whole rounds elsewhere,
    128-row workgroup pairs for the one-tap GEMMs (auto_tile rule iv), CFG-shared skip halves.  cs_unet_step must equal the
    Python sequencer bit for bit there too (both ask the same C plan functions), and equal-conditioned objects must come out
    equal whatever their place in the batch."""
    from commonscenes_amd import lib as L, ops, synth
    with L.debug_override(cfg_split_min_rows=65536):
        py, nat = _pair(False, "f16x3")
        py.split_min_rows = 65536
        B = 32
        x = synth.gaussian_like("nat64:x", (1, 3, 16, 16, 16)).cuda().repeat(B, 1, 1, 1, 1)
        t = torch.full((B,), 501, dtype=torch.long).cuda()
        c = synth.gaussian_like("nat64:c", (B, 1, 1280)).cuda()
        uc = synth.gaussian_like("nat64:uc", (B, 1, 1280)).cuda()
        c[29], uc[29] = c[2], uc[2]                       # a twin at another place in the batch
        c_in = torch.cat([uc, c])
        a = py.forward_cfg(x, t, c_in)
        b = nat.forward_cfg(x, t, c_in)
        torch.cuda.synchronize()
        ops.check_overflow()
    assert a.shape == (2 * B, 3, 16, 16, 16) and torch.isfinite(b).all()
    assert torch.equal(a, b)
    assert torch.equal(b[2], b[29]) and torch.equal(b[B + 2], b[B + 29])
    assert not torch.equal(b[B + 2], b[B + 3])

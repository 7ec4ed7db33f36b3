"""Long-horizon and failure-mode parity on the MI355X:

  * WHOLE sampling trajectories against the reference's own sampler loop -- BASELINE configs[1] (one object, 50 CFG DDIM
    steps) at reduced width and at the SHIPPED width (413.5 M parameters), in both GEMM numerics modes, with the
    per-step growth of the deviation written to gpurun_out/parity_trajectories.json;
  * the PLMS sampler (samplers/plms.py) over its whole 50-step run;
  * the F16X3 overflow story: a sticky device flag, surfaced as CsOverflowError / an automatic fp32 re-run;
  * scene-graph index errors raise IndexError like the reference's nn.Embedding;
  * BASELINE configs[3]'s workload (256 objects) on one GPU as size-independent properties.
"""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, rel_l2

pytestmark = pytest.mark.gpu


def _g(name):
    p = GOLDEN / f"{name}.npz"
    if not p.exists():
        pytest.skip(f"{p.name} not generated")
    return {k: v for k, v in np.load(p).items()}


def _report(key, value):
    """merge one entry into gpurun_out/parity_trajectories.json (kept by gpurun; copied to profiles/ by hand)."""
    d = ROOT / "gpurun_out"
    try:
        d.mkdir(exist_ok=True)
        f = d / "parity_trajectories.json"
        cur = json.loads(f.read_text()) if f.exists() else {}
        cur[key] = value
        f.write_text(json.dumps(cur, indent=1, sort_keys=True))
    except OSError:
        pass


def _model(small, math, route="split"):
    from test_model_gpu import _SamplerModel, _unet
    return _SamplerModel(_unet(small, math, route))


def _run_traj(name, sampler_cls, small, math, route="split"):
    g = _g(name)
    m = _model(small, math, route)
    B = g["c"].shape[0]
    x_T = torch.from_numpy(g["x_T"]).cuda()
    if x_T.shape[0] != B:
        x_T = x_T.repeat(B, 1, 1, 1, 1)
    x, inter = sampler_cls(m).sample(S=int(g["S"]), batch_size=B, shape=(3, 16, 16, 16),
                                     conditioning=torch.from_numpy(g["c"]).cuda(), x_T=x_T, verbose=False,
                                     unconditional_guidance_scale=float(g["scale"]),
                                     unconditional_conditioning=torch.from_numpy(g["uc"]).cuda(), eta=0.0,
                                     log_every_t=1)
    torch.cuda.synchronize()
    from commonscenes_amd import ops
    ops.check_overflow()                                   # the run must not have left the fp16 range
    xi = inter["x_inter"]
    assert len(xi) == int(g["S"]) + 1
    dev = {int(k): rel_l2(xi[int(k)], torch.from_numpy(g["x"][i])) for i, k in enumerate(g["keep"])}
    p0 = rel_l2(inter["pred_x0"][-1], torch.from_numpy(g["pred_x0_final"]))
    return dev, p0


@pytest.mark.parametrize("route", ["split", "product"])
@pytest.mark.parametrize("math", ["f16x3", "fp32"])
@pytest.mark.parametrize("small", [True, False])
def test_ddim_whole_trajectory_vs_reference_golden(small, math, route):
    """C2: one object, all 50 classifier-free-guided DDIM steps through DDIMSampler.sample (ddim.py:60-179) vs the
    reference's own loop.  Gate: rel-L2 <= 1e-4 at EVERY kept step (SURVEY 8d "k-step DDIM latent 1e-4"), final
    latent and final pred_x0 included; the per-step growth is reported."""
    from commonscenes_amd.ddim import DDIMSampler
    name = "traj_small" if small else "traj_full"
    if route == "product" and math != "f16x3":
        pytest.skip("the product-threshold route is run in the product's math mode")
    dev, p0 = _run_traj(name, DDIMSampler, small, math, route)
    print(f"[{name} {math} {route}] rel-L2 by step: " + ", ".join(f"{k}:{e:.2e}" for k, e in dev.items()) + f"; pred_x0 {p0:.2e}")
    _report(f"{name}:{math}" + ("" if route == "split" else ":product_route"), dict(per_step=dev, pred_x0_final=p0))
    assert max(dev.values()) < 1e-4, dev
    assert dev[50] < 1e-4 and p0 < 1e-4


@pytest.mark.parametrize("math,route", [("f16x3", "split"), ("f16x3", "product"), ("fp32", "split")])
def test_ddim_100_step_trajectory_at_the_shipped_width_vs_reference_golden(math, route):
    """r6: BASELINE configs[2]'s own schedule depth -- the reference's DDIMSampler.sample(S=100)
    (sdfusion_txt2shape_model.py:128,460; ddim.py:126-179), one object at the SHIPPED width (413.5 M parameters), x after
    steps {1, 2, 5, 10, 25, 50, 75, 100} + the final pred_x0.  Gate: rel-L2 <= 1e-4 at every kept step."""
    from commonscenes_amd.ddim import DDIMSampler
    dev, p0 = _run_traj("traj100_full", DDIMSampler, False, math, route)
    print(f"[traj100_full {math} {route}] rel-L2 by step: " + ", ".join(f"{k}:{e:.2e}" for k, e in dev.items()) + f"; pred_x0 {p0:.2e}")
    _report(f"traj100_full:{math}" + ("" if route == "split" else ":product_route"), dict(per_step=dev, pred_x0_final=p0))
    assert sorted(dev) == [1, 2, 5, 10, 25, 50, 75, 100]
    assert max(dev.values()) < 1e-4, dev
    assert dev[100] < 1e-4 and p0 < 1e-4


def test_c3_benchmark_batch_at_full_depth_carries_the_reference_trajectory():
    """r6: BASELINE configs[2] itself -- 32 objects, 100 CFG DDIM steps, the UNet at batch 64 (the workload bench.py times, on the
    product's own route: Winograd F(4,3) position GEMMs with the tail plan, channel-split ResBlocks, static operand scales) --
    ASSERTED, not only timed: object 17 of the batch is the reference's one-object S=100 run (`traj100_full`: its x_T is the
    call's shared x_T, its c / uc sit in row 17), the other 31 objects carry their own conditioning.  Samples are independent
    (sdfusion_txt2shape_model.py:459-516 loops over mini-batches of the same sampler), so row 17 must follow the golden at every
    kept step <= 1e-4; a copy of the golden's conditioning in row 3 must come out bit-identical to row 17 (results do not
    depend on the place in the batch); every other object must differ."""
    from commonscenes_amd import ops, synth
    from commonscenes_amd.ddim import DDIMSampler
    g = _g("traj100_full")
    m = _model(False, "f16x3", "product")
    B, k, twin = 32, 17, 3
    c = synth.gaussian_like("c3depth:c", (B, 1, 1280)).cuda()
    uc = synth.gaussian_like("c3depth:uc", (B, 1, 1280)).cuda()
    for r in (k, twin):
        c[r] = torch.from_numpy(g["c"][0]).cuda()
        uc[r] = torch.from_numpy(g["uc"][0]).cuda()
    x_T = torch.from_numpy(g["x_T"]).cuda().repeat(B, 1, 1, 1, 1)
    x, inter = DDIMSampler(m).sample(S=100, batch_size=B, shape=(3, 16, 16, 16), conditioning=c, x_T=x_T, verbose=False,
                                     unconditional_guidance_scale=float(g["scale"]), unconditional_conditioning=uc,
                                     eta=0.0, log_every_t=1)
    torch.cuda.synchronize()
    ops.check_overflow()
    xi = inter["x_inter"]
    assert len(xi) == 101 and xi[-1].shape[0] == B
    dev = {int(s): rel_l2(xi[int(s)][k:k + 1], torch.from_numpy(g["x"][i])) for i, s in enumerate(g["keep"])}
    p0 = rel_l2(inter["pred_x0"][-1][k:k + 1], torch.from_numpy(g["pred_x0_final"]))
    print("[c3 32 objects x 100 steps, object 17 vs traj100_full] rel-L2 by step: "
          + ", ".join(f"{s}:{e:.2e}" for s, e in dev.items()) + f"; pred_x0 {p0:.2e}")
    _report("c3_batch64_100_steps:object17_vs_traj100_full", dict(per_step=dev, pred_x0_final=p0))
    assert max(dev.values()) < 1e-4 and p0 < 1e-4, dev
    assert torch.equal(x[k], x[twin])
    others = [r for r in range(B) if r not in (k, twin)]
    assert min(rel_l2(x[r:r + 1], x[k:k + 1]) for r in others) > 1e-2


def test_plms_whole_trajectory_vs_reference_golden():
    """N4: PLMSSampler (samplers/plms.py:61-236) over its whole 50-step run, B=2, CFG 3.0, reduced width: pseudo
    improved Euler start-up (two model evaluations) + Adams-Bashforth orders 2-4, fused into cs_plms_update."""
    from commonscenes_amd.plms import PLMSSampler
    for math in ("f16x3", "fp32"):
        dev, p0 = _run_traj("plms_small", PLMSSampler, True, math)
        print(f"[plms_small {math}] rel-L2 by step: " + ", ".join(f"{k}:{e:.2e}" for k, e in dev.items()))
        _report(f"plms_small:{math}", dict(per_step=dev, pred_x0_final=p0))
        assert max(dev.values()) < 1e-4 and p0 < 1e-4, dev


def test_plms_update_kernel_vs_oracle_forms():
    """cs_plms_update's five modes against the reference's expressions evaluated in fp32 by torch on the host."""
    from commonscenes_amd import ops, synth
    n = (3, 3, 16, 16, 16)
    x = synth.gaussian_like("pl:x", n)
    eps = synth.gaussian_like("pl:e", (6, 3, 16, 16, 16))
    h = [synth.gaussian_like(f"pl:h{i}", n) for i in range(3)]
    a_t, a_prev, s1m, scale = 0.3217, 0.4411, float(np.sqrt(np.float32(1 - 0.3217))), 3.0
    e_uc, e_c = eps.chunk(2)
    e = e_uc + scale * (e_c - e_uc)
    forms = {ops.PLMS_PLAIN: e, ops.PLMS_AB2: (3 * e - h[0]) / 2, ops.PLMS_AB3: (23 * e - 16 * h[0] + 5 * h[1]) / 12,
             ops.PLMS_AB4: (55 * e - 59 * h[0] + 37 * h[1] - 9 * h[2]) / 24, ops.PLMS_EULER_AVG: (h[0] + e) / 2}
    need = {ops.PLMS_PLAIN: 0, ops.PLMS_AB2: 1, ops.PLMS_AB3: 2, ops.PLMS_AB4: 3, ops.PLMS_EULER_AVG: 1}
    at = torch.full((3, 1, 1, 1, 1), a_t)
    ap = torch.full((3, 1, 1, 1, 1), a_prev)
    for mode, ep in forms.items():
        p0 = (x - torch.full_like(at, s1m) * ep) / at.sqrt()
        xp = ap.sqrt() * p0 + (1.0 - ap).sqrt() * ep
        gx, gp, ge = ops.plms_update(x.cuda(), eps.cuda(), [t.cuda() for t in h[:need[mode]]], mode, a_t, a_prev, s1m,
                                     scale, True)
        torch.cuda.synchronize()
        assert torch.equal(ge.cpu(), e), mode                       # the guidance combine is bit-exact
        assert rel_l2(gx, xp) < 1e-6 and rel_l2(gp, p0) < 1e-6, mode
        assert (gx.cpu() - xp).abs().max() <= 4e-6 * xp.abs().max(), mode


# ----------------------------------------------------------------------------------------------------------------------
# F16X3 overflow
# ----------------------------------------------------------------------------------------------------------------------
def test_f16x3_gemm_raises_the_overflow_flag_at_1e4():
    """|a| = 1e4 cannot be carried as fp16(a * 16): the kernel must say so (sticky CS_STATUS_F16X3_OVERFLOW), and
    only then; fp32-mode GEMMs never touch the flag."""
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    ops.read_status()                                                   # clear
    w = synth.tensor("ov.weight", (224, 64, 3, 3, 3), (3.0 / (64 * 27)) ** 0.5).cuda()
    pw16 = ops.pack_weight(w, None, math=L.MATH_F16X3)
    pw32 = ops.pack_weight(w, None, math=L.MATH_FP32)
    x = synth.gaussian_like("ov:x", (2, 8, 8, 8, 64)).cuda()
    ref = ops.conv_gemm(x, pw32)
    out = ops.conv_gemm(x, pw16)
    torch.cuda.synchronize()
    assert ops.read_status() == 0 and rel_l2(out, ref) < 1e-5
    for big, expect in ((4000.0, 0), (1.0e4, L.STATUS_F16X3_OVERFLOW), (-1.0e4, L.STATUS_F16X3_OVERFLOW),
                        (float("inf"), L.STATUS_F16X3_OVERFLOW)):
        xb = x.clone()
        xb[1, 3, 4, 5, 17] = big
        ops.conv_gemm(xb, pw16)
        torch.cuda.synchronize()
        assert ops.read_status(reset=False) == expect, big
        if expect:
            with pytest.raises(L.CsOverflowError):
                ops.check_overflow()
            assert ops.read_status() == 0                               # check_overflow cleared it
            ok = ops.conv_gemm(xb, pw32)                                # the fp32-input MFMA path handles the value
            torch.cuda.synchronize()
            assert ops.read_status() == 0
            if np.isfinite(big):
                refb = torch.nn.functional.conv3d(xb.permute(0, 4, 1, 2, 3).double().cpu(), w.double().cpu(), padding=1)
                assert rel_l2(ok.permute(0, 4, 1, 2, 3), refb) < 1e-5
    # small tiles / split-K / pointwise shapes report too
    lin = ops.pack_weight(synth.tensor("ov.lin", (96, 64), 0.1).cuda(), None, math=L.MATH_F16X3)
    v = synth.gaussian_like("ov:v", (5, 64)).cuda()
    v[2, 9] = 7.0e3
    ops.linear(v, lin)
    torch.cuda.synchronize()
    assert ops.read_status() == L.STATUS_F16X3_OVERFLOW


def test_f16x3_attention_raises_the_overflow_flag():
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    ops.read_status()
    q = synth.gaussian_like("ova:q", (1, 256, 3 * 64)).cuda()
    a = ops.attention(q[..., :64], q[..., 64:128], q[..., 128:], 2, 32 ** -0.5, math=L.MATH_F16X3)
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    q2 = q.clone()
    q2[0, 100, 128 + 5] = 5.0e3                                         # one V entry beyond 65504 / 16
    ops.attention(q2[..., :64], q2[..., 64:128], q2[..., 128:], 2, 32 ** -0.5, math=L.MATH_F16X3)
    torch.cuda.synchronize()
    assert ops.read_status() == L.STATUS_F16X3_OVERFLOW
    b = ops.attention(q2[..., :64], q2[..., 64:128], q2[..., 128:], 2, 32 ** -0.5, math=L.MATH_FP32)
    torch.cuda.synchronize()
    assert ops.read_status() == 0 and torch.isfinite(b).all()


def test_rel2shape_reruns_an_overflowing_minibatch_in_fp32(tmp_path):
    """A checkpoint whose activations leave the fp16 range where the operand scale is still the constant 16: the first
    transformer block's to_v weights are scaled by 1e5, so its V (and the attention output feeding to_out) sit at ~1e5.
    (Until r4 this test put one residual-stream channel at 1e4 through conv_in's bias; the raw-stream consumers now take
    their scale from the tensor's magnitude bound and that checkpoint runs on F16X3 without a flag --
    test_unet_residual_stream_scaled_up_and_down_needs_no_fallback.)  The F16X3 run raises the flag; rel2shape re-runs
    the mini-batch on the fp32 kernels (policy 'fp32') and matches a pure-fp32 model bit for bit; policy 'raise'
    propagates CsOverflowError."""
    import warnings
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    from test_model_gpu import _scene
    m = _scene(tmp_path)
    sd = dict(m.Diff.df.state_dict())
    kv = next(k for k in sorted(sd) if k.endswith("attn1.to_v.weight"))
    sd[kv] = sd[kv] * 1.0e5                      # V of the first transformer block at ~1e5: outside 16 x fp16
    m.Diff.df.load_state_dict(sd)
    B = 3
    data = {"sdf": torch.zeros(B, 1), "rel": synth.gaussian_like("ovr:c", (B, 1, 1280)).cuda(),
            "uc": synth.gaussian_like("ovr:uc", (B, 1, 1280)).cuda()}
    x_T = synth.gaussian_like("ovr:xT", (1, 3, 16, 16, 16))
    kw = dict(ddim_steps=50, uc_scale=3.0, x_T=x_T, return_latents=True, max_steps=2)
    ops.read_status()
    m.Diff.df.set_math("f16x3")
    # r5: with the static bounds of the transformer-internal operands (the default) THIS checkpoint runs on F16X3 without a
    # flag -- test_transformer_operands_take_static_scales_and_need_no_fallback; the detect-and-re-run machinery is
    # exercised here with the feature off (CS_NO_STATIC_SCALES: the r4 behaviour)
    with L.debug_override(no_static_scales=1):
        m.Diff.overflow_policy = "raise"
        with pytest.raises(L.CsOverflowError):
            m.Diff.rel2shape(data, **kw)
        assert m.Diff.df.math == L.MATH_F16X3
        m.Diff.overflow_policy = "fp32"
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            sdf_a, lat_a = m.Diff.rel2shape(data, **kw)
        assert any("overflow" in str(w.message) for w in wlist)
        assert m.Diff.df.math == L.MATH_FP32                                 # stays on the fp32 kernels afterwards
        sdf_b, lat_b = m.Diff.rel2shape(data, **kw)                          # a pure fp32 run
        torch.cuda.synchronize()
        assert torch.isfinite(lat_a).all() and torch.equal(lat_a, lat_b) and torch.equal(sdf_a, sdf_b)
        assert ops.read_status() == 0


@pytest.mark.parametrize("scaled", ["to_v x1e5", "to_q,to_k x4", "ff.net.0 x300", "context x1e4", "none"])
def test_transformer_operands_take_static_scales_and_need_no_fallback(scaled):
    """the operands BORN INSIDE a transformer block (attention.py:237-245: q / k / v, the attention
    output, the GEGLU product, t2) used to ride the constant scale 16 + the overflow flag + a whole-mini-batch fp32 re-run.
    Their magnitude is bounded by the weights alone (cs_transformer_static_scales), so the scales are chosen per checkpoint
    and NO input can push them out of the fp16 range.  Checkpoints whose to_v (x 1e5: the r4 overflow recipe), to_q / to_k,
    GEGLU projection or context sit far outside PyTorch-default scales: one UNet forward of the reduced model on F16X3 --
    no flag -- within fp32 grade of the fp64 oracle on the same weights; both hosts agree bit for bit; with the feature
    off the same checkpoint raises the flag."""
    from commonscenes_amd import lib as L, ops, synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    from oracle import ref_torch as R
    from test_model_gpu import _unet_cfg
    cfg = _unet_cfg(True)
    sd = synth.synth_state_dict(unet_param_shapes(cfg))
    t0 = next(k for k in sorted(sd) if k.endswith("attn1.to_v.weight"))[:-len("attn1.to_v.weight")]
    ctx_mag = 1.0
    if scaled == "to_v x1e5":
        sd[t0 + "attn1.to_v.weight"] = sd[t0 + "attn1.to_v.weight"] * 1.0e5
    elif scaled == "to_q,to_k x4":          # (logits x16: still a soft softmax -- x900 makes the fp64 comparison itself ill-conditioned)
        for n in ("to_q", "to_k"):
            sd[t0 + f"attn1.{n}.weight"] = sd[t0 + f"attn1.{n}.weight"] * 4.0
    elif scaled == "ff.net.0 x300":
        for n in ("weight", "bias"):
            sd[t0 + f"ff.net.0.proj.{n}"] = sd[t0 + f"ff.net.0.proj.{n}"] * 300.0
    elif scaled == "context x1e4":
        ctx_mag = 1.0e4
    B = 2
    x = synth.gaussian_like("ss:x", (B, 3, 16, 16, 16))
    ctx = synth.gaussian_like("ss:ctx", (B, 1, 1280)) * ctx_mag
    t = torch.tensor([981, 21], dtype=torch.long)
    with torch.no_grad():
        ref = R.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())
    df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math("f16x3")
    df.load_state_dict(sd)
    ops.read_status()
    eps = df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
    torch.cuda.synchronize()
    flag = ops.read_status()
    e = rel_l2(eps, ref)
    blk = next(iter(df._tstat))
    ss = next(v for (k0, _, _), v in df._sscache.items() if k0 == blk)
    with L.debug_override(no_static_scales=1):
        df.reset_run_cache()
        old = df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
        torch.cuda.synchronize()
        fold = ops.read_status()
    df.reset_run_cache()
    print(f"static scales [{scaled}]: rel-L2 vs fp64 {e:.2e}, flag {flag}; first block scales attn {ss['attn']} a {ss['a']:g} "
          f"gg {ss['gg']:g} t2 {ss['t2']:g}, bounds {({k_: float(f'{v_:.3g}') for k_, v_ in ss['bounds'].items()})}; "
          f"constant 16: rel-L2 {rel_l2(old, ref):.2e}, flag {fold}")
    assert flag == 0
    assert e < 5e-6                                     # the UNet-forward gate of the goldens is 1e-5
    if scaled == "to_v x1e5":
        assert fold & L.STATUS_F16X3_OVERFLOW          # ... where the constant scale overflowed (the r4 recipe)
    elif fold == 0:
        assert rel_l2(eps, old) < 2e-6
    # the native driver applies the same rule (cs_unet.hip::attn_block): bit-identical
    nd = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device="cuda", math="f16x3")
    nd.load_state_dict(sd)
    en = nd(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    assert torch.equal(en, eps), f"native vs python: {rel_l2(en, eps):.2e}"


@pytest.mark.parametrize("mag", [1.0, 3.0e4, 1.0e-4])
def test_conv_in_operand_scale_follows_the_latent(mag):
    """r6: the UNet's conv_in reads the RAW latent x_t (openai_model_3d.py:752-766), which used to ride
    the constant operand scale 16 + the overflow flag (|x| >= 4094 -> fp32 re-run of the mini-batch, a 4.6x cliff).  Both
    hosts now leave its exact max |.| in a bound slot (cs_absmax -> CsConvGemm.a_bound): a latent at 3e4 (and one at 1e-4)
    runs on F16X3 without a flag, at fp32 grade of the fp64 oracle, the two hosts bit for bit; with the feature off
    (CS_NO_DYN_SCALE: the constant 16) the 3e4 latent raises the flag."""
    from commonscenes_amd import lib as L, ops, synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    from oracle import ref_torch as R
    from test_model_gpu import _unet_cfg
    cfg = _unet_cfg(True)
    sd = synth.synth_state_dict(unet_param_shapes(cfg))
    B = 2
    x = synth.gaussian_like("cin:x", (B, 3, 16, 16, 16)) * mag
    ctx = synth.gaussian_like("cin:ctx", (B, 1, 1280))
    t = torch.tensor([981, 21], dtype=torch.long)
    with torch.no_grad():
        ref = R.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())
        # the reference's own fp32 arithmetic against fp64 on this input: a 1e-4 latent makes every feature map nearly
        # constant in space and the forward ill-conditioned (2.5e-5 on the CPU oracle; 1.5e-6 at |x| ~ 1) -- the gate follows it
        e32 = rel_l2(R.unet_forward(sd, cfg, x, t, ctx), ref)
    df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math("f16x3")
    df.load_state_dict(sd)
    ops.read_status()
    eps = df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
    torch.cuda.synchronize()
    flag = ops.read_status()
    e = rel_l2(eps, ref)
    with L.debug_override(no_dyn_scale=1):
        df.reset_run_cache()
        df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
        torch.cuda.synchronize()
        fold = ops.read_status()
    df.reset_run_cache()
    print(f"conv_in operand at |x| ~ {mag:g}: rel-L2 vs fp64 {e:.2e} (fp32 oracle {e32:.2e}), flag {flag}; constant scale 16: flag {fold}")
    assert flag == 0 and e < max(5e-6, 2.0 * e32)
    if mag >= 1e4:
        assert fold & L.STATUS_F16X3_OVERFLOW
    nd = NativeDiffusionUNet(cfg, conditioning_key="crossattn", device="cuda", math="f16x3")
    nd.load_state_dict(sd)
    en = nd(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    assert torch.equal(en, eps), f"native vs python: {rel_l2(en, eps):.2e}"


@pytest.mark.parametrize("scaled", ["v x3e3", "q,k x4", "none"])
def test_vq_decoder_attention_takes_static_scales_and_needs_no_fallback(scaled):
    """r6: the VQ decoder's single-head 4096-token attention (vqvae_modules.py:154-178) rode the constant
    operand scale 16 + the overflow flag.  Its q / k / v are Conv1x1(Normalize(x)) + bias: bounded by the weights and the norm's
    affine parameters alone (cs_attnblock_static_scales), so NO input can push them -- or the attention output feeding proj_out
    -- out of the fp16 range.  A checkpoint whose v conv is x3e3 (|v| ~ 5e3 > 4094 = what the constant scale carries) decodes on
    F16X3 without a flag, at fp32 grade of the same model on the fp32-input MFMA kernels; both hosts bit for bit; with the
    feature off the same checkpoint raises the flag."""
    from commonscenes_amd import lib as L, ops, synth
    from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes
    from commonscenes_amd.vqvae_native import NativeVQVAE
    from oracle.ref_torch import VQ_FULL
    sd = synth.synth_state_dict(vqvae_param_shapes(VQ_FULL, 8192, 3), device="cuda")
    a = "decoder.mid.attn_1."
    if scaled == "v x3e3":
        # (proj_out / 3e3 keeps the residual stream where it was: this test is about the attention block's own operands -- the
        # decoder's RAW-stream consumers further down, nin_shortcut and the folded Upsample convs, still carry the constant scale)
        sd[a + "v.weight"] = sd[a + "v.weight"] * 3.0e3
        sd[a + "v.bias"] = sd[a + "v.bias"] * 3.0e3
        sd[a + "proj_out.weight"] = sd[a + "proj_out.weight"] / 3.0e3
    elif scaled == "q,k x4":
        for n in ("q", "k"):
            sd[a + f"{n}.weight"] = sd[a + f"{n}.weight"] * 4.0
    lat = synth.gaussian_like("vqa:lat", (2, 3, 16, 16, 16), scale=0.8).cuda()
    vq = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math("f16x3")
    vq.load_state_dict(sd)
    ops.read_status()
    out = vq.decode_no_quant(lat)
    torch.cuda.synchronize()
    flag = ops.read_status()
    with L.debug_override(no_static_scales=1):
        vq.decode_no_quant(lat)
        torch.cuda.synchronize()
        fold = ops.read_status()
    ref = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math("fp32")
    ref.load_state_dict(sd)
    want = ref.decode_no_quant(lat)
    torch.cuda.synchronize()
    e = rel_l2(out, want)
    print(f"VQ attention [{scaled}]: F16X3 (static scales) vs fp32-input MFMA rel-L2 {e:.2e}, flag {flag}; constant 16: flag {fold}")
    assert flag == 0 and e < 2e-5
    if scaled == "v x3e3":
        assert fold & L.STATUS_F16X3_OVERFLOW
    nv = NativeVQVAE(VQ_FULL, 8192, 3, device="cuda", math="f16x3")
    nv.load_state_dict(sd)
    on = nv.decode_no_quant(lat)
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    assert torch.equal(on, out), f"native vs python: {rel_l2(on, out):.2e}"


@pytest.mark.parametrize("scaled", ["v x1e4", "none"])
def test_concat_attentionblock_takes_static_scales_and_needs_no_fallback(scaled):
    """r6: the same for the concat family's AttentionBlock (openai_model_3d.py:316-366: GroupNorm -> Conv1d qkv -> attention ->
    proj_out): the v rows of the fused qkv weight x1e4 -- one forward of the reduced concat UNet on F16X3 without a flag, at
    fp32 grade of the fp64 oracle, both hosts bit for bit; with the feature off the flag is raised."""
    from commonscenes_amd import lib as L, ops, synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from commonscenes_amd.unet_native import NativeDiffusionUNet
    from oracle import ref_torch as R
    cfg = dict(R.UNET_CONCAT_SMALL)
    sd = synth.synth_state_dict(unet_param_shapes(cfg))
    kq = next(k for k in sorted(sd) if k.endswith(".qkv.weight"))
    if scaled == "v x1e4":
        w, b = sd[kq].clone(), sd[kq[:-6] + "bias"].clone()
        heads = cfg["num_heads"]
        ch = w.shape[0] // (3 * heads)
        for h in range(heads):                     # the legacy layout: [head][q | k | v][ch]
            w[h * 3 * ch + 2 * ch:(h + 1) * 3 * ch] *= 1.0e4
            b[h * 3 * ch + 2 * ch:(h + 1) * 3 * ch] *= 1.0e4
        sd[kq], sd[kq[:-6] + "bias"] = w, b
    B = 2
    x = synth.gaussian_like("cab:x", (B, 3, 16, 16, 16))
    cvol = synth.gaussian_like("cab:c", (B, 1, 16, 16, 16))
    t = torch.tensor([981, 21], dtype=torch.long)
    with torch.no_grad():
        ref = R.unet_forward({k: v.double() for k, v in sd.items()}, cfg, torch.cat([x, cvol], 1).double(), t, None)
    df = DiffusionUNet(cfg, conditioning_key="concat", device="cuda").set_math("f16x3")
    df.load_state_dict(sd)
    ops.read_status()
    eps = df(x.cuda(), t.cuda(), c_concat=[cvol.cuda()])
    torch.cuda.synchronize()
    flag = ops.read_status()
    e = rel_l2(eps, ref)
    with L.debug_override(no_static_scales=1):
        df(x.cuda(), t.cuda(), c_concat=[cvol.cuda()])
        torch.cuda.synchronize()
        fold = ops.read_status()
    print(f"concat AttentionBlock [{scaled}]: rel-L2 vs fp64 {e:.2e}, flag {flag}; constant 16: flag {fold}")
    assert flag == 0 and e < 5e-6
    if scaled == "v x1e4":
        assert fold & L.STATUS_F16X3_OVERFLOW
    nd = NativeDiffusionUNet(cfg, conditioning_key="concat", device="cuda", math="f16x3")
    nd.load_state_dict(sd)
    en = nd(x.cuda(), t.cuda(), c_concat=[cvol.cuda()])
    torch.cuda.synchronize()
    assert ops.read_status() == 0
    assert torch.equal(en, eps), f"native vs python: {rel_l2(en, eps):.2e}"


def test_native_vqvae_overflow_falls_back_to_fp32(tmp_path, monkeypatch):
    """with unet_driver='native' the decode runs on NativeVQVAE, whose F16X3 overflow fall-back needs
    set_math().  A decoder whose conv_in bias puts one channel at 1e4 overflows the first raw-activation consumer
    (nin_shortcut / Upsample conv); policy 'fp32' re-runs the decode on the fp32 kernels and equals the Python-sequenced
    fp32 decode bit for bit; policy 'raise' propagates CsOverflowError (not an AttributeError)."""
    import warnings
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    from test_model_gpu import _scene
    monkeypatch.setenv("CS_UNET_DRIVER", "native")
    m = _scene(tmp_path)
    assert type(m.Diff.vqvae).__name__ == "NativeVQVAE"
    vsd = {k: v.clone() for k, v in m.Diff.vqvae.state_dict().items()}
    vsd["decoder.conv_in.bias"][0] = 1.0e4
    m.Diff.vqvae.load_state_dict(vsd)
    B = 2
    data = {"sdf": torch.zeros(B, 1), "rel": synth.gaussian_like("nvo:c", (B, 1, 1280)).cuda(),
            "uc": synth.gaussian_like("nvo:uc", (B, 1, 1280)).cuda()}
    kw = dict(ddim_steps=50, uc_scale=3.0, x_T=synth.gaussian_like("nvo:xT", (1, 3, 16, 16, 16)), return_latents=True,
              max_steps=1)
    m.Diff.overflow_policy = "raise"
    with pytest.raises(L.CsOverflowError):
        m.Diff.rel2shape(data, **kw)
    assert m.Diff.vqvae.math == L.MATH_F16X3
    m.Diff.overflow_policy = "fp32"
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        sdf_a, lat_a = m.Diff.rel2shape(data, **kw)
    torch.cuda.synchronize()
    assert any("VQ-VAE decoder" in str(w.message) for w in wlist)
    assert m.Diff.vqvae.math == L.MATH_FP32 and m.Diff.df.math == L.MATH_F16X3 and torch.isfinite(sdf_a).all()
    monkeypatch.setenv("CS_UNET_DRIVER", "python")
    from commonscenes_amd.vqvae import VQVAE
    from oracle.ref_torch import VQ_FULL
    py = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math("fp32")
    py.load_state_dict(vsd)
    assert torch.equal(py.decode_no_quant(lat_a), sdf_a)
    assert ops.read_status() == 0


def test_a_stale_overflow_flag_is_not_attributed_to_the_next_run(tmp_path):
    """the status word is sticky per device; a bit left by an earlier, unchecked launch must not make the next
    rel2shape switch the UNet to the fp32 kernels (5x slower) with a misleading warning."""
    import warnings
    from commonscenes_amd import lib as L
    from commonscenes_amd import ops, synth
    from test_model_gpu import _scene
    m = _scene(tmp_path)
    B = 2
    data = {"sdf": torch.zeros(B, 1), "rel": synth.gaussian_like("st:c", (B, 1, 1280)).cuda(),
            "uc": synth.gaussian_like("st:uc", (B, 1, 1280)).cuda()}
    ops.status_word().fill_(L.STATUS_F16X3_OVERFLOW)                    # what an unchecked direct call would leave
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        sdf = m.Diff.rel2shape(data, ddim_steps=50, uc_scale=3.0, x_T=synth.gaussian_like("st:xT", (1, 3, 16, 16, 16)),
                               max_steps=1)
    torch.cuda.synchronize()
    assert not any("overflow" in str(w.message) for w in wlist)
    assert m.Diff.df.math == L.MATH_F16X3 and m.Diff.vqvae.math == L.MATH_F16X3 and torch.isfinite(sdf).all()


def test_scene_graph_index_errors_raise_like_the_reference(tmp_path):
    """an object id beyond the embedding table or a triple endpoint beyond the node count raised IndexError
    in the reference (nn.Embedding / tensor indexing); the HIP gather kernels skip the entry and set a device flag,
    which the GCN wrapper reads back once per call."""
    from commonscenes_amd import synth
    from test_model_gpu import _scene
    m = _scene(tmp_path)
    g = synth.random_scene_graph(6, seed=7)
    args = lambda objs, tri: (g["z"].cuda(), objs.cuda(), tri.cuda(), g["text_feats"].cuda(), g["rel_feats"].cuda())
    uc, c = m.encoder_2(*args(g["objs"], g["triples"]))
    assert torch.isfinite(c).all()
    bad_objs = g["objs"].clone()
    bad_objs[2] = 10_000                                                # vocabulary has 35 + 1 entries
    with pytest.raises(IndexError):
        m.encoder_2(*args(bad_objs, g["triples"]))
    bad_tri = g["triples"].clone()
    bad_tri[1, 2] = g["objs"].shape[0] + 3                              # object endpoint past the last node
    with pytest.raises(IndexError):
        m.encoder_2(*args(g["objs"], bad_tri))
    bad_pred = g["triples"].clone()
    bad_pred[0, 1] = 999                                                # predicate id past the table
    with pytest.raises(IndexError):
        m.encoder_2(*args(g["objs"], bad_pred))
    uc2, c2 = m.encoder_2(*args(g["objs"], g["triples"]))               # the flag does not stick to later calls
    assert torch.equal(c, c2) and torch.equal(uc, uc2)


# ----------------------------------------------------------------------------------------------------------------------
# C4's workload on one GPU
# ----------------------------------------------------------------------------------------------------------------------
def test_256_objects_properties_on_one_gpu(tmp_path):
    """BASELINE configs[3] is 256 objects sharded 8 x 32.  Its whole workload on ONE GPU through the product API
    (reduced-width UNet so it runs in seconds): a 258-node scene graph -> encoder_2 -> rel2shape (mini-batch 32) ->
    decode.  Properties (the oracle cannot run this size in seconds): every object finite; any 32-object shard computed
    on its own equals the same rows of the 256-object run bit for bit (what makes 8-way sharding exact);
    objects with identical conditioning get identical shapes (shared x_T)."""
    from commonscenes_amd import synth
    from test_model_gpu import _scene
    m = _scene(tmp_path)
    nobj = 256
    g = synth.random_scene_graph(nobj, seed=3)
    O = g["objs"].shape[0]
    assert O == nobj + 2
    uc, c = m.encoder_2(g["z"].cuda(), g["objs"].cuda(), g["triples"].cuda(), g["text_feats"].cuda(),
                        g["rel_feats"].cuda())
    assert uc.shape == (O, 1, 1280) and torch.isfinite(c).all() and torch.isfinite(uc).all()
    c, uc = c[:nobj].clone(), uc[:nobj].clone()
    c[255], uc[255] = c[0], uc[0]
    x_T = synth.gaussian_like("c4:xT", (1, 3, 16, 16, 16))
    data = {"sdf": torch.zeros(nobj, 1), "rel": c, "uc": uc}
    sdf, lat = m.Diff.rel2shape(data, ddim_steps=100, uc_scale=3.0, x_T=x_T, mini_B=32, return_latents=True, max_steps=2)
    torch.cuda.synchronize()
    assert sdf.shape == (nobj, 1, 64, 64, 64) and torch.isfinite(sdf).all() and torch.isfinite(lat).all()
    assert torch.equal(lat[0], lat[255]) and torch.equal(sdf[0], sdf[255])
    assert not torch.equal(lat[0], lat[1])
    for r in (0, 3, 7):                                                 # three of the eight rank shards
        sl = slice(32 * r, 32 * r + 32)
        s2, l2 = m.Diff.rel2shape({"sdf": torch.zeros(32, 1), "rel": c[sl], "uc": uc[sl]}, ddim_steps=100, uc_scale=3.0,
                                  x_T=x_T, mini_B=32, return_latents=True, max_steps=2)
        torch.cuda.synchronize()
        assert torch.equal(l2, lat[sl]) and torch.equal(s2, sdf[sl]), r


def test_c4_full_width_256_objects_on_one_gpu(tmp_path):
    """BASELINE configs[3] (C4) at its real size: the SHIPPED 413.5 M-parameter UNet, 256 objects (what eight ranks of 32
    hold together), a 258-node scene graph, the product API -- encoder_2 -> rel2shape (two DDIM steps of the 100-step
    schedule, mini-batch 32) -> decode to 64^3.  The oracle cannot run this size; the properties that make the 8 x 32
    sharding exact are checked instead: everything finite; any 32-object rank shard computed on its own equals the same
    rows of the 256-object run bit for bit; objects with identical conditioning get identical shapes (shared x_T)."""
    from commonscenes_amd import synth
    from test_model_gpu import _scene
    m = _scene(tmp_path, small=False)
    nobj = 256
    g = synth.random_scene_graph(nobj, seed=3)
    uc, c = m.encoder_2(g["z"].cuda(), g["objs"].cuda(), g["triples"].cuda(), g["text_feats"].cuda(),
                        g["rel_feats"].cuda())
    assert uc.shape == (nobj + 2, 1, 1280) and torch.isfinite(c).all() and torch.isfinite(uc).all()
    c, uc = c[:nobj].clone(), uc[:nobj].clone()
    c[255], uc[255] = c[0], uc[0]
    x_T = synth.gaussian_like("c4:xT", (1, 3, 16, 16, 16))
    kw = dict(ddim_steps=100, uc_scale=3.0, x_T=x_T, mini_B=32, return_latents=True, max_steps=2)
    sdf, lat = m.Diff.rel2shape({"sdf": torch.zeros(nobj, 1), "rel": c, "uc": uc}, **kw)
    torch.cuda.synchronize()
    assert sdf.shape == (nobj, 1, 64, 64, 64) and torch.isfinite(sdf).all() and torch.isfinite(lat).all()
    assert torch.equal(lat[0], lat[255]) and torch.equal(sdf[0], sdf[255])
    assert not torch.equal(lat[0], lat[1])
    for r in (0, 5):                                                    # two of the eight rank shards
        sl = slice(32 * r, 32 * r + 32)
        s2, l2 = m.Diff.rel2shape({"sdf": torch.zeros(32, 1), "rel": c[sl], "uc": uc[sl]}, **kw)
        torch.cuda.synchronize()
        assert torch.equal(l2, lat[sl]) and torch.equal(s2, sdf[sl]), r
    # the reference's own mini-batching (7) of the same objects: same shapes to fp32 summation-order noise
    # (launch_B=0: one sampler run per mini-batch, exactly the reference's loop)
    s7, l7 = m.Diff.rel2shape({"sdf": torch.zeros(9, 1), "rel": c[:9], "uc": uc[:9]}, **dict(kw, mini_B=7, launch_B=0))
    torch.cuda.synchronize()
    from conftest import rel_l2
    assert m.Diff.last_launch_sizes == [7, 2]
    assert rel_l2(l7, lat[:9]) < 1e-4
    # r4: the DEFAULT call (mini_B = 7, launch_B = 32) coalesces the reference's mini-batches into launches of
    # ceil(32 / 7) * 7 = 35 objects; every object must come out as its own 7-object mini-batch would have produced it
    # (<= 1e-5: only the GEMM tilings, i.e. fp32 summation orders, differ), and equal conditioning -> equal result
    kd = dict(ddim_steps=100, uc_scale=3.0, x_T=x_T, return_latents=True, max_steps=2)
    sl40 = {"sdf": torch.zeros(40, 1), "rel": c[:40], "uc": uc[:40]}
    assert m.Diff.mini_B == 7 and m.Diff.launch_B == 32
    sc, lc = m.Diff.rel2shape(sl40, **kd)
    assert m.Diff.last_launch_sizes == [21, 19]     # whole mini-batches dealt evenly over two launches (r5)
    sr, lr = m.Diff.rel2shape(sl40, **dict(kd, launch_B=0))
    torch.cuda.synchronize()
    assert m.Diff.last_launch_sizes == [7, 7, 7, 7, 7, 5]
    worst = max(rel_l2(lc[i], lr[i]) for i in range(40))
    print(f"[coalesced mini-batches] worst per-object latent rel-L2 vs the reference's own mini-batching: {worst:.2e}")
    assert worst < 1e-5, worst
    assert rel_l2(lc, lat[:40]) < 1e-5
    # eta > 0 draws one noise tensor per mini-batch and step (ddim.py:240): those calls are never coalesced
    m.Diff.rel2shape({"sdf": torch.zeros(9, 1), "rel": c[:9], "uc": uc[:9]}, ddim_steps=100, uc_scale=3.0, x_T=x_T,
                     ddim_eta=0.5, max_steps=1)
    assert m.Diff.last_launch_sizes == [7, 2]


def test_full_size_conv_paths_agree_at_the_benchmark_shapes():
    """BASELINE configs[2]'s own shapes (CFG batch 64), as size-independent properties: (1) the 16^3 x 224 -> 224 ResBlock
    conv through GroupNorm's pre-split operand pair and the slab kernel equals the per-tap gather kernel on fp32 GroupNorm
    output bit for bit (same accumulation order by construction); (2) the level-1 -> level-0 Upsample conv folded onto the
    source grid equals the direct 27-tap form to fp32 accumulation noise (each is 1-2e-7 from fp64 at small K, test_f16x3_gpu.py); (3) the 4^3-level conv's four K slices match the unsplit
    slab kernel to rounding, and a 32-object batch reproduces the first half of a 64-object one bit for bit."""
    import torch
    from commonscenes_amd import lib as L, ops, synth
    from conftest import rel_l2
    nb = 64
    # (1)
    x = synth.tensor_device("fs:x0", (nb, 16, 16, 16, 224), 1.0)
    g, b = synth.tensor_device("fs:g", (224,), 0.3) + 1.0, synth.tensor_device("fs:b", (224,), 0.1)
    w = synth.tensor_device("fs:w0", (224, 224, 3, 3, 3), (224 * 27) ** -0.5)
    pk = ops.pack_weight(w, synth.tensor_device("fs:c0", (224,), 0.1), math=L.MATH_F16X3)
    assert ops.wants_split16(nb * 4096, pk)
    h16 = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, split16=True)
    assert isinstance(h16, ops.Split16)
    h32 = ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU)
    slab = ops.conv_gemm(h16, pk, res=x)                  # auto: 256x224 tile, slab, pre-split operands
    gather = ops.conv_gemm(h32, pk, res=x, tile=3)        # 64x64 tiles: per-tap gather, in-loop operand split
    torch.cuda.synchronize()
    assert torch.isfinite(slab).all() and torch.equal(slab, gather)
    del h16, h32, slab, gather, x
    # (2)
    xs = synth.tensor_device("fs:x1", (nb, 16, 8, 8, 448), 1.0)
    w1 = synth.tensor_device("fs:w1", (448, 448, 3, 3, 3), (448 * 27) ** -0.5)
    b1 = synth.tensor_device("fs:c1", (448,), 0.1)
    folded = ops.conv_gemm(xs, ops.pack_weight(w1, b1, math=L.MATH_F16X3, fold_up=(0, 1, 1)), up=(0, 1, 1))
    direct = ops.conv_gemm(xs, ops.pack_weight(w1, b1, math=L.MATH_F16X3), up=(0, 1, 1))
    torch.cuda.synchronize()
    assert folded.shape == (nb, 16, 16, 16, 448)
    e = rel_l2(folded, direct)
    print(f"full-size Upsample conv, folded vs direct: rel-L2 {e:.2e}")
    assert e < 3e-6          # two fp32-accumulated sums over K = 12096 / 5376 terms in different orders
    del folded, direct, xs
    # (3)
    x2 = synth.tensor_device("fs:x2", (2 * nb, 16, 4, 4, 672), 1.0)
    pk2 = ops.pack_weight(synth.tensor_device("fs:w2", (672, 672, 3, 3, 3), (672 * 27) ** -0.5),
                          synth.tensor_device("fs:c2", (672,), 0.1), math=L.MATH_F16X3)
    whole = ops.conv_gemm(x2, pk2)
    half = ops.conv_gemm(x2[:nb], pk2)
    with L.debug_override(no_splitk=1):          # (the CsDebug view: reaches both hosts, restored on exit -- ADVICE r4)
        unsplit = ops.conv_gemm(x2[:nb], pk2)
    torch.cuda.synchronize()
    assert torch.equal(whole[:nb], half)
    assert rel_l2(half, unsplit) < 3e-6          # K = 18144 summed in four slices vs one chain


def test_benchmark_conv_instantiation_against_an_independent_fp64_evaluation_at_full_size():
    """the benchmarked instantiation conv_gemm_f16x3_kernel<1,7,8,1,true,32> (256x224 tile, A slab,
    GroupNorm-pre-split operands) was compared with the fp64 oracle on small geometries only; at the CFG-batch-64 shapes
    it was checked HIP-vs-HIP.  Here: the heaviest ResBlock conv of BASELINE configs[2] at its own size -- batch 64,
    672 -> 224 channels at 16^3 (M = 262 144 rows, K = 18 144) -- through GroupNorm(split16) + the slab kernel, and 4 096
    random output elements re-evaluated independently: GroupNorm + SiLU in fp64 by plain torch ops (no kernel of this
    package), the 27 x 672-term dot products in fp64 with numpy on the HOST."""
    import numpy as np
    from commonscenes_amd import lib as L, ops, synth
    nb, D, C, N = 64, 16, 672, 224
    x = synth.tensor_device("fsp:x", (nb, D, D, D, C), 1.0)
    x[:, :, :, :, :40] *= 3.0                                            # unequal group statistics
    g, b = synth.tensor_device("fsp:g", (C,), 0.3) + 1.0, synth.tensor_device("fsp:b", (C,), 0.1)
    w = synth.tensor_device("fsp:w", (N, C, 3, 3, 3), (C * 27) ** -0.5)
    bias = synth.tensor_device("fsp:c", (N,), 0.1)
    pk = ops.pack_weight(w, bias, math=L.MATH_F16X3)
    assert ops.wants_split16(nb * D ** 3, pk)
    prof = ops.GEMM_PROFILE = []
    try:
        out = ops.conv_gemm(ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU, split16=True), pk)
    finally:
        ops.GEMM_PROFILE = None
    torch.cuda.synchronize()
    assert (prof[0]["tile"], prof[0]["slab"], prof[0]["pre"]) == (4, 32, True)      # the benchmarked instantiation
    # independent evaluation: GroupNorm + SiLU in fp64 with torch ops, one sample at a time (memory)
    rs = np.random.RandomState(7)
    pick = np.stack([rs.randint(0, nb, 4096), rs.randint(0, D, 4096), rs.randint(0, D, 4096), rs.randint(0, D, 4096),
                     rs.randint(0, N, 4096)], 1)
    pick[:64, 1:4] = rs.choice([0, D - 1], size=(64, 3))                 # corners / faces: the zero-padded taps
    ref = np.zeros(4096)
    w64 = w.double().cpu().numpy()                                       # [N, C, 3, 3, 3]
    b64 = bias.double().cpu().numpy()
    for n in np.unique(pick[:, 0]):
        xs = x[n].double().reshape(D ** 3, 32, C // 32)
        mu = xs.mean(dim=(0, 2), keepdim=True)
        var = ((xs - mu) ** 2).mean(dim=(0, 2), keepdim=True)
        y = ((xs - mu) / torch.sqrt(var + 1e-5)).reshape(D, D, D, C) * g.double() + b.double()
        y = (y * torch.sigmoid(y))
        ypad = torch.zeros((D + 2, D + 2, D + 2, C), dtype=torch.float64, device=y.device)
        ypad[1:-1, 1:-1, 1:-1] = y
        for i in np.nonzero(pick[:, 0] == n)[0]:
            _, d_, h_, w_, co = pick[i]
            patch = ypad[d_:d_ + 3, h_:h_ + 3, w_:w_ + 3].cpu().numpy()          # [3, 3, 3, C] -> the host
            ref[i] = float(np.einsum("dhwc,cdhw->", patch, w64[co])) + b64[co]
    got = out[torch.from_numpy(pick[:, 0]).cuda(), torch.from_numpy(pick[:, 1]).cuda(), torch.from_numpy(pick[:, 2]).cuda(),
              torch.from_numpy(pick[:, 3]).cuda(), torch.from_numpy(pick[:, 4]).cuda()].double().cpu().numpy()
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    worst = float(np.max(np.abs(got - ref)) / np.sqrt(np.mean(ref ** 2)))
    # the same conv on the fp32-input MFMA kernel (bit-equal to an fp32 fma chain): what fp32 accumulation of K = 18 144
    # products of a non-centred operand (SiLU output: the partial sums drift) costs by itself
    sel = tuple(torch.from_numpy(pick[:, j]).cuda() for j in range(5))
    o32 = ops.conv_gemm(ops.groupnorm(x, g, b, 32, 1e-5, L.ACT_SILU), ops.pack_weight(w, bias, math=L.MATH_FP32))
    got32 = o32[sel].double().cpu().numpy()
    e32 = float(np.linalg.norm(got32 - ref) / np.linalg.norm(ref))
    print(f"full-size <1,7,8,1,true,32> vs independent fp64: rel-L2 {err:.2e} over 4096 elements (fp32 fma chain: "
          f"{e32:.2e}), worst |d|/rms {worst:.2e}")
    # r3 measurement: 1.44e-6 / worst 1.03e-5 -- fp32 accumulation noise of an 18 144-term chain, not the operand split
    assert err < 3e-6 and worst < 2.5e-5 and err < 3 * e32 + 3e-7


# ----------------------------------------------------------------------------------------------------------------------
# F16X3 range story beyond the PyTorch-default synthetic weights (VERDICT r2 weak #3 / next #7c)
# ----------------------------------------------------------------------------------------------------------------------
def test_f16x3_uniformly_tiny_operand_keeps_its_absolute_floor():
    """CS_MATH_F16X3 carries an activation a as the fp16 pair of 16 a, so operands with 16 |a| below fp16's normal range
    (|a| < 3.8e-6) keep an ABSOLUTE accuracy of 2^-29 per element instead of a relative one.  Mixed-scale tensors are
    covered by test_f16x3_wide_dynamic_range; here the WHOLE tensor sits at 1e-6 / 1e-7 / 1e-9: the error must stay under
    the documented floor (2^-29 * sum_k |w_k| per output element) -- harmless wherever the result joins an O(1)
    residual stream, which is every raw-activation consumer of the UNet / decoder -- and its size relative to the tiny
    result is REPORTED.  GroupNorm / LayerNorm-fed GEMMs never see such tensors: the norm rescales its input."""
    from commonscenes_amd import lib as L, ops, synth
    m, k, n = 512, 512, 224
    w = synth.gaussian_like("tiny:w", (n, k), scale=k ** -0.5)
    pk = ops.pack_weight(w.cuda(), math=L.MATH_F16X3)
    for mag in (1e-6, 1e-7, 1e-9):
        x = synth.gaussian_like(f"tiny:x{mag}", (m, k)) * mag
        ref = x.double() @ w.double().t()
        out = ops.linear(x.cuda(), pk).cpu().double()
        torch.cuda.synchronize()
        floor = 2.0 ** -29 * w.double().abs().sum(1)                     # per output column
        worst = float(((out - ref).abs() / floor).max())
        rel = float((out - ref).norm() / ref.norm())
        print(f"uniformly {mag:g}: max |err| = {worst:.2f} x the 2^-29 floor, rel-L2 of the tiny result {rel:.2e}")
        assert worst <= 1.0 + 1e-3 and ops.read_status() == 0
    # the same tensor through a GroupNorm first (what every 3x3x3 conv of a ResBlock sees): scale-free, fp32-grade
    x5 = synth.gaussian_like("tiny:x5", (2, 4, 8, 8, 64))
    g, b = torch.ones(64).cuda(), torch.zeros(64).cuda()
    w3 = synth.gaussian_like("tiny:w3", (224, 64, 3, 3, 3), scale=(64 * 27) ** -0.5).cuda()
    pk3 = ops.pack_weight(w3, None, math=L.MATH_F16X3)
    big = ops.conv_gemm(ops.groupnorm(x5.cuda(), g, b, 32, 0.0, L.ACT_NONE), pk3)
    small = ops.conv_gemm(ops.groupnorm((x5 * 2.0 ** -20).cuda(), g, b, 32, 0.0, L.ACT_NONE), pk3)
    torch.cuda.synchronize()
    assert torch.equal(big, small)                                       # a power-of-two input scale changes nothing


def test_norm_fed_gemms_take_their_operand_scale_from_the_producer():
    """VERDICT r2 next #7c: a_scale was the constant 16 everywhere.  A GroupNorm / LayerNorm bounds its output --
    |y| <= max|gamma| sqrt(n - 1) + max|beta| -- so the GEMM it feeds takes the largest power of two that keeps that bound
    inside the fp16 range (ops.norm_a_scale): overflow is impossible there whatever the input, and a layer whose affine
    parameters are uniformly tiny (gamma = 1e-6) keeps RELATIVE fp32-grade accuracy, which the fixed scale loses."""
    from commonscenes_amd import lib as L, ops, synth
    from oracle import ref_ops as R
    assert ops.norm_a_scale(1.2, 0.1, 28672) == 256.0 and ops.norm_a_scale(1.2, 0.1, 448) == 2048.0
    for g_, b_, n in ((1.2, 0.1, 28672), (37.0, 5.0, 172032), (1e-6, 1e-7, 1024), (3e4, 0.0, 7), (0.0, 0.0, 9)):
        s_ = ops.norm_a_scale(g_, b_, n)
        assert s_ == 2.0 ** round(__import__("math").log2(s_)) and 2.0 ** -8 <= s_ <= 2.0 ** 40
        if s_ > 2.0 ** -8:
            assert (g_ * (n - 1) ** 0.5 + b_) * s_ <= 65000.0
    nb, d, c, co = 2, 8, 64, 224
    x = synth.gaussian_like("ns:x", (nb, d, d, d, c))
    x[0, 0, 0, 0, :2] = 4000.0                                     # an outlier GroupNorm turns into x^ ~ 30
    gam = (synth.gaussian_like("ns:g", (c,)) * 0.1 + 1.0) * 1e-6
    bet = synth.gaussian_like("ns:b", (c,)) * 1e-7
    w = synth.gaussian_like("ns:w", (co, c, 3, 3, 3), scale=(c * 27) ** -0.5)
    ref = R.conv_ndhwc(R.groupnorm_ndhwc(x.double(), gam.double(), bet.double(), 32, 1e-5, "silu"), w.double(), None)
    pk = ops.pack_weight(w.cuda(), None, math=L.MATH_F16X3)
    s = ops.norm_a_scale(float(gam.abs().max()), float(bet.abs().max()), d ** 3 * (c // 32))
    assert s >= 2.0 ** 28          # (1.3e-6 * 32 + 3e-7) * 2^30 = 45000
    ops.read_status()
    errs = {}
    for name, sc in (("derived", s), ("fixed16", None)):
        for split in (False, True):
            hn = ops.groupnorm(x.cuda(), gam.cuda(), bet.cuda(), 32, 1e-5, L.ACT_SILU, split16=split, a_scale=sc)
            out = ops.conv_gemm(hn, pk, a_scale=sc)
            torch.cuda.synchronize()
            errs[(name, split)] = rel_l2(out, ref)
    print("tiny-gamma GroupNorm -> conv, rel-L2 vs fp64:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert ops.read_status() == 0
    assert errs[("derived", False)] < 1e-6 and errs[("derived", True)] < 1e-6
    assert errs[("fixed16", False)] > 10 * errs[("derived", False)]          # what the constant scale gave up


def test_unet_with_heavy_tailed_weights_and_latent_outliers():
    """the reduced-width UNet with Student-t (3 degrees of freedom) conv / linear weights -- the heavy-tailed shape trained
    networks have, which the PyTorch-default uniform synthetic weights lack -- and a latent with +-30 outliers, against the
    fp64 oracle on the same weights.  Activations reach the hundreds to thousands here; F16X3 must stay fp32-grade while
    they are below its range (|a| < 4094) and say so when they are not (the overflow flag; then the fp32 path is gated)."""
    import numpy as np
    from commonscenes_amd import lib as L, ops, synth
    from commonscenes_amd.unet import DiffusionUNet, unet_param_shapes
    from oracle import ref_torch as R
    from test_model_gpu import _unet_cfg
    cfg = _unet_cfg(True)
    shapes = unet_param_shapes(cfg)
    sd = synth.synth_state_dict(shapes)
    rs = np.random.RandomState(11)
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            t3 = rs.standard_t(3, size=tuple(v.shape)) / np.sqrt(3.0)                         # unit variance, heavy tails
            sd[k] = torch.from_numpy((t3 * fan_in ** -0.5).astype(np.float32))
    B = 2
    x = synth.gaussian_like("ht:x", (B, 3, 16, 16, 16))
    x[0, 1, 3, 4, 5], x[1, 2, 9, 9, 9] = 30.0, -30.0
    ctx = synth.gaussian_like("ht:ctx", (B, 1, 1280))
    t = torch.tensor([981, 21], dtype=torch.long)
    with torch.no_grad():
        ref = R.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())
    df = DiffusionUNet(cfg, conditioning_key="crossattn", device="cuda").set_math("f16x3")
    df.load_state_dict(sd)
    df.trace = {}
    ops.read_status()
    eps = df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()])
    torch.cuda.synchronize()
    amax = max(float(v.abs().max()) for v in df.trace.values())
    flagged = bool(ops.read_status() & L.STATUS_F16X3_OVERFLOW)
    e16 = rel_l2(eps, ref)
    df.set_math("fp32")
    e32 = rel_l2(df(x.cuda(), t.cuda(), c_crossattn=[ctx.cuda()]), ref)
    print(f"heavy-tailed UNet: largest block activation {amax:.0f}, F16X3 rel-L2 vs fp64 {e16:.2e} (fp32 MFMA {e32:.2e}), "
          f"overflow flag {flagged}")
    assert e32 < 2e-5
    if flagged:
        assert amax > 1000.0            # the flag is raised only by genuinely large activations ...
    else:
        assert e16 < 2e-5 and e16 < 4 * e32 + 2e-6      # ... and without it F16X3 is as good as the fp32 chain

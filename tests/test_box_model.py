"""BASELINE configs[0] (`v2_box`: layout-only GCN + box-VAE on one synthetic 8-node scene graph).
CPU: the oracle restatement against the golden generated from the reference model/VAEGAN_V2BOX.py.
GPU: commonscenes_amd.scene_box.Sg2ScVAEModel (HIP kernels) against the same golden -- encoder, decoder,
manipulate, decoder_with_changes / decoder_with_additions / sampleBoxes with numpy's RNG seeded like the generator."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2


def _g():
    p = GOLDEN / "box_small.npz"
    if not p.exists():
        pytest.skip("box_small.npz not generated")
    return {k: v for k, v in np.load(p).items()}


def _t(g, k):
    return torch.from_numpy(np.ascontiguousarray(g[k]))


VOCAB = dict(object_idx_to_name=[f"obj{i}\n" for i in range(35)], pred_idx_to_name=[f"pred{i}\n" for i in range(16)])


def test_oracle_box_model_matches_reference():
    from commonscenes_amd import synth
    from commonscenes_amd.scene_box import box_param_shapes
    from oracle import ref_torch as R
    g = _g()
    sd = synth.synth_state_dict(box_param_shapes(35, 16))
    a = (_t(g, "objs"), _t(g, "triples"))
    tf, rf = _t(g, "text_feats"), _t(g, "rel_feats")
    with torch.no_grad():
        mu, logvar = R.box_encoder(sd, *a, _t(g, "boxes_gt"), tf, rf, _t(g, "angles_gt"))
        d3, ang = R.box_decoder(sd, _t(g, "z"), *a, tf, rf)
        chg = synth.gaussian_like("box:chg", (a[0].shape[0], 64))
        man = R.box_manipulate(sd, torch.cat([_t(g, "z"), chg], dim=1), *a, tf, rf)
        np.random.seed(1234)
        (d3c, angc), keep = R.box_decoder_with_changes(sd, _t(g, "z_in"), *a, tf, rf, [2], [4])
    assert mu.shape == (8, 64) and logvar.shape == (8, 64)
    for mine, key in ((mu, "mu"), (logvar, "logvar"), (d3, "d3"), (ang, "angles"), (man, "man"), (d3c, "d3_changes"),
                      (angc, "angles_changes")):
        # 3e-6: the golden was produced on another host; fp32 BLAS summation order differs between CPUs (measured
        # 1.4e-6 on the MI355X box's host for the 5-layer, K=1920 GCN stack, 4e-7 in the build container)
        assert rel_l2(mine, _t(g, key)) < 3e-6, key
    assert torch.equal(keep, _t(g, "keep_changes"))
    assert float(_t(g, "d3").pow(2).mean().sqrt()) > 0.01          # non-vacuous


def test_box_param_table_and_config_guard():
    from commonscenes_amd.scene_box import Sg2ScVAEModel, box_param_shapes
    S = box_param_shapes(35, 16)
    assert S["gconv_net_manipulation.gconvs.4.net2.3.weight"] == (64, 256)
    assert S["gconv_net_manipulation.gconvs.3.net2.3.weight"] == (704, 256)
    assert S["d3_embeddings.weight"] == (48, 6) and S["angle_embeddings.weight"] == (24, 16)
    with pytest.raises(NotImplementedError):
        Sg2ScVAEModel(VOCAB, embedding_dim=64, decoder_cat=False, mlp_normalization="batch", device="cpu")


@pytest.mark.gpu
def test_hip_box_model_vs_reference_golden():
    from commonscenes_amd import synth
    from commonscenes_amd.scene_box import Sg2ScVAEModel, box_param_shapes
    g = _g()
    m = Sg2ScVAEModel(VOCAB, embedding_dim=64, decoder_cat=True, mlp_normalization="batch", input_dim=6,
                      replace_latent=True, use_angles=True, residual=True, gconv_pooling="avg", gconv_num_layers=5)
    m.load_state_dict(synth.synth_state_dict(box_param_shapes(35, 16), device="cuda"))
    a = (_t(g, "objs"), _t(g, "triples"))
    tf, rf = _t(g, "text_feats"), _t(g, "rel_feats")
    mu, logvar = m.encoder(*a, _t(g, "boxes_gt"), None, tf, rf, _t(g, "angles_gt"))
    d3, ang = m.decoder(_t(g, "z"), *a, tf, rf, None)
    chg = synth.gaussian_like("box:chg", (a[0].shape[0], 64))
    man = m.manipulate(torch.cat([_t(g, "z"), chg], dim=1), *a, tf, rf, None)
    np.random.seed(1234)
    (d3c, angc), keepc = m.decoder_with_changes(_t(g, "z_in"), *a, tf, rf, None, [2], [4])
    np.random.seed(99)
    (d3a, anga), keepa = m.decoder_with_additions(_t(g, "z_in"), *a, tf, rf, None, [2], [4],
                                                  distribution=(np.zeros(64), np.eye(64)))
    np.random.seed(5)
    d3s, angs = m.sampleBoxes(np.zeros(64), np.eye(64), *a, tf, rf, None)
    torch.cuda.synchronize()
    for mine, key in ((mu, "mu"), (logvar, "logvar"), (d3, "d3"), (ang, "angles"), (man, "man"),
                      (d3c, "d3_changes"), (angc, "angles_changes"), (d3a, "d3_add"), (anga, "angles_add"),
                      (d3s, "d3_sample"), (angs, "angles_sample")):
        assert mine.is_cuda and rel_l2(mine, _t(g, key)) < 3e-6, key
    assert torch.equal(keepc.cpu(), _t(g, "keep_changes")) and torch.equal(keepa.cpu(), _t(g, "keep_add"))

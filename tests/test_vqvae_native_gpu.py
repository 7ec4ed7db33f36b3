"""cs_vqvae_decode (native whole-decode driver) on the MI355X: reference golden, bit-equality with the Python
sequencer (commonscenes_amd.vqvae.VQVAE) in both math modes, workspace accounting."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def _pair(math):
    from commonscenes_amd import synth
    from commonscenes_amd.vqvae import VQVAE, vqvae_param_shapes
    from commonscenes_amd.vqvae_native import NativeVQVAE
    from oracle.ref_torch import VQ_FULL
    sd = synth.synth_state_dict(vqvae_param_shapes(VQ_FULL, 8192, 3), device="cuda")
    py = VQVAE(VQ_FULL, 8192, 3, device="cuda").set_math(math)
    py.load_state_dict(sd)
    nat = NativeVQVAE(VQ_FULL, 8192, 3, device="cuda", math=math)
    assert list(nat.shapes.items()) == list(vqvae_param_shapes(VQ_FULL, 8192, 3).items())
    nat.load_state_dict(sd)
    return py, nat


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_native_vq_decode_vs_golden_and_python_driver(math):
    p = GOLDEN / "vq_decode.npz"
    if not p.exists():
        pytest.skip("vq_decode.npz not generated")
    g = {k: v for k, v in np.load(p).items()}
    py, nat = _pair(math)
    lat = torch.from_numpy(g["latent"]).cuda()
    dec = nat.decode_no_quant(lat)
    torch.cuda.synchronize()
    assert dec.shape == (1, 1, 64, 64, 64)
    assert np.array_equal(nat.last_indices.cpu().numpy().reshape(-1), g["indices"].reshape(-1))
    assert rel_l2(dec, torch.from_numpy(g["dec"])) < 1e-4
    assert torch.equal(dec, py.decode_no_quant(lat))
    q = torch.from_numpy(g["quant"]).cuda()
    dq = nat.decode(q)          # the reference's straight-through `quant` = z + (z_q - z): rounds differently from z_q
    assert torch.equal(dq, py.decode(q)) and rel_l2(dq, dec) < 1e-5
    # a batch: objects decode independently of their neighbours
    from commonscenes_amd import synth
    lat3 = torch.cat([lat, synth.gaussian_like("nv:l", (2, 3, 16, 16, 16), scale=0.8).cuda()])
    d3 = nat.decode_no_quant(lat3)
    torch.cuda.synchronize()
    assert torch.equal(d3, py.decode_no_quant(lat3)) and torch.equal(d3[0], dec[0])


def test_native_vq_workspace_checked():
    from commonscenes_amd import lib as L
    _, nat = _pair("f16x3")
    lib = L.load()
    need = int(lib.cs_vqvae_workspace_bytes(nat._h, 1))
    assert need > 0
    lat = torch.zeros(1, 3, 16, 16, 16, device="cuda")
    out = torch.empty(1, 1, 64, 64, 64, device="cuda")
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    args = (nat._h, nat._arena.data_ptr(), lat.data_ptr(), out.data_ptr(), None, 1, 1, None, ws.data_ptr())
    assert lib.cs_vqvae_decode(*args, need - 4096, s) == L.CS_ENOMEM
    assert lib.cs_vqvae_decode(*args, need, s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_decode_of_a_32_object_batch_is_sliced():
    """BASELINE configs[2] decodes 32 objects: a 64^3 x 128-channel activation of 32 objects is 4.3 GB, past the 4 GiB
    buffer-descriptor window, so both sequencers decode in slices of 16 -- per object identical to a small batch."""
    from commonscenes_amd import synth
    py, nat = _pair("f16x3")
    lat = synth.gaussian_like("big:l", (32, 3, 16, 16, 16), scale=0.8).cuda()
    a = py.decode_no_quant(lat)
    b = nat.decode_no_quant(lat)
    torch.cuda.synchronize()
    assert a.shape == (32, 1, 64, 64, 64) and torch.isfinite(a).all()
    assert torch.equal(a, b) and torch.equal(py.last_indices, nat.last_indices)
    one = py.decode_no_quant(lat[21:22])
    assert torch.equal(one[0], a[21])


def test_load_vqvae_from_checkpoint_file_and_quantize(tmp_path):
    """model_utils.load_vqvae's job (model/model_utils.py:7-31): build from the YAML-shaped config, read a checkpoint
    file (raw state_dict or {'vqvae': ...}, encoder entries ignored); VQVAE.quantize returns the golden's indices."""
    from commonscenes_amd import synth
    from commonscenes_amd.vqvae import load_vqvae, vqvae_param_shapes
    from oracle.ref_torch import VQ_FULL
    g = {k: v for k, v in np.load(GOLDEN / "vq_decode.npz").items()}
    sd = dict(synth.synth_state_dict(vqvae_param_shapes(VQ_FULL, 8192, 3)))
    sd["encoder.conv_in.weight"] = torch.zeros(64, 1, 3, 3, 3)          # decode side must ignore the encoder
    conf = dict(model=dict(params=dict(embed_dim=3, n_embed=8192, ddconfig=dict(
        double_z=False, z_channels=3, resolution=64, in_channels=1, out_ch=1, ch=64, ch_mult=[1, 2, 4],
        num_res_blocks=1, attn_resolutions=[], dropout=0.0))))
    py, _ = _pair("f16x3")
    lat = torch.from_numpy(g["latent"]).cuda()
    want = py.decode_no_quant(lat)
    for payload, name in ((sd, "raw.pth"), ({"vqvae": sd}, "wrapped.pth")):
        torch.save(payload, tmp_path / name)
        vq = load_vqvae(conf, str(tmp_path / name), device="cuda").set_math("f16x3")   # same mode as `py`, whatever
        assert torch.equal(vq.decode_no_quant(lat), want)                                # CS_MATH made the default
    quant, idx = vq.quantize(lat)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy().reshape(-1), g["indices"].reshape(-1))
    assert quant.shape == lat.shape and torch.equal(vq.decode(quant), want)

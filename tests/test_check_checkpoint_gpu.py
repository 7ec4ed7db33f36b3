"""r5: tools/check_checkpoint.py -- the operator's one-command range / numerics check of a CommonScenes checkpoint on the
MI355X path (the reference saves 'df' next to the scene tensors and 'vqvae': VAEGAN_V2FULL.py:687-699; the UNet is loaded
from it at sdfusion_txt2shape_model.py:246-248).  Run as the operator runs it: a subprocess on a checkpoint FILE in the
reference's layout, at a reduced width so the check takes seconds."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
WIDTH = 64


def _run(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_checkpoint.py"), *args], capture_output=True, text=True,
                       env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("CHECK_CHECKPOINT ")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return r.returncode, json.loads(lines[-1][len("CHECK_CHECKPOINT "):]), r.stdout


def _state_dict():
    from commonscenes_amd import configs as K, synth
    from commonscenes_amd.unet import unet_param_shapes
    cfg = K.reduced(K.UNET_CROSSATTN, WIDTH)
    return {k: v.cpu() for k, v in synth.synth_state_dict(unet_param_shapes(cfg), device="cuda").items()}


def test_a_checkpoint_file_in_the_reference_layout_is_reported_ok(tmp_path):
    sd = _state_dict()
    path = tmp_path / "model100.pth"
    torch.save({"df": sd, "vqvae": {}, "epoch": 100}, path)
    rc, rep, out = _run(str(path), "--width", str(WIDTH), "--steps", "4", "--objects", "2")
    assert rc == 0 and rep["ok"] and rep["overflow_flag"] == 0 and rep["finite"]
    assert rep["source"] == str(path) and len(rep["trajectory_rel_l2_vs_fp32"]) == 4
    assert max(rep["trajectory_rel_l2_vs_fp32"]) < 1e-4
    # every transformer block is listed with its static bounds and power-of-two operand scales
    assert len(rep["transformer_blocks"]) == 11
    for b in rep["transformer_blocks"]:
        for s in (*b["scales"]["attn"], b["scales"]["a"], b["scales"]["gg"], b["scales"]["t2"]):
            assert s > 0 and abs(s) == 2.0 ** round(__import__("math").log2(s))
        assert all(v > 0 for v in b["bounds"].values())
    assert len(rep["norms"]) > 40 and "NONE raised" in out


def test_a_stressed_checkpoint_needs_no_fallback_and_a_broken_one_is_rejected(tmp_path):
    sd = _state_dict()
    path = tmp_path / "bare_state_dict.pth"
    torch.save(sd, path)                                     # a bare 'df' state_dict is accepted too
    # r4's overflow recipe (to_v x 1e5): static bounds move the operand scales, no flag, still inside the gate
    rc, rep, _ = _run(str(path), "--width", str(WIDTH), "--steps", "3", "--scale", "attn1.to_v.weight=1e5")
    assert rc == 0 and rep["ok"] and rep["overflow_flag"] == 0
    vmax = max(b["bounds"]["v"] for b in rep["transformer_blocks"])
    assert vmax > 1e4                                        # the report shows the stressed operand's bound
    # a checkpoint with a non-finite weight is NOT ok (exit code 1)
    bad = dict(sd)
    k = next(k for k in bad if k.endswith("input_blocks.1.0.in_layers.2.weight"))
    bad[k] = bad[k].clone()
    bad[k].view(-1)[0] = float("nan")
    torch.save({"df": bad}, tmp_path / "bad.pth")
    rc, rep, _ = _run(str(tmp_path / "bad.pth"), "--width", str(WIDTH), "--steps", "2")
    assert rc == 1 and not rep["ok"]

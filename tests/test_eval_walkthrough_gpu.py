"""BASELINE configs[4] (C5): tools/eval_walkthrough.py walks scripts/eval_3dfront.py:484-722's call sequence
(VAE facade -> sample_box_and_shape(gen_shape=True) -> sdf_to_mesh -> sample_points -> chamfer diversity) on a synthetic
dataset through the HIP path, single rank and object-sharded over two ranks, with the default attention and with the
opt-in fp16 MFMA attention C5 names."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run(extra, nproc=1, port=29571, base=("--scenes", "2", "--samples", "3", "--points", "2000"), timeout=900):
    env = dict(os.environ, CS_ONE_DEVICE="1", CS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [str(ROOT / "tools" / "eval_walkthrough.py"), *base] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("EVAL_WALKTHROUGH ")]
    assert len(lines) == 1
    return json.loads(lines[0].split(" ", 1)[1])


def _check(d, world):
    assert d["world"] == world and len(d["scenes"]) == 2
    for s, nobj in zip(d["scenes"], (4, 5)):
        assert s["finite"] and s["shapes"] == nobj and s["nodes"] == nobj + 2     # floor and _scene_ get no shape
        assert len(s["verts"]) == nobj and all(v > 0 for v in s["verts"])          # every object meshes to something
        assert -180 <= s["angle_range"][0] <= s["angle_range"][1] <= 180
    assert d["chamfer_diversity_n"] == 9 and d["chamfer_diversity_mean"] > 0       # fresh noise per call: shapes differ
    assert d["box_std_mean"] > 0


def test_eval_walkthrough_single_rank():
    _check(_run([]), 1)


def test_eval_walkthrough_fp16_attention_two_ranks_sharded():
    _check(_run(["--attention", "f16"], nproc=2), 2)


def test_eval_walkthrough_c5_at_full_size():
    """BASELINE configs[4] (C5) at its real size (scripts/eval_3dfront.py:484-722, VAEGAN_V2FULL.py:600-618): the shipped
    UNet (width 224, 413.5 M parameters), the full 100-step DDIM schedule, one SG-FRONT-livingroom-sized scene -- 32
    shaped objects, 34 nodes, >= 100 triples -- through VAE.sample_box_and_shape(gen_shape=True) -> sdf_to_mesh ->
    chamfer diversity, on the default (fp32-grade) attention and, C5's own option, the fp16 MFMA attention; the latent /
    SDF deviation between the two from identical z / x_T is REPORTED (report-only), with a loose sanity
    bound.  No oracle can run this size: finite, every object meshes, diversity > 0, timings printed."""
    base = ("--scenes", "1", "--samples", "2", "--points", "2000", "--width", "224", "--ddim-steps", "100",
            "--objects", "32", "--mini-b", "32")
    d = _run(["--compare-attention"], base=base, timeout=1500)
    assert d["world"] == 1 and d["width"] == 224 and d["ddim_steps"] == 100 and len(d["scenes"]) == 1
    s = d["scenes"][0]
    assert s["finite"] and s["shapes"] == 32 and s["nodes"] == 34 and s["triples"] >= 100
    assert len(s["verts"]) == 32 and all(v > 0 for v in s["verts"])
    assert d["chamfer_diversity_n"] == 32 and d["chamfer_diversity_mean"] > 0 and d["box_std_mean"] > 0
    cmp_ = d["attention_compare"]
    print(f"C5 full size: sample() {s['sample_s']:.2f} s for 32 objects x 100 steps + decode + mesh; "
          f"fp16-attention vs default: latent rel-L2 {cmp_['latent_rel_l2']:.2e} (worst object "
          f"{cmp_['latent_rel_l2_worst_object']:.2e}), SDF rel-L2 {cmp_['sdf_rel_l2']:.2e}, "
          f"{cmp_['sdf_sign_flips']} of {cmp_['sdf_voxels']} voxels change side at level 0.02; "
          f"sample() {cmp_['sample_s_default']:.2f} s default vs {cmp_['sample_s_f16']:.2f} s fp16 attention")
    assert cmp_["objects"] == 32 and cmp_["latent_rel_l2"] < 0.2       # report-only mode: sanity bound, not a parity gate
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "c5_full_size.json").write_text(json.dumps(d, indent=1))

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


def _run(extra, nproc=1, port=29571):
    env = dict(os.environ, CS_ONE_DEVICE="1", CS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [str(ROOT / "tools" / "eval_walkthrough.py"), "--scenes", "2", "--samples", "3", "--points", "2000"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
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

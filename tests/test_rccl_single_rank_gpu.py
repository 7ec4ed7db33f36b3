"""every multi-rank test on a one-GPU box takes the gloo host hop, so dist.py's device-to-device
RCCL lines (all_gather_into_tensor / all_reduce / broadcast on device tensors) had never executed.  A one-rank
`backend="nccl"` group + the test-only CS_DIST_FORCE_COLLECTIVES switch runs exactly those lines on the MI355X, so the
first 8-GPU run does not discover a dtype / contiguity error (tests/_rccl_single_rank_worker.py does the work)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_rccl_collectives_on_device_tensors_in_a_one_rank_group(tmp_path):
    out = tmp_path / "rccl.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CS_RCCL_OUT=str(out), CS_RCCL_PORT="29571")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "_rccl_single_rank_worker.py")], capture_output=True,
                       text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["backend"] == "nccl" and res["world"] == 1 and res["forced"]
    assert res["bcast_equal"] and res["bcast_on_device"] and res["bcast_volume_equal"]
    assert res["gather_equal"] and res["gather_noncontig_equal"] and res["gather_latents_equal"]
    assert res["gather_b0_shape"] == [0, 1, 64, 64, 64]
    assert res["any_failed_false"] is False and res["any_failed_true"] is True
    assert res["flags"] == [0.0, 1.0]
    assert res["rel2shape_equal"] and res["rel2shape_shape"] == [3, 1, 64, 64, 64]
    # r6: the empty-shard padded all-gather, and bench.py through the `nccl` + `device_id=` init path
    assert res["gather_empty_local_shape"] == [3, 1, 8, 8, 8] and res["gather_empty_local_zero"]
    assert res["bench_rc"] == 0, res["bench_tail"]
    assert res["bench_group"].startswith("forced one-rank group") and "rccl" in res["bench_backend"] and res["bench_finite"]


def test_force_switch_is_inert_without_a_process_group():
    """CPU: the switch only matters inside an initialised group; the product default never sets it."""
    from commonscenes_amd import dist as D
    os.environ["CS_DIST_FORCE_COLLECTIVES"] = "1"
    try:
        assert D._force() is False
        import torch
        t = torch.zeros(2, 3)
        assert D.all_gather_objects(t, 2) is t and D.any_rank_failed(True, "cpu") is True
    finally:
        del os.environ["CS_DIST_FORCE_COLLECTIVES"]

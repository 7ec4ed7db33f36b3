"""bench.py's multi-rank control flow (rank arithmetic, packed-conditioning broadcast, barrier + max-over-ranks timing,
rank-0-only JSON) run for real: two ranks launched by torch.distributed.run.  A 1-GPU box cannot host two RCCL ranks,
so the test hooks put both ranks on cuda:0 and use the gloo backend; the collectives and everything around them are the
code the driver's `--gpus N` run executes over RCCL."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, CS_BENCH_ONE_DEVICE="1", CS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(ROOT / "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--objects", "4"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["finite"] is True and d["value"] > 0
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["config"]["objects_per_gpu"] == 4 and "x2" in d["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_spawns_two_ranks():
    """`python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset -- the form the driver uses) must measure TWO ranks by
    itself: it re-executes under torch.distributed.run and the JSON line says n_gpus 2 with per-rank timings.  (gloo +
    one device here; on an 8-GPU node the same path runs RCCL, one device per rank.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CS_BENCH_ONE_DEVICE="1", CS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--objects", "4"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2
    assert len(d["ranks"]["ms_per_step_by_rank"]) == 2 and d["ranks"]["ms_per_step_max"] >= d["ranks"]["ms_per_step_min"] > 0
    assert d["ranks"]["launched_by"].startswith("bench.py") and d["ranks"]["broadcast_ms"] > 0
    assert abs(d["ms_per_step"] - d["ranks"]["ms_per_step_max"]) < 0.5 * d["ms_per_step"]


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_device():
    """`bench.py --gpus 8` end to end -- eight ranks spawned by bench.py itself, the packed-conditioning
    broadcast, barrier + max-over-ranks timing, ONE JSON line whose value is the eight-rank aggregate it claims (reduced
    UNet, one device + gloo: what is exercised is the control flow the driver's 8-GPU run takes over RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CS_BENCH_ONE_DEVICE="1", CS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--small", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--objects", "2"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2 and d["finite"] is True
    assert d["ranks"]["world_size"] == 8 and len(d["ranks"]["ms_per_step_by_rank"]) == 8
    assert all(t > 0 for t in d["ranks"]["ms_per_step_by_rank"])
    assert abs(d["ms_per_step"] - max(d["ranks"]["ms_per_step_by_rank"])) < 0.5 * d["ms_per_step"]
    # whole-job aggregate: eight ranks each advance their objects one step per ms_per_step
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["config"]["objects_per_gpu"] == 2 and "x8" in d["config"]["parallelism"]


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """No silent one-rank fallback: with fewer visible devices than --gpus (here: none) the launcher exits non-zero and
    prints no metric line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CS_BENCH_ONE_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode != 0
    assert '{"metric"' not in r.stdout and "refusing" in r.stderr

"""One-rank RCCL group on the GPU box (tests/test_rccl_single_rank_gpu.py): with CS_DIST_FORCE_COLLECTIVES=1 the
early returns of commonscenes_amd/dist.py at world size 1 are skipped, so `backend="nccl"` (RCCL) runs the very calls
an 8-GPU job makes -- device-to-device broadcast of the packed conditioning, all_gather_into_tensor of a padded slab
(incl. an empty local shard), all_reduce MAX of the fp32 failure / math flags -- on device tensors, without a host hop."""
import json
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["CS_DIST_FORCE_COLLECTIVES"] = "1"

import torch
import torch.distributed as td


def main():
    port = int(os.environ.get("CS_RCCL_PORT", "29571"))
    torch.cuda.set_device(0)
    td.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    from commonscenes_amd import dist as D
    res = dict(backend=td.get_backend(), world=td.get_world_size(), forced=D._force())
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    # (1) broadcast of the packed [x_T | uc | c] buffer, device to device
    B = 5
    x_T = torch.randn(1, 3, 16, 16, 16, generator=g).to(dev)
    uc = torch.randn(B, 1, 1280, generator=g).to(dev)
    c = torch.randn(B, 1, 1280, generator=g).to(dev)
    x2, uc2, c2 = D.broadcast_conditioning(x_T, uc, c, B, dev)
    res["bcast_equal"] = bool(torch.equal(x2, x_T) and torch.equal(uc2, uc) and torch.equal(c2, c))
    res["bcast_on_device"] = bool(x2.is_cuda and uc2.is_cuda and c2.is_cuda)
    # the concat family's condition volume (cond_shape) and a non-contiguous source view
    cv = torch.randn(B, 2, 16, 16, 16, generator=g).to(dev)[:, :1]
    x3, u3, c3 = D.broadcast_conditioning(x_T, cv, cv, B, dev, ctx_dim=4096, cond_shape=(1, 16, 16, 16))
    res["bcast_volume_equal"] = bool(torch.equal(c3, cv) and c3.shape == (B, 1, 16, 16, 16))
    # (2) all-gather of the decoded slab: full shard, a non-contiguous slab, an EMPTY local shard, B = 0
    sdf = torch.randn(B, 1, 64, 64, 64, generator=g).to(dev)
    out = D.all_gather_objects(sdf, B)
    res["gather_equal"] = bool(out.is_cuda and torch.equal(out, sdf))
    nc = torch.randn(B, 2, 8, 8, 8, generator=g).to(dev)[:, 1:]
    res["gather_noncontig_equal"] = bool(torch.equal(D.all_gather_objects(nc, B), nc))
    lat = torch.randn(B, 3, 16, 16, 16, generator=g).to(dev)
    res["gather_latents_equal"] = bool(torch.equal(D.all_gather_objects(lat, B), lat))
    empty = torch.empty((0, 1, 64, 64, 64), dtype=torch.float32, device=dev)
    res["gather_b0_shape"] = list(D.all_gather_objects(empty, 0).shape)
    # the EMPTY-SHARD shape of a job with more ranks than objects (total < world cannot happen at world 1, so it is emulated:
    # a zero-row local slab joins a padded all-gather sized for `total` rows): the pad-to-largest-shard copy of nothing, the
    # device-to-device all_gather_into_tensor of a slab that is all padding, the per-rank slicing of the result
    e2 = D.all_gather_objects(torch.empty((0, 1, 8, 8, 8), dtype=torch.float32, device=dev), 3)
    res["gather_empty_local_shape"] = list(e2.shape)
    res["gather_empty_local_zero"] = bool(e2.is_cuda and float(e2.abs().max()) == 0.0)
    # (3) the fp32 flags
    res["any_failed_false"] = D.any_rank_failed(False, dev)
    res["any_failed_true"] = D.any_rank_failed(True, dev)
    fl = D.all_reduce_max(torch.tensor([0.0, 1.0], device=dev))
    res["flags"] = [float(v) for v in fl.cpu()]
    # (4) the product call: rel2shape(sharded=True) through the same collectives == the unsharded run, bit for bit
    from commonscenes_amd import synth
    from test_model_gpu import _scene
    with tempfile.TemporaryDirectory() as tdir:
        m = _scene(Path(tdir))
        n = 3
        cc = synth.gaussian_like("r1:c", (n, 1, 1280)).cuda()
        uu = synth.gaussian_like("r1:uc", (n, 1, 1280)).cuda()
        xx = synth.gaussian_like("r1:xT", (1, 3, 16, 16, 16))
        data = {"sdf": torch.zeros(n, 1), "rel": cc, "uc": uu}
        kw = dict(ddim_steps=50, uc_scale=3.0, x_T=xx, return_latents=True, max_steps=2)
        a, la = m.Diff.rel2shape(data, sharded=True, **kw)
        os.environ["CS_DIST_FORCE_COLLECTIVES"] = "0"
        b, lb = m.Diff.rel2shape(data, sharded=False, **kw)
        torch.cuda.synchronize()
        res["rel2shape_equal"] = bool(torch.equal(a, b) and torch.equal(la, lb))
        res["rel2shape_shape"] = list(a.shape)
    td.barrier()
    td.destroy_process_group()
    # (5) bench.py itself through `init_process_group(backend="nccl", device_id=...)` (bench.py:414 -- the call the driver's
    # multi-GPU launch makes) in a forced one-rank group, with dist.py's collectives forced on: the reduced-width model, two
    # steps; its broadcast / barrier / all-reduce MAX / all-gather all run on RCCL
    import subprocess
    env = dict(os.environ, CS_BENCH_FORCE_GROUP="1", CS_DIST_FORCE_COLLECTIVES="1", CS_RCCL_PORT=str(port + 1))
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--small",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    res["bench_rc"] = r.returncode
    res["bench_tail"] = (r.stdout[-1500:] + r.stderr[-1500:]) if r.returncode else ""
    if r.returncode == 0:
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        res["bench_group"] = line["ranks"].get("process_group")
        res["bench_backend"] = line["ranks"].get("backend")
        res["bench_finite"] = line["finite"]
    Path(os.environ["CS_RCCL_OUT"]).write_text(json.dumps(res))


if __name__ == "__main__":
    main()

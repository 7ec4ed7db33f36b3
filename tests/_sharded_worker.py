"""Worker of tests/test_sharded_gpu.py: one rank of the object-sharded product path (launched by torch.distributed.run).

Every rank builds the same Sg2ScVAEModel on synthetic weights and calls the PRODUCT entry points -- rel2shape(...,
sharded=True) and Sg2ScVAEModel.sample(gen_shape=True) -- with the real HIP kernels; rank 0 also computes the
unsharded results in-process and checks SURVEY 8e's equivalence: gathered == single-rank, bit for bit per object.
A 1-GPU box cannot host two RCCL ranks, so CS_ONE_DEVICE=1 puts every rank on cuda:0 and the backend is gloo; on a
multi-GPU node the same script runs with the nccl (= RCCL) backend, one GPU per rank."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
import torch.distributed as td

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one = os.environ.get("CS_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one else int(os.environ.get("LOCAL_RANK", rank)))
    td.init_process_group(os.environ.get("CS_DIST_BACKEND", "gloo" if one else "nccl"), rank=rank, world_size=ws)
    from commonscenes_amd import dist as D
    from commonscenes_amd import synth
    from test_model_gpu import _scene
    nobj = int(os.environ.get("CS_SHARD_OBJECTS", "9"))
    with tempfile.TemporaryDirectory() as tdir:
        m = _scene(Path(tdir))
        c = synth.gaussian_like("sh:c", (nobj, 1, 1280)).cuda()
        uc = synth.gaussian_like("sh:uc", (nobj, 1, 1280)).cuda()
        x_T = synth.gaussian_like("sh:xT", (1, 3, 16, 16, 16))
        if rank != 0:
            # the broadcast must make rank 0's conditioning authoritative: hand the other ranks garbage
            c, uc, x_T = c * 0 + 7.0, uc * 0 - 3.0, x_T * 0
        data = {"sdf": torch.zeros(nobj, 1), "rel": c, "uc": uc}
        kw = dict(ddim_steps=50, uc_scale=3.0, x_T=x_T, mini_B=32, return_latents=True, max_steps=2)
        sdf, lat = m.Diff.rel2shape(data, sharded=True, **kw)
        torch.cuda.synchronize()
        res = dict(rank=rank, shape=list(sdf.shape), finite=bool(torch.isfinite(sdf).all()))
        if rank == 0:
            # (a) identical per-rank mini-batching: each shard on its own, concatenated
            parts = [m.Diff.rel2shape({"sdf": torch.zeros(hi - lo, 1), "rel": c[lo:hi], "uc": uc[lo:hi]},
                                      sharded=False, **kw)
                     for lo, hi in (D.shard_range(nobj, ws, r) for r in range(ws)) if hi > lo]
            ref_sdf = torch.cat([p[0] for p in parts])
            ref_lat = torch.cat([p[1] for p in parts])
            res["equal_same_minibatching"] = bool(torch.equal(sdf, ref_sdf) and torch.equal(lat, ref_lat))
            # (b) the plain single-rank call over all objects
            one_sdf, one_lat = m.Diff.rel2shape(data, sharded=False, **kw)
            res["equal_single_call"] = bool(torch.equal(sdf, one_sdf))
            res["rel_l2_single_call"] = float((sdf.double() - one_sdf.double()).norm() / one_sdf.double().norm())
            res["lat_rel_l2_single_call"] = float((lat.double() - one_lat.double()).norm() / one_lat.double().norm())
        if os.environ.get("CS_SHARD_LIGHT") == "1":       # the 8-rank pre-flight: the sharded call and its equivalence only
            td.barrier()
            Path(os.environ["CS_SHARD_OUT"], f"rank{rank}.json").write_text(json.dumps(res))
            td.destroy_process_group()
            return
        # the outer API: every rank calls sample() with the same scene; sharding is automatic under a process group
        g = synth.random_scene_graph(6, seed=11)
        O = g["objs"].shape[0]
        dec_sdfs = torch.zeros(O, 1, 4, 4, 4)
        dec_sdfs[:6] = 1.0
        args = (None, np.zeros(64), np.eye(64), g["objs"], g["triples"], dec_sdfs, g["text_feats"], g["rel_feats"])
        xs = synth.gaussian_like("sh:xs", (1, 3, 16, 16, 16))
        boxes, gen = m.sample(*args, gen_shape=True, z=g["z"], x_T=xs, ddim_steps=2)
        torch.cuda.synchronize()
        res["sample_shape"] = list(gen.shape)
        if rank == 0:
            _, gen1 = m.sample(*args, gen_shape=True, z=g["z"], x_T=xs, ddim_steps=2, sharded=False)
            res["sample_rel_l2"] = float((gen.double() - gen1.double()).norm() / gen1.double().norm())
        # more ranks than objects: an empty shard must not hang or crash the collective
        tiny = {"sdf": torch.zeros(1, 1), "rel": c[:1], "uc": uc[:1]}
        s1 = m.Diff.rel2shape(tiny, sharded=True, ddim_steps=50, uc_scale=3.0, x_T=x_T, max_steps=1)
        res["tiny_shape"] = list(s1.shape)
        res["tiny_finite"] = bool(torch.isfinite(s1).all())
        # a failure on ONE rank (an F16X3 overflow in the shard that holds the last object: its context is
        # 1e4) must reach every rank before the all-gather -- policy 'raise': the owner raises CsOverflowError, its
        # peers RuntimeError, nobody hangs; policy 'fp32': the owner falls back and ALL ranks switch math together
        from commonscenes_amd import lib as L
        cbad = c.clone()
        cbad[nobj - 1] = 1.0e4
        bad = {"sdf": torch.zeros(nobj, 1), "rel": cbad, "uc": uc}
        kwb = dict(ddim_steps=50, uc_scale=3.0, x_T=x_T, mini_B=32, max_steps=1, sharded=True)
        # r5: with the device-side bound of the context linears and the static scales of the transformer-internal operands
        # this input no longer overflows anything (res["no_flag_with_static_scales"]); the failure plumbing is exercised
        # with those features off (the r4 behaviour)
        m.Diff.overflow_policy = "raise"
        try:
            sg = m.Diff.rel2shape(bad, **kwb)
            res["no_flag_with_static_scales"] = bool(torch.isfinite(sg).all())
        except Exception as e:          # noqa: BLE001
            res["no_flag_with_static_scales"] = f"{type(e).__name__}: {e}"
        with L.debug_override(no_static_scales=1):
            try:
                m.Diff.rel2shape(bad, **kwb)
                res["raise_policy"] = "no exception"
            except L.CsOverflowError:
                res["raise_policy"] = "CsOverflowError"
            except RuntimeError as e:
                res["raise_policy"] = "RuntimeError" if "another rank failed" in str(e) else f"RuntimeError: {e}"
            m.Diff.overflow_policy = "fp32"
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sb = m.Diff.rel2shape(bad, **kwb)
            res["fallback_finite"] = bool(torch.isfinite(sb).all())
            res["math_after_fallback"] = int(m.Diff.df.math)
    td.barrier()
    # one file per rank: the ranks share stdout and their lines can interleave
    Path(os.environ["CS_SHARD_OUT"], f"rank{rank}.json").write_text(json.dumps(res))
    td.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""BASELINE configs[4] (C5) exerciser: the call sequence of scripts/eval_3dfront.py:484-722 for the v2_full model with
--gen_shape True --visualize / --evaluate_diversity, on a synthetic SG-FRONT-like dataset (the real dataset, CLIP
features and trained checkpoints are not reachable offline), through this package only:

    VAE(type='v2_full', ...).load_networks / compute_statistics           scripts/eval_3dfront.py:150-175
    for each scene of the test loader                                      :484-510
        boxes, shapes = model.sample_box_and_shape(..., gen_shape=True)    :513
        angles = -180 + (argmax + 1) * 15                                  :514-516
        meshes  = sdf_to_mesh(shapes, render_all=True)                     helpers/util.py:298 (--visualize)
        diversity: num_samples x [sample_box_and_shape -> sdf_to_mesh -> verts_list -> sample_points(5000)
                   -> normalize]; chamfer(shape_k, shape_k+1)              :577-690
    box / angle / shape diversity summaries                                :660-690

Under `torch.distributed.run` (one process per GPU) every rank runs the same script and rel2shape shards the objects
over the ranks (one broadcast in, one all-gather out) -- the 8xMI355X form of C5.  `--attention f16` selects the opt-in
"fp16 MFMA attention" C5 names (reduced precision; the default is the fp32-grade F16X3 attention).

    python tools/eval_walkthrough.py [--scenes 2] [--samples 3] [--width 32] [--ddim-steps 2] [--attention f16]
prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch
import yaml

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

VOCAB = dict(object_idx_to_name=[f"obj{i}\n" for i in range(35)], pred_idx_to_name=[f"pred{i}\n" for i in range(16)],
             object_idx_to_name_grained=[f"objg{i}\n" for i in range(35)])


def sample_points(points_list, num):                         # helpers/util.py:31-44
    out = []
    for pc in points_list:
        n = pc.size(0)
        idx = torch.randperm(n)[:num] if n >= num else torch.randint(n, size=(num,))
        out.append(pc[idx.to(pc.device)])
    return out


def normalize(vertices, scale=1):                            # scripts/eval_3dfront.py:783-796
    for a in range(3):
        lo, hi = np.amin(vertices[:, a]), np.amax(vertices[:, a])
        vertices[:, a] += -lo - (hi - lo) * 0.5
    return vertices / np.max(vertices, axis=0) * scale


def synthetic_loader(n_scenes, seed, objects=None):
    """`objects`: shaped objects per scene (default 4, 5, ...).  32 gives an SG-FRONT-livingroom-sized graph: 34 nodes
    (objects + floor + _scene_) and ~160 triples (dataset/threedfront_dataset.py:448-452: an `in` edge from every node
    to `_scene_` plus the typed relations)."""
    from commonscenes_amd import synth
    out = []
    for s in range(n_scenes):
        nobj = (4 + s) if objects is None else objects
        g = synth.random_scene_graph(nobj, seed=seed + s)
        O = g["objs"].shape[0]
        sdfs = torch.zeros(O, 1, 4, 4, 4)
        sdfs[:nobj] = 1.0                                     # floor / _scene_ carry all-zero SDFs
        boxes = torch.cat([synth.gaussian_like(f"ev:{s}", (O, 6)), torch.randint(0, 24, (O, 1)).float()], dim=1)
        out.append({"scan_id": [f"Synthetic-{s}"], "instance_id": [list(range(O))],
                    "decoder": {"objs": g["objs"], "tripltes": g["triples"], "boxes": boxes, "sdfs": sdfs,
                                "text_feats": g["text_feats"], "rel_feats": g["rel_feats"]}})
    return out


def build_experiment(tmp: Path, width: int):
    """an experiment directory in the reference's layout: checkpoint/model{epoch}.pth with synthetic weights."""
    from commonscenes_amd import configs as K
    from commonscenes_amd import synth
    from commonscenes_amd.scene import scene_param_shapes
    from commonscenes_amd.unet import unet_param_shapes
    from commonscenes_amd.vqvae import vqvae_param_shapes
    ucfg = K.reduced(K.UNET_CROSSATTN, width) if width != 224 else dict(K.UNET_CROSSATTN)
    gen = "cuda" if torch.cuda.is_available() else "cpu"      # same values either way (synth.py); the device is faster
    df_yaml = dict(model=dict(params=dict(conditioning_key="crossattn", **K.DIFFUSION)),
                   unet=dict(params={k: (list(v) if isinstance(v, tuple) else v) for k, v in ucfg.items()}))
    vq_yaml = dict(model=dict(params=dict(embed_dim=K.VQVAE_EMBED_DIM, n_embed=K.VQVAE_N_EMBED,
                                          ddconfig={k: (list(v) if isinstance(v, tuple) else v)
                                                    for k, v in K.VQVAE_DDCONFIG.items()})))
    (tmp / "df.yaml").write_text(yaml.safe_dump(df_yaml))
    (tmp / "vq.yaml").write_text(yaml.safe_dump(vq_yaml))
    opt = dict(hyper=dict(device="cuda", batch_size=4),
               network=dict(df_cfg=str(tmp / "df.yaml"), vq_cfg=str(tmp / "vq.yaml"), vq_ckpt=None), misc=dict(seed=111))
    cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
    ck = dict(synth.synth_state_dict(scene_param_shapes(35, 16)))
    ck["vqvae"] = synth.synth_state_dict(vqvae_param_shapes(K.VQVAE_DDCONFIG, K.VQVAE_N_EMBED, K.VQVAE_EMBED_DIM))
    ck["df"] = cpu(synth.synth_state_dict(unet_param_shapes(ucfg), device=gen))
    ck.update(opt={}, epoch=100, counter=0)
    (tmp / "checkpoint").mkdir(exist_ok=True)
    torch.save(ck, tmp / "checkpoint" / "model100.pth")
    return opt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=2)
    ap.add_argument("--samples", type=int, default=3, help="diversity runs per scene (eval_3dfront.py num_samples)")
    ap.add_argument("--width", type=int, default=32, help="UNet model_channels (224 = the shipped network)")
    ap.add_argument("--ddim-steps", type=int, default=2)
    ap.add_argument("--attention", choices=["same", "f16"], default="same")
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--objects", type=int, default=None,
                    help="shaped objects per scene (default 4, 5, ...; 32 = a livingroom-sized SG-FRONT graph: 34 nodes)")
    ap.add_argument("--mini-b", type=int, default=None,
                    help="sampler mini-batch (default: the reference's 7, sdfusion_txt2shape_model.py:493)")
    ap.add_argument("--compare-attention", action="store_true",
                    help="additionally sample the first scene once with the default (fp32-grade F16X3) attention and once "
                         "with the fp16 MFMA attention from the SAME z / x_T and report the latent / SDF deviation "
                         "(SURVEY 8d: fp16-attention mode is report-only)")
    ap.add_argument("--batch-scenes", action="store_true",
                    help="additionally time the shape sampling of ALL scenes scene by scene (the reference's loop, "
                         "scripts/eval_3dfront.py:484-513) against ONE coalesced call (VAE.sample_box_and_shape_many: each "
                         "scene's graph and shared x_T kept, one sampler + decode over all scenes' objects), same z / x_T, "
                         "and report the wall times and the per-object latent deviation")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    one = os.environ.get("CS_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one else int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        torch.distributed.init_process_group(os.environ.get("CS_DIST_BACKEND", "gloo" if one else "nccl"))
    torch.manual_seed(48)                                     # eval_3dfront.py:62-63
    np.random.seed(48)

    from commonscenes_amd.chamfer import chamferDist
    from commonscenes_amd.mesh import sdf_to_mesh
    from commonscenes_amd.vae import VAE
    chamfer = chamferDist()
    res = dict(scenes=[], world=world, attention=a.attention, width=a.width, ddim_steps=a.ddim_steps,
               mini_B=a.mini_b or 7)
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td)
        opt = build_experiment(tmp, a.width)
        model = VAE(type="v2_full", diff_opt=opt, vocab=VOCAB, replace_latent=True, with_changes=True, residual=True,
                    with_angles=True, clip=True, with_E2=True)
        model.load_networks(str(tmp), 100)
        if a.mini_b:
            model.vae_v2.Diff.mini_B = a.mini_b
        if a.attention == "f16":
            model.vae_v2.Diff.df.set_attention_math("f16")
        model.compute_statistics(str(tmp), 100, synthetic_loader(3, seed=900))
        model.eval()
        loader = synthetic_loader(a.scenes, seed=500, objects=a.objects)
        if a.compare_attention:
            from commonscenes_amd import synth
            d = loader[0]["decoder"]
            O = d["objs"].shape[0]
            z = synth.gaussian_like("ev:cmp:z", (O, 64))
            x_T = synth.gaussian_like("ev:cmp:xT", (1, 3, 16, 16, 16))
            got = {}
            for mode in ("same", "f16"):
                model.vae_v2.Diff.df.set_attention_math(mode)
                t0 = time.perf_counter()
                with torch.no_grad():
                    _, sdf = model.sample_box_and_shape(None, d["objs"].cuda(), d["tripltes"].cuda(), d["sdfs"],
                                                        d["text_feats"].cuda(), d["rel_feats"].cuda(), attributes=None,
                                                        gen_shape=True, ddim_steps=a.ddim_steps, z=z, x_T=x_T)
                torch.cuda.synchronize()
                got[mode] = (sdf.double(), model.vae_v2.Diff.last_latents.double(), time.perf_counter() - t0)
            rl2 = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-30))
            per_obj = [(rl2(got["f16"][1][i], got["same"][1][i])) for i in range(got["same"][1].shape[0])]
            res["attention_compare"] = dict(
                objects=int(got["same"][0].shape[0]), ddim_steps=a.ddim_steps,
                latent_rel_l2=rl2(got["f16"][1], got["same"][1]), latent_rel_l2_worst_object=max(per_obj),
                sdf_rel_l2=rl2(got["f16"][0], got["same"][0]),
                sdf_sign_flips=int(((got["f16"][0] > 0.02) != (got["same"][0] > 0.02)).sum()),
                sdf_voxels=int(got["same"][0].numel()), sample_s_default=got["same"][2], sample_s_f16=got["f16"][2],
                note="fp16 MFMA attention (one fp16 pass, fp32 softmax/accumulate) vs the default fp32-grade F16X3 "
                     "attention, same z / x_T / weights; report-only (SURVEY 8d)")
            model.vae_v2.Diff.df.set_attention_math("f16" if a.attention == "f16" else "same")
        if a.batch_scenes:
            from commonscenes_amd import synth
            scs = []
            for si, data in enumerate(loader):
                d = data["decoder"]
                O = d["objs"].shape[0]
                scs.append(dict(dec_objs=d["objs"].cuda(), dec_triplets=d["tripltes"].cuda(), dec_sdfs=d["sdfs"],
                                encoded_dec_text_feat=d["text_feats"].cuda(), encoded_dec_rel_feat=d["rel_feats"].cuda(),
                                z=synth.gaussian_like(f"ev:bs:z{si}", (O, 64)),
                                x_T=synth.gaussian_like(f"ev:bs:x{si}", (1, 3, 16, 16, 16))))

            def per_scene():
                outs, lats = [], []
                for sc in scs:
                    outs.append(model.sample_box_and_shape(None, sc["dec_objs"], sc["dec_triplets"], sc["dec_sdfs"],
                                                           sc["encoded_dec_text_feat"], sc["encoded_dec_rel_feat"],
                                                           gen_shape=True, ddim_steps=a.ddim_steps, z=sc["z"], x_T=sc["x_T"]))
                    lats.append(model.vae_v2.Diff.last_latents)
                return outs, torch.cat(lats)

            def batched():
                outs = model.sample_box_and_shape_many(scs, gen_shape=True, ddim_steps=a.ddim_steps)
                return outs, model.vae_v2.Diff.last_latents

            with torch.no_grad():
                per_scene(); batched()                        # warm-up (weight packing, allocator)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); o1, l1 = per_scene(); torch.cuda.synchronize(); t_ps = time.perf_counter() - t0
                t0 = time.perf_counter(); o2, l2 = batched(); torch.cuda.synchronize(); t_b = time.perf_counter() - t0
            dev = float((l2.double() - l1.double()).norm() / l1.double().norm())
            res["batch_scenes"] = dict(scenes=len(scs), objects=[int(o[1].shape[0]) for o in o1], ddim_steps=a.ddim_steps,
                                       scene_by_scene_s=t_ps, coalesced_s=t_b, speedup=t_ps / t_b,
                                       launch_sizes=list(model.vae_v2.Diff.last_launch_sizes), latent_rel_l2=dev,
                                       boxes_equal=all(bool(torch.equal(x[0][0], y[0][0])) for x, y in zip(o1, o2)),
                                       note="shape sampling + decode of all scenes: one sample_box_and_shape call per scene "
                                            "(eval_3dfront.py:484-513) vs VAE.sample_box_and_shape_many")
        x_T = None                                            # like the reference: fresh noise per call
        all_div_boxes, all_div_angles, all_div_chamfer = [], [], []
        t_all = time.perf_counter()
        for data in loader:
            d = data["decoder"]
            dec_objs, dec_triples = d["objs"].cuda(), d["tripltes"].cuda()
            text, rel, dec_sdfs = d["text_feats"].cuda(), d["rel_feats"].cuda(), d["sdfs"]
            t0 = time.perf_counter()
            with torch.no_grad():
                boxes_pred, shapes_pred = model.sample_box_and_shape(None, dec_objs, dec_triples, dec_sdfs, text, rel,
                                                                     attributes=None, gen_shape=True,
                                                                     ddim_steps=a.ddim_steps)
                boxes_pred, angles_pred = boxes_pred
                angles_pred = -180 + (torch.argmax(angles_pred, dim=1, keepdim=True) + 1) * 15.0
            meshes = sdf_to_mesh(shapes_pred, render_all=True)                 # --visualize (helpers/util.py:298)
            torch.cuda.synchronize()
            t_scene = time.perf_counter() - t0
            nshape = shapes_pred.shape[0]
            # --evaluate_diversity
            boxes_div, angle_div, shapes_sample = [], [], []
            for _ in range(a.samples):
                with torch.no_grad():
                    dboxes, dsdf = model.sample_box_and_shape(None, dec_objs, dec_triples, dec_sdfs, text, rel,
                                                              attributes=None, gen_shape=True, ddim_steps=a.ddim_steps)
                pts = sample_points(sdf_to_mesh(dsdf, render_all=True).verts_list(), a.points)
                dboxes, dangles = dboxes
                norm = [torch.from_numpy(normalize(p.cpu().numpy())).cuda() for p in pts]
                boxes_div.append(dboxes)
                angle_div.append(np.expand_dims(np.argmax(dangles.cpu().numpy(), 1), 1) / 24.0 * 360.0)
                shapes_sample.append(torch.stack(norm))
            bd = torch.stack(boxes_div, 1)
            all_div_boxes += torch.std(bd, dim=1).cpu().numpy().tolist()
            all_div_angles += np.stack(angle_div, 1).std(axis=1).reshape(-1).tolist()
            ss = torch.stack(shapes_sample, 1)                                  # [objects, samples, points, 3]
            for sid in range(len(ss)):
                seq = []
                for k in range(ss.shape[1] - 1):
                    d1, d2 = chamfer(ss[sid, k:k + 1].float(), ss[sid, k + 1:k + 2].float())
                    seq.append(float((torch.mean(d1) + torch.mean(d2)).cpu()))
                all_div_chamfer.append(float(np.mean(seq)))
            res["scenes"].append(dict(scan=data["scan_id"][0], nodes=int(dec_objs.shape[0]), shapes=int(nshape),
                                      triples=int(dec_triples.shape[0]),
                                      sample_s=t_scene, verts=[int(v.shape[0]) for v in meshes.verts_list()],
                                      finite=bool(torch.isfinite(shapes_pred).all() and torch.isfinite(boxes_pred).all()),
                                      angle_range=[float(angles_pred.min()), float(angles_pred.max())]))
        torch.cuda.synchronize()
        res.update(total_s=time.perf_counter() - t_all, box_std_mean=float(np.mean(all_div_boxes)),
                   angle_std_mean=float(np.mean(all_div_angles)), chamfer_diversity_mean=float(np.mean(all_div_chamfer)),
                   chamfer_diversity_n=len(all_div_chamfer))
    if rank == 0:
        print("EVAL_WALKTHROUGH " + json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

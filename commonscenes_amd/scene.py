"""Scene-graph conditioning + sampling facade behind the reference's `Sg2ScVAEModel` interface.

Mirrors (inference side, v2_full)
  model/graph.py:89-211      GraphTripleConv       -> gather-cat kernel, Linear+BN(eval)+ReLU fused GEMM
                                                        epilogues, deterministic segment-mean kernel
  model/graph.py:214-288     GraphTripleConvNet / GraphTripleConvNet2
  model/layers.py:21-38      build_mlp
  model/VAEGAN_V2FULL.py:220-242  encoder_2   (conditioning producer: uc, c)
  model/VAEGAN_V2FULL.py:261-289  decoder     (layout boxes / angles)
  model/VAEGAN_V2FULL.py:600-618  sample
Weights use the reference's state_dict keys.  BatchNorm1d runs in eval mode
(model/VAE.py:60 mlp_normalization="batch"; eval_3dfront.py:180 .eval()): it is applied in the GEMM
epilogue as (x*W + b) * s + t with s = gamma / sqrt(var + eps), t = beta - mean * s.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from . import ops
from .sdfusion import SDFusionText2ShapeModel, load_yaml

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# parameter tables
# ------------------------------------------------------------------------------------------------
def _mlp_shapes(S, p: str, dims: List[int], final_nonlinearity: bool):
    idx = 0
    for i in range(len(dims) - 1):
        S[f"{p}.{idx}.weight"] = (dims[i + 1], dims[i])
        S[f"{p}.{idx}.bias"] = (dims[i + 1],)
        final = i == len(dims) - 2
        if not final or final_nonlinearity:
            for n in ("weight", "bias", "running_mean", "running_var"):
                S[f"{p}.{idx + 1}.{n}"] = (dims[i + 1],)
            idx += 3
        else:
            idx += 1


def _gcn_shapes(S, p: str, layers: int, d_obj: int, d_pred: int, hidden: int, residual: bool = True,
                output_dim: Optional[int] = None):
    """graph.py:214-244: `output_dim` applies to the last layer only."""
    for i in range(layers):
        q = f"{p}.gconvs.{i}"
        dout = output_dim if (output_dim is not None and i >= layers - 1) else d_obj
        _mlp_shapes(S, q + ".net1", [2 * d_obj + d_pred, hidden, 2 * hidden + dout], True)
        _mlp_shapes(S, q + ".net2", [hidden, hidden, dout], True)
        if residual:
            S[q + ".linear_projection.weight"] = (dout, d_obj)
            S[q + ".linear_projection.bias"] = (dout,)
            S[q + ".linear_projection_pred.weight"] = (dout, d_pred)
            S[q + ".linear_projection_pred.bias"] = (dout,)


def scene_param_shapes(num_objs: int, num_preds: int, embedding_dim: int = 64, clip_dim: int = 512,
                       gconv_num_layers: int = 5, num_box_params: int = 6, n_angle: int = 24,
                       rel_dims=(960, 1280)) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict entries of the reference Sg2ScVAEModel that the sampling path reads
    (VAEGAN_V2FULL.py:69-166 with embedding_dim=64, decoder_cat=True, clip=True, use_angles=True,
    use_E2=True, residual=True, mlp_normalization='batch')."""
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    e = embedding_dim
    d = 2 * e + clip_dim                      # 640
    hid = 4 * e                               # 256
    S["obj_embeddings_dc.weight"] = (num_objs + 1, e)
    S["pred_embeddings_dc.weight"] = (num_preds, 2 * e)
    _gcn_shapes(S, "gconv_net_ec_rel", gconv_num_layers, d, d, hid)
    _mlp_shapes(S, "rel_mlp", [d, rel_dims[0], rel_dims[1]], False)
    _gcn_shapes(S, "gconv_net_dc", gconv_num_layers, d, d, hid)
    _mlp_shapes(S, "d3_net", [d, hid, num_box_params], False)
    _mlp_shapes(S, "angle_net", [d, hid, n_angle], False)
    return S


# ------------------------------------------------------------------------------------------------
# MLP / GCN on the HIP kernels
# ------------------------------------------------------------------------------------------------
class _MLP:
    """build_mlp (layers.py:21-38) with BatchNorm1d(eval) folded into the GEMM epilogue."""

    def __init__(self, sd: Dict[str, Tensor], p: str, n_layers: int, final_nonlinearity: bool):
        self.layers = []
        idx = 0
        for i in range(n_layers):
            w = ops.pack_weight(sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"])
            final = i == n_layers - 1
            if not final or final_nonlinearity:
                bn = f"{p}.{idx + 1}"
                s = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
                t = sd[bn + ".bias"] - sd[bn + ".running_mean"] * s
                self.layers.append((w, s.contiguous(), t.contiguous(), L.ACT_RELU))
                idx += 3
            else:
                self.layers.append((w, None, None, L.ACT_NONE))
                idx += 1

    def __call__(self, x: Tensor) -> Tensor:
        for w, s, t, act in self.layers:
            x = ops.linear(x, w, scale=s, shift=t, act=act)
        return x


class GraphTripleConv:
    """graph.py:89-211 (pooling='avg')."""

    def __init__(self, sd: Dict[str, Tensor], p: str):
        self.net1 = _MLP(sd, p + ".net1", 2, True)
        self.net2 = _MLP(sd, p + ".net2", 2, True)
        self.H = sd[p + ".net2.0.weight"].shape[1]
        self.Dout = sd[p + ".net1.3.weight"].shape[0] - 2 * self.H
        self.residual = (p + ".linear_projection.weight") in sd
        if self.residual:
            self.proj = ops.pack_weight(sd[p + ".linear_projection.weight"], sd[p + ".linear_projection.bias"])
            self.proj_pred = ops.pack_weight(sd[p + ".linear_projection_pred.weight"],
                                             sd[p + ".linear_projection_pred.bias"])

    def __call__(self, obj_vecs: Tensor, pred_vecs: Tensor, edges: Tensor) -> Tuple[Tensor, Tensor]:
        H, Dout = self.H, self.Dout
        cur_t = ops.gcn_gather_cat(obj_vecs, pred_vecs, edges)            # [T, 2*Din + Dp]
        new_t = self.net1(cur_t)                                          # [T, 2H + Dout] = [s | p | o]
        pooled = ops.gcn_segment_mean(new_t, edges, obj_vecs.shape[0], H, H + Dout)
        new_p = new_t[:, H:H + Dout]
        if self.residual:
            # new_obj = net2(pooled) + proj(obj): the projection GEMM takes net2's output as its residual
            new_obj = ops.linear(obj_vecs, self.proj, res=self.net2(pooled))
            new_p = ops.linear(pred_vecs, self.proj_pred, res=new_p)
        else:
            new_obj = self.net2(pooled)
            new_p = new_p.contiguous()
        return new_obj, new_p


class GraphTripleConvNet:
    """graph.py:214-248 / 250-288 (GraphTripleConvNet2 is the same forward)."""

    def __init__(self, sd: Dict[str, Tensor], p: str, num_layers: int):
        self.gconvs = [GraphTripleConv(sd, f"{p}.gconvs.{i}") for i in range(num_layers)]

    def __call__(self, obj_vecs, pred_vecs, edges):
        for g in self.gconvs:
            obj_vecs, pred_vecs = g(obj_vecs, pred_vecs, edges)
        return obj_vecs, pred_vecs


GraphTripleConvNet2 = GraphTripleConvNet


# ------------------------------------------------------------------------------------------------
# the scene model
# ------------------------------------------------------------------------------------------------
class Sg2ScVAEModel:
    """Inference drop-in for the v2_full `Sg2ScVAEModel` (VAEGAN_V2FULL.py:17-760): `encoder_2`,
    `decoder`, `sample`, plus the checkpoint surface (`load_state_dict` / `state_dict`)."""

    def __init__(self, vocab, diff_opt, diffusion_bs=8, embedding_dim=128, batch_size=32, train_3d=True,
                 decoder_cat=False, num_box_params=6, distribution_before=True, gconv_pooling="avg",
                 gconv_num_layers=5, mlp_normalization="none", vec_noise_dim=0, use_E2=False, use_AE=False,
                 replace_latent=False, residual=False, use_angles=False, clip=True, device="cuda",
                 resolve_dir=None):
        if not (decoder_cat and use_E2 and clip and residual and mlp_normalization == "batch"
                and gconv_pooling == "avg"):
            raise NotImplementedError("only the eval-time v2_full configuration is implemented: decoder_cat, "
                                      "use_E2, clip, residual, mlp_normalization='batch', avg pooling "
                                      "(model/VAE.py:60-62, scripts/eval_3dfront.py:65-68)")
        self.vocab = vocab
        self.embedding_dim = embedding_dim
        self.use_angles = use_angles
        self.num_box_params = num_box_params
        self.gconv_num_layers = gconv_num_layers
        self.clip = clip
        self.device = torch.device(device)
        self.obj_classes_list = list(set(vocab["object_idx_to_name"]))
        self.edge_list = list(set(vocab["pred_idx_to_name"]))
        self.num_objs, self.num_preds = len(self.obj_classes_list), len(self.edge_list)
        self.diff_cfg = diff_opt if isinstance(diff_opt, dict) else load_yaml(diff_opt)
        self.diffusion_bs = diffusion_bs
        self.Diff = SDFusionText2ShapeModel(self.diff_cfg, resolve_dir=resolve_dir)
        if self.Diff.df.conditioning_key not in ("crossattn", "concat"):
            raise NotImplementedError(f"conditioning_key={self.Diff.df.conditioning_key!r}")
        # VAEGAN_V2FULL.py:152-155: the concat family's rel_mlp ends in a 16^3 = 4096-voxel condition volume
        rel_dims = (1280, 4096) if self.Diff.df.conditioning_key == "concat" else (960, 1280)
        self.shapes = scene_param_shapes(self.num_objs, self.num_preds, embedding_dim,
                                         gconv_num_layers=gconv_num_layers, num_box_params=num_box_params,
                                         rel_dims=rel_dims)
        self._sd: Dict[str, Tensor] = {}
        self._nets = None

    # ---- checkpoint surface ----
    def load_state_dict(self, sd, strict: bool = False):
        """Accepts the reference checkpoint's main state_dict; entries the sampling path does not read
        (encoder / manipulation nets, *_ec embeddings, mean/var heads) are ignored."""
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise RuntimeError(f"Sg2ScVAEModel.load_state_dict: missing {missing[:5]} (+{len(missing) - 5})")
        for k, shp in self.shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {shp}")
            self._sd[k] = sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._nets = None
        return self

    def state_dict(self):
        return OrderedDict(self._sd)

    def eval(self):
        return self

    def cuda(self):
        return self

    def _build(self):
        sd = self._sd
        if not sd:
            raise RuntimeError("Sg2ScVAEModel: weights not loaded")
        n = self.gconv_num_layers
        self._nets = dict(
            ec_rel=GraphTripleConvNet(sd, "gconv_net_ec_rel", n),
            rel_mlp=_MLP(sd, "rel_mlp", 2, False),
            dc=GraphTripleConvNet(sd, "gconv_net_dc", n),
            d3=_MLP(sd, "d3_net", 2, False),
            angle=_MLP(sd, "angle_net", 2, False),
        )

    def _node_edge_feats(self, objs, triples, text_feat, rel_feat, z):
        dev = self.device
        triples = triples.to(dev)
        objs = objs.to(dev)
        s, p, o = triples[:, 0].contiguous(), triples[:, 1].contiguous(), triples[:, 2].contiguous()
        edges = torch.stack([s, o], dim=1).contiguous()
        O, T = objs.shape[0], triples.shape[0]
        e = self.embedding_dim
        cd = text_feat.shape[1]
        # [clip | embedding | z] assembled in place: one buffer, three producers (VAEGAN_V2FULL.py:225-233)
        obj_vecs = torch.empty((O, cd + e + z.shape[1]), dtype=torch.float32, device=dev)
        obj_vecs[:, :cd].copy_(text_feat)
        ops.embedding(self._sd["obj_embeddings_dc.weight"], objs, out=obj_vecs[:, cd:cd + e])
        obj_vecs[:, cd + e:].copy_(z)
        pe = self._sd["pred_embeddings_dc.weight"].shape[1]
        pred_vecs = torch.empty((T, cd + pe), dtype=torch.float32, device=dev)
        pred_vecs[:, :cd].copy_(rel_feat)
        ops.embedding(self._sd["pred_embeddings_dc.weight"], p, out=pred_vecs[:, cd:])
        return obj_vecs, pred_vecs, edges

    @torch.no_grad()
    def encoder_2(self, z, objs, triples, dec_text_feat, dec_rel_feat, attributes=None, manipulate=False):
        """VAEGAN_V2FULL.py:220-242 -> (uc, c), each (O, 1, 1280) -- (O, 1, 4096) in the concat family."""
        if self._nets is None:
            self._build()
        rel_vecs_, pred_vecs_, edges = self._node_edge_feats(objs, triples, dec_text_feat, dec_rel_feat, z)
        rel2, _ = self._nets["ec_rel"](rel_vecs_, pred_vecs_, edges)
        c = self._nets["rel_mlp"](rel2).unsqueeze(1)
        uc = self._nets["rel_mlp"](rel_vecs_).unsqueeze(1)
        return uc, c

    @torch.no_grad()
    def decoder(self, z, objs, triples, dec_text_feat, dec_rel_feat, attributes=None, manipulate=False):
        """VAEGAN_V2FULL.py:261-289 (decoder_cat=True) -> d3_pred [, log-softmax angles]."""
        if self._nets is None:
            self._build()
        obj_vecs_, pred_vecs_, edges = self._node_edge_feats(objs, triples, dec_text_feat, dec_rel_feat, z)
        obj_vecs_, _ = self._nets["dc"](obj_vecs_, pred_vecs_, edges)
        d3 = self._nets["d3"](obj_vecs_)
        if self.use_angles:
            return d3, ops.log_softmax(self._nets["angle"](obj_vecs_))
        return d3

    @torch.no_grad()
    def sample(self, point_classes_idx, mean_est, cov_est, dec_objs, dec_triplets, dec_sdfs,
               encoded_dec_text_feat, encoded_dec_rel_feat, attributes=None, gen_shape=False,
               z: Optional[Tensor] = None, x_T: Optional[Tensor] = None, ddim_steps: int = 100):
        """VAEGAN_V2FULL.py:600-618.  Extensions: `z`, `x_T` and `ddim_steps` can be injected (the reference
        draws z from numpy's global RNG and x_T from a time-seeded torch RNG)."""
        dev = self.device
        if z is None:
            z = torch.from_numpy(np.random.multivariate_normal(mean_est, cov_est, dec_objs.size(0))).float()
        z = z.to(dev)
        text = encoded_dec_text_feat.to(device=dev, dtype=torch.float32)
        rel = encoded_dec_rel_feat.to(device=dev, dtype=torch.float32)
        gen_sdf = None
        if gen_shape:
            un_rel_feat, rel_feat = self.encoder_2(z, dec_objs, dec_triplets, text, rel, attributes)
            # nodes whose ground-truth SDF is non-zero (drops floor and _scene_); dec_sdfs stays on the host
            sdf_candidates = dec_sdfs
            mask = torch.ne(sdf_candidates, torch.zeros_like(sdf_candidates[0]))
            ids = torch.unique(torch.where(mask)[0])
            ids_d = ids.to(dev)
            diff_dict = {"sdf": dec_sdfs[ids], "rel": rel_feat[ids_d], "uc": un_rel_feat[ids_d]}
            gen_sdf = self.Diff.rel2shape(diff_dict, ddim_steps=ddim_steps, uc_scale=3., x_T=x_T)
        return self.decoder(z, dec_objs, dec_triplets, text, rel, attributes), gen_sdf

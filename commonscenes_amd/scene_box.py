"""Layout-only scene model (BASELINE configs[0], `v2_box`) behind the reference's `VAEGAN_V2BOX.Sg2ScVAEModel`
interface, on the same HIP kernels as the v2_full conditioning path (gather-cat, Linear+BN(eval)+ReLU GEMM
epilogues, deterministic segment mean, embedding gather, log-softmax).

Mirrors (inference side; model/VAE.py:52-56 builds it with embedding_dim=64, decoder_cat=True,
mlp_normalization="batch", gconv_num_layers=5, use_angles / residual / replace_latent from the CLI)
  model/VAEGAN_V2BOX.py:13-125   __init__ (parameter layout)
  model/VAEGAN_V2BOX.py:127-158  encoder               boxes (+ angle classes) -> (mu, logvar)
  model/VAEGAN_V2BOX.py:160-173  manipulate            [z | change] -> manipulated latent
  model/VAEGAN_V2BOX.py:175-200  decoder               z -> boxes (, log-softmax angle logits)
  model/VAEGAN_V2BOX.py:247-273  decoder_with_additions
  model/VAEGAN_V2BOX.py:292-344  decoder_with_changes
  model/VAEGAN_V2BOX.py:456-461  sampleBoxes
Random draws follow the reference call for call (numpy's global RNG: multivariate_normal per added node, then
normal(0, 1, embedding_dim) per touched node), so seeding numpy reproduces its latents.
Training (forward, losses) and `sampleShape` (v1 point auto-encoder) are out of scope.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from . import ops
from .scene import GraphTripleConvNet, _MLP, _gcn_shapes, _mlp_shapes

Tensor = torch.Tensor


def box_param_shapes(num_objs: int, num_preds: int, embedding_dim: int = 64, clip_dim: int = 512,
                     gconv_num_layers: int = 5, input_dim: int = 6, n_angle: int = 24, use_angles: bool = True,
                     residual: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict entries (minus BatchNorm's num_batches_tracked counters) of the reference v2_box model with
    decoder_cat=True, mlp_normalization='batch' (VAEGAN_V2BOX.py:27-110)."""
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    e = embedding_dim
    d = 2 * e + clip_dim                       # 640
    hid = 4 * e
    box_e = e * 3 // 4 if use_angles else e
    ang_e = e // 4
    S["obj_embeddings_ec.weight"] = (num_objs + 1, e)
    S["pred_embeddings_ec.weight"] = (num_preds, 2 * e)
    S["obj_embeddings_dc.weight"] = (num_objs + 1, e)
    S["pred_embeddings_dc.weight"] = (num_preds, 2 * e)
    S["pred_embeddings_man_dc.weight"] = (num_preds, 3 * e)
    S["d3_embeddings.weight"] = (box_e, input_dim)
    S["d3_embeddings.bias"] = (box_e,)
    if use_angles:
        S["angle_embeddings.weight"] = (n_angle, ang_e)
    _mlp_shapes(S, "mean_var", [d, hid, 2 * e], True)
    _mlp_shapes(S, "mean", [2 * e, box_e], False)
    _mlp_shapes(S, "var", [2 * e, box_e], False)
    if use_angles:
        _mlp_shapes(S, "angle_mean_var", [d, hid, 2 * e], True)
        _mlp_shapes(S, "angle_mean", [2 * e, ang_e], False)
        _mlp_shapes(S, "angle_var", [2 * e, ang_e], False)
    _gcn_shapes(S, "gconv_net_ec", gconv_num_layers, d, d, hid, residual)
    _gcn_shapes(S, "gconv_net_dc", gconv_num_layers, d, d, hid, residual)
    dm = 3 * e + clip_dim                      # 704
    _gcn_shapes(S, "gconv_net_manipulation", min(gconv_num_layers, 5), dm, dm, hid, residual, output_dim=e)
    _mlp_shapes(S, "d3_net", [d, hid, input_dim], False)
    if use_angles:
        _mlp_shapes(S, "angle_net", [d, hid, n_angle], False)
    return S


class Sg2ScVAEModel:
    """Inference drop-in for `model.VAEGAN_V2BOX.Sg2ScVAEModel`."""

    def __init__(self, vocab, embedding_dim=128, batch_size=32, train_3d=True, decoder_cat=False, input_dim=6,
                 gconv_pooling="avg", gconv_num_layers=5, mlp_normalization="none", vec_noise_dim=0, use_AE=False,
                 replace_latent=False, residual=False, use_angles=False, device="cuda"):
        if not (decoder_cat and mlp_normalization == "batch" and gconv_pooling == "avg"):
            raise NotImplementedError("only the configuration model/VAE.py:52-56 builds is implemented: "
                                      "decoder_cat=True, mlp_normalization='batch', gconv_pooling='avg'")
        self.vocab = vocab
        self.embedding_dim = embedding_dim
        self.replace_all_latent = replace_latent
        self.use_angles = use_angles
        self.input_dim = input_dim
        self.gconv_num_layers = gconv_num_layers
        self.device = torch.device(device)
        self.num_objs = len(list(set(vocab["object_idx_to_name"])))
        self.num_preds = len(list(set(vocab["pred_idx_to_name"])))
        self.shapes = box_param_shapes(self.num_objs, self.num_preds, embedding_dim, gconv_num_layers=gconv_num_layers,
                                       input_dim=input_dim, use_angles=use_angles, residual=residual)
        self._sd: Dict[str, Tensor] = {}
        self._nets = None

    # ---- checkpoint surface ----
    def load_state_dict(self, sd, strict: bool = False):
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise RuntimeError(f"v2_box load_state_dict: missing {missing[:5]} (+{max(len(missing) - 5, 0)})")
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
            raise RuntimeError("v2_box: weights not loaded")
        if self.device.type != "cuda":
            raise L.CsError("v2_box: weights must be on the HIP device (no CPU path)")
        n = self.gconv_num_layers
        # Linear(input_dim -> box_e): the GEMM wants K as a multiple of 4, pad the box parameters with zeros
        self._box_pad = (self.input_dim + 3) // 4 * 4
        nets = dict(
            d3_emb=ops.pack_weight(sd["d3_embeddings.weight"], sd["d3_embeddings.bias"], cin_pad=self._box_pad),
            mean_var=_MLP(sd, "mean_var", 2, True), mean=_MLP(sd, "mean", 1, False), var=_MLP(sd, "var", 1, False),
            ec=GraphTripleConvNet(sd, "gconv_net_ec", n), dc=GraphTripleConvNet(sd, "gconv_net_dc", n),
            man=GraphTripleConvNet(sd, "gconv_net_manipulation", min(n, 5)),
            d3=_MLP(sd, "d3_net", 2, False),
        )
        if self.use_angles:
            nets.update(angle_mean_var=_MLP(sd, "angle_mean_var", 2, True), angle_mean=_MLP(sd, "angle_mean", 1, False),
                        angle_var=_MLP(sd, "angle_var", 1, False), angle=_MLP(sd, "angle_net", 2, False))
        self._nets = nets

    # ---- shared feature assembly: [clip | class embedding | extra...] in one buffer ----
    def _feats(self, objs, triples, text_feat, rel_feat, obj_table: str, pred_table: str, extra: List[Tensor],
               lead: Optional[Tensor] = None):
        dev = self.device
        triples = triples.to(dev)
        objs = objs.to(dev)
        s, p, o = triples[:, 0].contiguous(), triples[:, 1].contiguous(), triples[:, 2].contiguous()
        edges = torch.stack([s, o], dim=1).contiguous()
        O, T = objs.shape[0], triples.shape[0]
        e = self.embedding_dim
        text_feat = text_feat.to(device=dev, dtype=torch.float32)
        rel_feat = rel_feat.to(device=dev, dtype=torch.float32)
        cd = text_feat.shape[1]
        l0 = lead.shape[1] if lead is not None else 0
        width = l0 + cd + e + sum(x.shape[1] for x in extra)
        obj_vecs = torch.empty((O, width), dtype=torch.float32, device=dev)
        if lead is not None:
            obj_vecs[:, :l0].copy_(lead)
        obj_vecs[:, l0:l0 + cd].copy_(text_feat)
        ops.embedding(self._sd[obj_table], objs, out=obj_vecs[:, l0 + cd:l0 + cd + e])
        at = l0 + cd + e
        for x in extra:
            obj_vecs[:, at:at + x.shape[1]].copy_(x)
            at += x.shape[1]
        pe = self._sd[pred_table].shape[1]
        pred_vecs = torch.empty((T, cd + pe), dtype=torch.float32, device=dev)
        pred_vecs[:, :cd].copy_(rel_feat)
        ops.embedding(self._sd[pred_table], p, out=pred_vecs[:, cd:])
        return obj_vecs, pred_vecs, edges

    @torch.no_grad()
    def encoder(self, objs, triples, boxes_gt, attributes, enc_text_feat, enc_rel_feat, angles_gt=None):
        """VAEGAN_V2BOX.py:127-158 -> (mu, logvar), each (O, embedding_dim)."""
        if self._nets is None:
            self._build()
        dev, nets = self.device, self._nets
        boxes = torch.zeros((boxes_gt.shape[0], self._box_pad), dtype=torch.float32, device=dev)
        boxes[:, :self.input_dim].copy_(boxes_gt.to(device=dev, dtype=torch.float32))
        extra = [ops.linear(boxes, nets["d3_emb"])]
        if self.use_angles:
            extra.append(ops.embedding(self._sd["angle_embeddings.weight"], angles_gt.to(dev)))
        obj_vecs_, pred_vecs_, edges = self._feats(objs, triples, enc_text_feat, enc_rel_feat, "obj_embeddings_ec.weight",
                                                   "pred_embeddings_ec.weight", extra)
        obj_vecs_, _ = nets["ec"](obj_vecs_, pred_vecs_, edges)
        h = nets["mean_var"](obj_vecs_)
        mu, logvar = nets["mean"](h), nets["var"](h)
        if self.use_angles:
            ha = nets["angle_mean_var"](obj_vecs_)
            mu = torch.cat([mu, nets["angle_mean"](ha)], dim=1)
            logvar = torch.cat([logvar, nets["angle_var"](ha)], dim=1)
        return mu, logvar

    @torch.no_grad()
    def manipulate(self, z, objs, triples, dec_text_feat, dec_rel_feat, attributes=None):
        """VAEGAN_V2BOX.py:160-173: GCN over [z | clip | class embedding] -> (O, embedding_dim)."""
        if self._nets is None:
            self._build()
        man_z, pred_vecs_, edges = self._feats(objs, triples, dec_text_feat, dec_rel_feat, "obj_embeddings_dc.weight",
                                               "pred_embeddings_man_dc.weight", [],
                                               lead=z.to(device=self.device, dtype=torch.float32))
        man_z, _ = self._nets["man"](man_z, pred_vecs_, edges)
        return man_z

    @torch.no_grad()
    def decoder(self, z, objs, triples, dec_text_feat, dec_rel_feat, attributes=None, manipulate=False):
        """VAEGAN_V2BOX.py:175-200 (decoder_cat=True) -> d3_pred [, log-softmax angle logits]."""
        if self._nets is None:
            self._build()
        obj_vecs_, pred_vecs_, edges = self._feats(objs, triples, dec_text_feat, dec_rel_feat, "obj_embeddings_dc.weight",
                                                   "pred_embeddings_dc.weight",
                                                   [z.to(device=self.device, dtype=torch.float32)])
        obj_vecs_, _ = self._nets["dc"](obj_vecs_, pred_vecs_, edges)
        d3 = self._nets["d3"](obj_vecs_)
        if self.use_angles:
            return d3, ops.log_softmax(self._nets["angle"](obj_vecs_))
        return d3

    # ---- manipulation paths (host-side bookkeeping identical to the reference, device work above) ----
    def _insert_nodes(self, z, missing_nodes, distribution, width):
        nodes_added = []
        z = z.to(device=self.device, dtype=torch.float32)
        for i in range(len(missing_nodes)):
            ad_id = missing_nodes[i] + i
            nodes_added.append(ad_id)
            if distribution is not None:
                mu, cov = distribution
                new = torch.from_numpy(np.random.multivariate_normal(mu, cov, 1)).float()
            else:
                new = torch.zeros(1, width)
            z = torch.cat([z[:ad_id], new.to(self.device), z[ad_id:]], dim=0)
        return z, nodes_added

    def _keep(self, n, nodes_added, manipulated_nodes):
        keep = [1 if (i not in nodes_added and i not in manipulated_nodes) else 0 for i in range(n)]
        return torch.from_numpy(np.asarray(keep).reshape(-1, 1)).float().to(self.device)

    @torch.no_grad()
    def decoder_with_additions(self, z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes,
                               missing_nodes, manipulated_nodes, distribution=None):
        """VAEGAN_V2BOX.py:247-273."""
        z, nodes_added = self._insert_nodes(z, missing_nodes, distribution, z.shape[1])
        keep = self._keep(len(z), nodes_added, manipulated_nodes)
        return self.decoder(z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes), keep

    @torch.no_grad()
    def decoder_with_changes(self, z, dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes,
                             missing_nodes, manipulated_nodes, distribution=None):
        """VAEGAN_V2BOX.py:292-344."""
        z, nodes_added = self._insert_nodes(z, missing_nodes, distribution, z.shape[1])
        change = []
        for i in range(len(z)):
            if i not in nodes_added and i not in manipulated_nodes:
                change.append(np.zeros(self.embedding_dim))
            else:
                change.append(np.random.normal(0, 1, self.embedding_dim))
        change_repr = torch.from_numpy(np.stack(change, axis=0)).float().to(self.device)
        z_prime = self.manipulate(torch.cat([z, change_repr], dim=1), dec_objs, dec_triples, encoded_dec_text_feat,
                                  encoded_dec_rel_feat, attributes)
        if not self.replace_all_latent:
            z = z.clone()
            for t in sorted(nodes_added + list(manipulated_nodes)):     # untouched nodes keep their latent
                z[t] = z_prime[t]
        else:
            z = z_prime
        pred = self.decoder(z, dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)
        n = len(pred[0]) if self.use_angles else len(pred)
        return pred, self._keep(n, nodes_added, manipulated_nodes)

    @torch.no_grad()
    def sampleBoxes(self, mean_est, cov_est, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                    attributes=None, z: Optional[Tensor] = None):
        """VAEGAN_V2BOX.py:456-461.  Extension: `z` can be injected (the reference draws it from numpy's RNG)."""
        if z is None:
            z = torch.from_numpy(np.random.multivariate_normal(mean_est, cov_est, dec_objs.size(0))).float()
        return self.decoder(z, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)

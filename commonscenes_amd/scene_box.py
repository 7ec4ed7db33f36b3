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
from .scene import BoxVAEMixin, _np, GraphTripleConvNet, _MLP, _gcn_shapes, _mlp_shapes

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


class Sg2ScVAEModel(BoxVAEMixin):
    """Inference drop-in for `model.VAEGAN_V2BOX.Sg2ScVAEModel`."""
    _EC_NET = "gconv_net_ec"

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
        nets = self._box_nets()
        nets.update(dc=GraphTripleConvNet(sd, "gconv_net_dc", self.gconv_num_layers), d3=_MLP(sd, "d3_net", 2, False))
        if self.use_angles:
            nets["angle"] = _MLP(sd, "angle_net", 2, False)
        self._nets = nets

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
        z, nodes_added = self._changed_latent(z, dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                              attributes, missing_nodes, manipulated_nodes, distribution)
        pred = self.decoder(z, dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)
        n = len(pred[0]) if self.use_angles else len(pred)
        return pred, self._keep(n, nodes_added, manipulated_nodes)

    @torch.no_grad()
    def sampleBoxes(self, mean_est, cov_est, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                    attributes=None, z: Optional[Tensor] = None):
        """VAEGAN_V2BOX.py:456-461.  Extension: `z` can be injected (the reference draws it from numpy's RNG)."""
        if z is None:
            z = torch.from_numpy(np.random.multivariate_normal(_np(mean_est), _np(cov_est), dec_objs.size(0))).float()
        return self.decoder(z, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)

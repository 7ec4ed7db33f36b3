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


def _np(a):
    """torch tensors (collect_train_statistics returns the mean as one) -> numpy, same values, for numpy's RNG calls."""
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a


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
    box_e, ang_e = e * 3 // 4, e // 4         # use_angles=True (VAEGAN_V2FULL.py:45-49)
    S["obj_embeddings_ec.weight"] = (num_objs + 1, e)
    S["pred_embeddings_ec.weight"] = (num_preds, 2 * e)
    S["obj_embeddings_dc.weight"] = (num_objs + 1, e)
    S["pred_embeddings_dc.weight"] = (num_preds, 2 * e)
    S["pred_embeddings_man_dc.weight"] = (num_preds, 3 * e)
    S["d3_embeddings.weight"] = (box_e, num_box_params)
    S["d3_embeddings.bias"] = (box_e,)
    S["angle_embeddings.weight"] = (n_angle, ang_e)
    _mlp_shapes(S, "mean_var", [d, hid, 2 * e], True)
    _mlp_shapes(S, "mean", [2 * e, box_e], False)
    _mlp_shapes(S, "var", [2 * e, box_e], False)
    _mlp_shapes(S, "angle_mean_var", [d, hid, 2 * e], True)
    _mlp_shapes(S, "angle_mean", [2 * e, ang_e], False)
    _mlp_shapes(S, "angle_var", [2 * e, ang_e], False)
    _gcn_shapes(S, "gconv_net_ec_box", gconv_num_layers, d, d, hid)
    _gcn_shapes(S, "gconv_net_manipulation", min(gconv_num_layers, 5), 3 * e + clip_dim, 3 * e + clip_dim, hid,
                output_dim=e)
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

    def __call__(self, obj_vecs: Tensor, pred_vecs: Tensor, edges: Tensor, csr: Optional[Tensor] = None
                 ) -> Tuple[Tensor, Tensor]:
        """`csr`: the graph's CSR-by-destination index (ops.gcn_csr), shared by the layers of a net; None -> the
        index-free pooling kernel (same bits)."""
        H, Dout = self.H, self.Dout
        cur_t = ops.gcn_gather_cat(obj_vecs, pred_vecs, edges)            # [T, 2*Din + Dp]
        new_t = self.net1(cur_t)                                          # [T, 2H + Dout] = [s | p | o]
        if csr is not None:
            pooled = ops.gcn_segment_mean_csr(new_t, csr, obj_vecs.shape[0], H, H + Dout)
        else:
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
        csr = ops.gcn_csr(edges, obj_vecs.shape[0])         # the incidence lists depend on the edges only
        for g in self.gconvs:
            obj_vecs, pred_vecs = g(obj_vecs, pred_vecs, edges, csr)
        # the gather kernels skip out-of-range indices and raise a device flag; the reference's indexing
        # (graph.py:146-147, nn.Embedding) raises IndexError -- one read-back per GCN call keeps that behaviour
        ops.check_index_errors(obj_vecs.device, "scene graph (object / predicate ids, triple endpoints)")
        return obj_vecs, pred_vecs


GraphTripleConvNet2 = GraphTripleConvNet


# ------------------------------------------------------------------------------------------------
# box-VAE encoder / manipulation pieces shared by v2_box (VAEGAN_V2BOX.py) and v2_full (VAEGAN_V2FULL.py)
# ------------------------------------------------------------------------------------------------
class BoxVAEMixin:
    """Needs: self._sd, self.device, self.embedding_dim, self.use_angles, self.input_dim, self.gconv_num_layers,
    self.replace_all_latent, self._EC_NET (state_dict prefix of the encoder GCN)."""

    def _box_nets(self) -> dict:
        sd, n = self._sd, self.gconv_num_layers
        # Linear(input_dim -> box_e): the GEMM wants K as a multiple of 4, pad the box parameters with zeros
        self._box_pad = (self.input_dim + 3) // 4 * 4
        nets = dict(
            d3_emb=ops.pack_weight(sd["d3_embeddings.weight"], sd["d3_embeddings.bias"], cin_pad=self._box_pad),
            mean_var=_MLP(sd, "mean_var", 2, True), mean=_MLP(sd, "mean", 1, False), var=_MLP(sd, "var", 1, False),
            ec=GraphTripleConvNet(sd, self._EC_NET, n),
            man=GraphTripleConvNet(sd, "gconv_net_manipulation", min(n, 5)),
        )
        if self.use_angles:
            nets.update(angle_mean_var=_MLP(sd, "angle_mean_var", 2, True), angle_mean=_MLP(sd, "angle_mean", 1, False),
                        angle_var=_MLP(sd, "angle_var", 1, False))
        return nets

    def _feats(self, objs, triples, text_feat, rel_feat, obj_table: str, pred_table: str, extra: List[Tensor],
               lead: Optional[Tensor] = None):
        """[lead | clip | class embedding | extra...] node features and [clip | predicate embedding] edge features,
        each assembled in one buffer (embedding gathers write their column slice in place)."""
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
        """VAEGAN_V2BOX.py:127-158 / VAEGAN_V2FULL.py:185-218 -> (mu, logvar), each (O, embedding_dim)."""
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
        """VAEGAN_V2BOX.py:160-173 / VAEGAN_V2FULL.py:244-259: GCN over [z | clip | class embedding]."""
        if self._nets is None:
            self._build()
        man_z, pred_vecs_, edges = self._feats(objs, triples, dec_text_feat, dec_rel_feat, "obj_embeddings_dc.weight",
                                               "pred_embeddings_man_dc.weight", [],
                                               lead=z.to(device=self.device, dtype=torch.float32))
        man_z, _ = self._nets["man"](man_z, pred_vecs_, edges)
        return man_z

    # host-side bookkeeping identical to the reference (numpy's global RNG, call for call)
    def _insert_nodes(self, z, missing_nodes, distribution, width):
        nodes_added = []
        z = z.to(device=self.device, dtype=torch.float32)
        for i in range(len(missing_nodes)):
            ad_id = missing_nodes[i] + i
            nodes_added.append(ad_id)
            if distribution is not None:
                mu, cov = distribution
                new = torch.from_numpy(np.random.multivariate_normal(_np(mu), _np(cov), 1)).float()
            else:
                new = torch.zeros(1, width)
            z = torch.cat([z[:ad_id], new.to(self.device), z[ad_id:]], dim=0)
        return z, nodes_added

    def _keep(self, n, nodes_added, manipulated_nodes):
        keep = [1 if (i not in nodes_added and i not in manipulated_nodes) else 0 for i in range(n)]
        return torch.from_numpy(np.asarray(keep).reshape(-1, 1)).float().to(self.device)

    def _changed_latent(self, z, dec_objs, dec_triples, text_feat, rel_feat, attributes, missing_nodes,
                        manipulated_nodes, distribution):
        """first half of decoder_with_changes (VAEGAN_V2BOX.py:292-327 / VAEGAN_V2FULL.py:332-365)."""
        z, nodes_added = self._insert_nodes(z, missing_nodes, distribution, z.shape[1])
        change = []
        for i in range(len(z)):
            if i not in nodes_added and i not in manipulated_nodes:
                change.append(np.zeros(self.embedding_dim))
            else:
                change.append(np.random.normal(0, 1, self.embedding_dim))
        change_repr = torch.from_numpy(np.stack(change, axis=0)).float().to(self.device)
        z_prime = self.manipulate(torch.cat([z, change_repr], dim=1), dec_objs, dec_triples, text_feat, rel_feat,
                                  attributes)
        if not self.replace_all_latent:
            z = z.clone()
            for t in sorted(nodes_added + list(manipulated_nodes)):     # untouched nodes keep their latent
                z[t] = z_prime[t]
        else:
            z = z_prime
        return z, nodes_added

    @torch.no_grad()
    def collect_train_statistics(self, train_loader, with_points=False):
        """VAEGAN_V2FULL.py:700-760 / VAEGAN_V2BOX.py:463-520: mean and covariance of the encoder means over a loader
        yielding the reference's batch dicts (data['decoder'][objs|tripltes|boxes|text_feats|rel_feats])."""
        mean_cat = []
        for data in train_loader:
            if data == -1:
                continue
            try:
                dec = data["decoder"]
                objs, triples, tight_boxes = dec["objs"], dec["tripltes"], dec["boxes"]
                text, rel = dec["text_feats"], dec["rel_feats"]
            except Exception as e:      # the reference skips malformed batches the same way
                print("Exception", str(e))
                continue
            boxes = tight_boxes[:, :6]
            angles = tight_boxes[:, 6].long() - 1
            angles = torch.where(angles > 0, angles, torch.zeros_like(angles))
            mean, _ = self.encoder(objs, triples, boxes, None, text, rel, angles)
            mean_cat.append(mean.cpu().clone())
        mean_cat = torch.cat(mean_cat, dim=0)
        mean_est = torch.mean(mean_cat, dim=0, keepdim=True)
        cov_est = np.cov((mean_cat - mean_est).numpy().T)
        return mean_est[0], cov_est


# ------------------------------------------------------------------------------------------------
# the scene model
# ------------------------------------------------------------------------------------------------
class Sg2ScVAEModel(BoxVAEMixin):
    """Inference drop-in for the v2_full `Sg2ScVAEModel` (VAEGAN_V2FULL.py:17-760): `encoder`, `encoder_2`,
    `manipulate`, `decoder`, `decoder_with_changes`, `decoder_with_additions`, `sample`, `sampleBoxes`,
    `collect_train_statistics`, plus the checkpoint surface (`load_state_dict` / `state_dict`)."""
    _EC_NET = "gconv_net_ec_box"

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
        self.num_box_params = self.input_dim = num_box_params
        self.replace_all_latent = replace_latent
        self.gconv_num_layers = gconv_num_layers
        self.clip = clip
        if not use_angles:
            raise NotImplementedError("v2_full is evaluated with use_angles=True (scripts/eval_3dfront.py --with_angles)")
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
        # entries this implementation does not read (BatchNorm num_batches_tracked counters, training-only nets of other
        # builds, ...) are kept as they came so that state_dict() round-trips a reference checkpoint and the reference's
        # strict load_state_dict accepts what save() wrote
        self._extra_sd = OrderedDict((k, v) for k, v in sd.items() if k not in self.shapes and torch.is_tensor(v))
        self._nets = None
        return self

    def state_dict(self):
        out = OrderedDict(self._sd)
        out.update(getattr(self, "_extra_sd", {}))
        return out

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
            **self._box_nets(),
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

    def _shapes_for(self, z, objs, triples, text, rel, dec_sdfs, attributes, x_T, ddim_steps, sharded=None):
        """the gen_shape branch shared by sample / decoder_with_changes / decoder_with_additions
        (VAEGAN_V2FULL.py:309-318, 367-376, 606-616)."""
        dev = self.device
        un_rel_feat, rel_feat = self.encoder_2(z, objs, triples, text, rel, attributes)
        # nodes whose ground-truth SDF is non-zero (drops floor and _scene_); dec_sdfs stays on the host
        mask = torch.ne(dec_sdfs, torch.zeros_like(dec_sdfs[0]))
        ids = torch.unique(torch.where(mask)[0])
        ids_d = ids.to(dev)
        diff_dict = {"sdf": dec_sdfs[ids], "rel": rel_feat[ids_d], "uc": un_rel_feat[ids_d]}
        # under torch.distributed (one process per GPU) rel2shape shards the objects over the ranks: one broadcast of
        # rank 0's conditioning in, one all-gather of the SDFs out; `sharded` None = automatic
        return self.Diff.rel2shape(diff_dict, ddim_steps=ddim_steps, uc_scale=3., x_T=x_T, sharded=sharded)

    @torch.no_grad()
    def decoder_with_additions(self, z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, dec_sdfs,
                               attributes, missing_nodes, manipulated_nodes, distribution=None, gen_shape=False,
                               x_T: Optional[Tensor] = None, ddim_steps: int = 100):
        """VAEGAN_V2FULL.py:291-330 -> (boxes[, angles]), gen_sdf | None, keep."""
        z, nodes_added = self._insert_nodes(z, missing_nodes, distribution, z.shape[1])
        text = encoded_dec_text_feat.to(device=self.device, dtype=torch.float32)
        rel = encoded_dec_rel_feat.to(device=self.device, dtype=torch.float32)
        gen_sdf = self._shapes_for(z, objs, triples, text, rel, dec_sdfs, attributes, x_T, ddim_steps) if gen_shape else None
        pred = self.decoder(z, objs, triples, text, rel, attributes)
        return pred, gen_sdf, self._keep(len(z), nodes_added, manipulated_nodes)

    @torch.no_grad()
    def decoder_with_changes(self, z, dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat, dec_sdfs,
                             attributes, missing_nodes, manipulated_nodes, distribution=None, gen_shape=False,
                             x_T: Optional[Tensor] = None, ddim_steps: int = 100):
        """VAEGAN_V2FULL.py:332-396 -> (boxes[, angles]), gen_sdf | None, keep."""
        text = encoded_dec_text_feat.to(device=self.device, dtype=torch.float32)
        rel = encoded_dec_rel_feat.to(device=self.device, dtype=torch.float32)
        z, nodes_added = self._changed_latent(z, dec_objs, dec_triples, text, rel, attributes, missing_nodes,
                                              manipulated_nodes, distribution)
        gen_sdf = (self._shapes_for(z, dec_objs, dec_triples, text, rel, dec_sdfs, attributes, x_T, ddim_steps)
                   if gen_shape else None)
        pred = self.decoder(z, dec_objs, dec_triples, text, rel, attributes)
        n = len(pred[0]) if self.use_angles else len(pred)
        return pred, gen_sdf, self._keep(n, nodes_added, manipulated_nodes)

    @torch.no_grad()
    def sampleBoxes(self, mean_est, cov_est, dec_objs, dec_triplets, encoded_dec_text_feat=None,
                    encoded_dec_rel_feat=None, attributes=None, z: Optional[Tensor] = None):
        """VAEGAN_V2FULL.py:593-598 (whose own decoder call lacks the CLIP features; they are required here)."""
        if encoded_dec_text_feat is None or encoded_dec_rel_feat is None:
            raise ValueError("sampleBoxes needs the CLIP text / relation features (clip=True)")
        if z is None:
            z = torch.from_numpy(np.random.multivariate_normal(_np(mean_est), _np(cov_est), dec_objs.size(0))).float()
        return self.decoder(z, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)

    @torch.no_grad()
    def sample(self, point_classes_idx, mean_est, cov_est, dec_objs, dec_triplets, dec_sdfs,
               encoded_dec_text_feat, encoded_dec_rel_feat, attributes=None, gen_shape=False,
               z: Optional[Tensor] = None, x_T: Optional[Tensor] = None, ddim_steps: int = 100,
               sharded: Optional[bool] = None):
        """VAEGAN_V2FULL.py:600-618.  Extensions: `z`, `x_T` and `ddim_steps` can be injected (the reference
        draws z from numpy's global RNG and x_T from a time-seeded torch RNG); `sharded` (None = automatic when a
        torch.distributed process group exists) splits the shaped objects over the ranks -- every rank calls sample()
        with the same arguments and every rank gets all the SDFs back."""
        dev = self.device
        if z is None:
            z = torch.from_numpy(np.random.multivariate_normal(_np(mean_est), _np(cov_est), dec_objs.size(0))).float()
        z = z.to(dev)
        text = encoded_dec_text_feat.to(device=dev, dtype=torch.float32)
        rel = encoded_dec_rel_feat.to(device=dev, dtype=torch.float32)
        gen_sdf = (self._shapes_for(z, dec_objs, dec_triplets, text, rel, dec_sdfs, attributes, x_T, ddim_steps, sharded)
                   if gen_shape else None)
        return self.decoder(z, dec_objs, dec_triplets, text, rel, attributes), gen_sdf

    @torch.no_grad()
    def sample_many(self, scenes, gen_shape: bool = True, ddim_steps: int = 100, launch_B: Optional[int] = None):
        """Extension: `sample` for SEVERAL scenes with ONE coalesced sampler + decode.

        scenes: a list of dicts with sample()'s arguments -- dec_objs, dec_triplets, dec_sdfs, encoded_dec_text_feat,
        encoded_dec_rel_feat, and optionally attributes, z, x_T (point_classes_idx / mean_est / cov_est as for sample():
        z is drawn from (mean_est, cov_est) when absent).  Each scene's graph runs through encoder_2 and the layout decoder
        SEPARATELY (graphs must not mix: the GCN pools over a scene's own triples, VAEGAN_V2FULL.py:220-242); the shaped
        objects of all scenes then go through Diff.rel2shape_many -- per-scene shared x_T kept -- which is where the
        reference's scene-by-scene loop (scripts/eval_3dfront.py:484-513) leaves the chip mostly idle.
        Returns [(boxes_or_(boxes, angles), gen_sdf | None)] in scene order, exactly what sample() returns per scene."""
        dev = self.device
        prepared, datas, x_Ts = [], [], []
        for sc in scenes:
            objs, triples = sc["dec_objs"], sc["dec_triplets"]
            z = sc.get("z")
            if z is None:
                z = torch.from_numpy(np.random.multivariate_normal(_np(sc["mean_est"]), _np(sc["cov_est"]),
                                                                   objs.size(0))).float()
            z = z.to(dev)
            text = sc["encoded_dec_text_feat"].to(device=dev, dtype=torch.float32)
            rel = sc["encoded_dec_rel_feat"].to(device=dev, dtype=torch.float32)
            attributes = sc.get("attributes")
            prepared.append((z, objs, triples, text, rel, attributes))
            if gen_shape:
                un_rel_feat, rel_feat = self.encoder_2(z, objs, triples, text, rel, attributes)
                dec_sdfs = sc["dec_sdfs"]
                mask = torch.ne(dec_sdfs, torch.zeros_like(dec_sdfs[0]))
                ids = torch.unique(torch.where(mask)[0])
                ids_d = ids.to(dev)
                datas.append({"sdf": dec_sdfs[ids], "rel": rel_feat[ids_d], "uc": un_rel_feat[ids_d]})
                x_Ts.append(sc.get("x_T"))
        gens = [None] * len(scenes)
        if gen_shape and scenes:
            gens = self.Diff.rel2shape_many(datas, ddim_steps=ddim_steps, uc_scale=3., x_Ts=x_Ts, launch_B=launch_B)
        return [(self.decoder(z, objs, triples, text, rel, attributes), gens[i])
                for i, (z, objs, triples, text, rel, attributes) in enumerate(prepared)]

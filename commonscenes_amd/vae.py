"""The evaluation scripts' entry object: `VAE` (model/VAE.py:18-340) for the two v2 network types, over the HIP
models of this package -- what scripts/eval_3dfront.py and scripts/eval_3dfront_manivis.py construct and call
(`load_networks`, `compute_statistics`, `sample_box_and_shape`, `encode_box`, `decoder_with_changes_boxes_and_shape`,
`decoder_with_additions_boxes_and_shape`, ...).  Inference only; the v1 types (DeepSDF retrieval / AtlasNet,
outside the shape-branch path) raise NotImplementedError.
"""
from __future__ import annotations

import os
import pickle
from typing import Optional

import torch

from .scene import Sg2ScVAEModel as v2_full
from .scene_box import Sg2ScVAEModel as v2_box


class VAE:
    def __init__(self, root="../GT", type="v1_box", diff_opt="../config/v2_full.yaml", vocab=None, replace_latent=False,
                 with_changes=True, distribution_before=True, residual=False, gconv_pooling="avg", with_angles=False,
                 num_box_params=6, lr_full=None, deepsdf=False, clip=True, with_E2=True, device="cuda",
                 resolve_dir: Optional[str] = None):
        assert type in ["v1_box", "v1_full", "v2_box", "v2_full"], "{} is not included".format(type)
        if type in ("v1_box", "v1_full"):
            raise NotImplementedError(f"network type {type!r}: only the v2 models are on the MI355X path")
        self.type_ = type
        self.vocab = vocab
        self.with_angles = with_angles
        self.epoch = 0
        self.counter = 0
        self.diff_opt = diff_opt
        if type == "v2_box":                                                    # model/VAE.py:52-56
            assert replace_latent is not None
            self.vae_box = v2_box(vocab, embedding_dim=64, decoder_cat=True, mlp_normalization="batch",
                                  input_dim=num_box_params, replace_latent=replace_latent, use_angles=with_angles,
                                  residual=residual, gconv_pooling=gconv_pooling, gconv_num_layers=5, device=device)
        else:                                                                   # model/VAE.py:57-62
            assert distribution_before is not None and replace_latent is not None and with_changes is not None
            self.vae_v2 = v2_full(vocab, diff_opt, diffusion_bs=16, embedding_dim=64, decoder_cat=True,
                                  mlp_normalization="batch", gconv_num_layers=5, use_angles=with_angles,
                                  distribution_before=distribution_before, use_E2=with_E2, replace_latent=replace_latent,
                                  num_box_params=num_box_params, residual=residual, clip=clip, device=device,
                                  resolve_dir=resolve_dir)

    # nn.Module-like no-ops the scripts call
    def eval(self):
        return self

    def cuda(self):
        return self

    def set_cuda(self):
        return None

    # ---- checkpoints (model/VAE.py:104-165) ----
    def load_networks(self, exp, epoch, strict=True, restart_optim=False):
        if self.type_ == "v2_box":
            self.vae_box.load_state_dict(torch.load(os.path.join(exp, "checkpoint", "model_box_{}.pth".format(epoch)),
                                                    map_location="cpu"), strict=strict)
            return
        ckpt = torch.load(os.path.join(exp, "checkpoint", "model{}.pth".format(epoch)), map_location="cpu")
        vq_sd, df_sd = ckpt.pop("vqvae"), ckpt.pop("df")
        ckpt.pop("opt", None)                       # optimiser state: training only
        self.epoch = ckpt.pop("epoch", self.epoch)
        self.counter = ckpt.pop("counter", self.counter)
        self.vae_v2.load_state_dict(ckpt, strict=strict)
        self.vae_v2.Diff.vqvae.load_state_dict(vq_sd)
        self.vae_v2.Diff.df.load_state_dict(df_sd)
        self.vae_v2.Diff.df_module = self.vae_v2.Diff.df
        self.vae_v2.Diff.vqvae_module = self.vae_v2.Diff.vqvae

    def save(self, exp, outf, epoch, counter=None):
        if self.type_ == "v2_box":
            torch.save(self.vae_box.state_dict(), os.path.join(exp, outf, "model_box_{}.pth".format(epoch)))
        else:                                       # VAEGAN_V2FULL.py:687-699 layout
            sd = dict(self.vae_v2.state_dict())
            sd.update(self.vae_v2.Diff.state_dict())
            # the reference's load_networks pops 'opt' unconditionally (model/VAE.py:118): an inference-only build has no
            # optimiser state, so the entry is an empty placeholder
            sd.update(opt={}, epoch=epoch, counter=counter)
            torch.save(sd, os.path.join(exp, outf, "model{}.pth".format(epoch)))

    def compute_statistics(self, exp, epoch, stats_dataloader, force=False):
        """model/VAE.py:167-197: mean / covariance of the encoder means over the training set, cached as a pickle."""
        box = self.type_ == "v2_box"
        f = os.path.join(exp, "checkpoint", ("model_stats_box_{}.pkl" if box else "model_stats_{}.pkl").format(epoch))
        if os.path.exists(f) and not force:
            with open(f, "rb") as fh:
                stats = pickle.load(fh)
        else:
            stats = list((self.vae_box if box else self.vae_v2).collect_train_statistics(stats_dataloader))
            with open(f, "wb") as fh:
                pickle.dump(stats, fh)
        if box:
            self.mean_est_box, self.cov_est_box = stats[0], stats[1]
        else:
            self.mean_est, self.cov_est = stats[0], stats[1]

    # ---- encode ----
    def encode_box(self, objs, triples, encoded_enc_text_feat, encoded_enc_rel_feat, boxes, angles=None, attributes=None):
        m = self.vae_box if self.type_ == "v2_box" else self.vae_v2
        return m.encoder(objs, triples, boxes, attributes, encoded_enc_text_feat, encoded_enc_rel_feat, angles)

    def encode_box_and_shape(self, objs, triples, encoded_enc_text_feat, encoded_enc_rel_feat, feats, boxes, angles=None,
                             attributes=None):
        if not self.with_angles:
            angles = None
        return self.encode_box(objs, triples, encoded_enc_text_feat, encoded_enc_rel_feat, boxes, angles,
                               attributes), (None, None)

    # ---- sample ----
    def sample_box_and_shape(self, point_classes_idx, dec_objs, dec_triplets, dec_sdfs, encoded_dec_text_feat,
                             encoded_dec_rel_feat, attributes=None, gen_shape=False, **inject):
        """model/VAE.py:286-294.  `inject` forwards the z / x_T / ddim_steps extensions of Sg2ScVAEModel.sample."""
        if self.type_ == "v2_full":
            return self.vae_v2.sample(point_classes_idx, self.mean_est, self.cov_est, dec_objs, dec_triplets, dec_sdfs,
                                      encoded_dec_text_feat, encoded_dec_rel_feat, attributes, gen_shape=gen_shape,
                                      **inject)
        return self.sample_box(dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, attributes), None

    def sample_box_and_shape_many(self, scenes, gen_shape=True, ddim_steps: int = 100, launch_B=None):
        """Extension: sample_box_and_shape for a LIST of scenes -- each a dict with its keyword arguments (dec_objs,
        dec_triplets, dec_sdfs, encoded_dec_text_feat, encoded_dec_rel_feat[, attributes, z, x_T]) -- with every scene's
        graph encoded and laid out on its own and ONE coalesced DDIM sampler + VQ decode over all scenes' shaped objects
        (Sg2ScVAEModel.sample_many).  Returns [(boxes_or_(boxes, angles), gen_sdf | None)] in scene order.  The default
        per-scene call (model/VAE.py:286-294) is unchanged."""
        if self.type_ != "v2_full":
            return [(self.sample_box(sc["dec_objs"], sc["dec_triplets"], sc["encoded_dec_text_feat"],
                                     sc["encoded_dec_rel_feat"], sc.get("attributes")), None) for sc in scenes]
        full = [dict(sc, mean_est=sc.get("mean_est", self.mean_est), cov_est=sc.get("cov_est", self.cov_est))
                for sc in scenes]
        return self.vae_v2.sample_many(full, gen_shape=gen_shape, ddim_steps=ddim_steps, launch_B=launch_B)

    def sample_box(self, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, attributes=None):
        if self.type_ == "v2_box":
            return self.vae_box.sampleBoxes(self.mean_est_box, self.cov_est_box, dec_objs, dec_triplets,
                                            encoded_dec_text_feat, encoded_dec_rel_feat, attributes)
        return self.vae_v2.sampleBoxes(self.mean_est, self.cov_est, dec_objs, dec_triplets, encoded_dec_text_feat,
                                       encoded_dec_rel_feat, attributes)

    def sample_shape(self, point_classes_idx, dec_objs, dec_triplets, attributes=None):
        return None                                  # v1_full only in the reference (model/VAE.py:329-331)

    # ---- manipulation ----
    def decoder_with_changes_boxes_and_shape(self, z_box, z_shape, objs, triples, encoded_dec_text_feat,
                                             encoded_dec_rel_feat, dec_sdfs, attributes, missing_nodes, manipulated_nodes,
                                             box_data=None, gen_shape=False, **inject):
        if self.type_ == "v2_box":
            boxes, keep = self.decoder_with_changes_boxes(z_box, objs, triples, encoded_dec_text_feat,
                                                          encoded_dec_rel_feat, attributes, missing_nodes,
                                                          manipulated_nodes)
            return boxes, None, keep
        return self.vae_v2.decoder_with_changes(z_box, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                dec_sdfs, attributes, missing_nodes, manipulated_nodes,
                                                gen_shape=gen_shape, **inject)

    def decoder_with_changes_boxes(self, z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes,
                                   missing_nodes, manipulated_nodes):
        if self.type_ != "v2_box":
            raise NotImplementedError("decoder_with_changes_boxes is a box-model call (model/VAE.py:207-211)")
        return self.vae_box.decoder_with_changes(z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                 attributes, missing_nodes, manipulated_nodes)

    def decoder_with_additions_boxes_and_shape(self, z_box, z_shape, objs, triples, encoded_dec_text_feat,
                                               encoded_dec_rel_feat, dec_sdfs, attributes, missing_nodes,
                                               manipulated_nodes, gen_shape=False, **inject):
        if self.type_ == "v2_box":
            boxes, _, keep = self.decoder_with_additions_boxs(z_box, objs, triples, encoded_dec_text_feat,
                                                              encoded_dec_rel_feat, attributes, missing_nodes,
                                                              manipulated_nodes)
            return boxes, None, keep
        return self.vae_v2.decoder_with_additions(z_box, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                  dec_sdfs, attributes, missing_nodes, manipulated_nodes,
                                                  gen_shape=gen_shape, **inject)

    def decoder_with_additions_boxs(self, z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes,
                                    missing_nodes, manipulated_nodes):
        if self.type_ != "v2_box":
            raise NotImplementedError("decoder_with_additions_boxs is a box-model call (model/VAE.py:251-259)")
        boxes, keep = self.vae_box.decoder_with_additions(z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                          attributes, missing_nodes, manipulated_nodes,
                                                          (self.mean_est_box, self.cov_est_box))
        return boxes, None, keep

    def decoder_boxes(self, z, objs, triples, attributes, encoded_dec_text_feat=None, encoded_dec_rel_feat=None):
        """model/VAE.py:226-233 (whose v2 call lacks the CLIP features the v2 decoder needs; required here)."""
        if self.type_ != "v2_box":
            raise NotImplementedError("decoder_boxes is a box-model call")
        out = self.vae_box.decoder(z, objs, triples, encoded_dec_text_feat, encoded_dec_rel_feat, attributes)
        return out if self.with_angles else (out, None)

"""Deterministic synthetic weights and inputs (no checkpoints or datasets are reachable offline).

Values come from a counter-based integer hash (FNV-1a of the tensor name -> SplitMix64 per
element) mapped to uniform [-1, 1) with exact integer->float arithmetic, so the same tensors are
regenerated bit-identically in the build container (golden generation against the reference) and on
the GPU box (parity tests, bench) without shipping 1.65 GB of weights.  Zero-initialised reference
tensors are randomised like every other tensor, otherwise parity would be vacuous.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(name: str, n: int, seed: int = 111) -> np.ndarray:
    """n values in [-1, 1), float64, each exactly representable in fp32."""
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    out = np.empty(n, dtype=np.float64)
    CH = 1 << 22
    with np.errstate(over="ignore"):
        for s in range(0, n, CH):
            e = min(n, s + CH)
            z = np.arange(s, e, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            k = (z >> np.uint64(40)).astype(np.int64)          # 24 bits
            out[s:e] = (2 * k - (1 << 24)).astype(np.float64) / float(1 << 24)
    return out


def tensor(name: str, shape: Tuple[int, ...], scale: float = 1.0, offset: float = 0.0, seed: int = 111) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(name, n, seed) * scale + offset
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def _fan_in(shape: Tuple[int, ...]) -> int:
    f = 1
    for s in shape[1:]:
        f *= s
    return max(f, 1)


def _entry_params(name: str, shape, shapes, gain: float):
    """(scale, offset) of the uniform law for one state_dict entry; None for integer bookkeeping entries."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return None
    if leaf == "running_mean":
        return 0.2, 0.0
    if leaf == "running_var":
        return 0.5, 1.0
    if "embedding" in name and leaf == "weight" and len(shape) == 2 and "quantize" in name:
        return 1.5, 0.0                                   # VQ codebook spread over the latent range
    if "embeddings" in name and leaf == "weight":
        return 1.0, 0.0
    if leaf == "weight" and len(shape) == 1:
        return 0.2, 1.0                                   # norm scale
    if leaf == "bias" and (".norm" in name or "in_layers.0" in name or "out_layers.0" in name
                           or name.endswith("out.0.bias") or _is_bn(name, shapes)):
        return 0.1, 0.0                                   # norm shift
    if leaf == "weight":
        return float(gain / np.sqrt(_fan_in(shape))), 0.0
    if leaf == "bias":
        wshape = shapes.get(name[:-4] + "weight", (shape[0], 1))
        return float(1.0 / np.sqrt(_fan_in(wshape))), 0.0
    return 1.0, 0.0


def tensor_device(name: str, shape: Tuple[int, ...], scale: float = 1.0, offset: float = 0.0, seed: int = 111,
                  device="cuda") -> torch.Tensor:
    """Same values as `tensor`, generated on the HIP device by cs_synth_fill (bit-identical)."""
    import ctypes as C
    from . import lib as L
    n = int(np.prod(shape)) if len(shape) else 1
    out = torch.empty(shape, dtype=torch.float32, device=device)
    base = (_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.load().cs_synth_fill(out.data_ptr(), n, base, float(scale), float(offset), stream), "cs_synth_fill")
    return out


def synth_state_dict(shapes: "Dict[str, Tuple[int, ...]]", seed: int = 111, gain: float = 3.0 ** 0.5,
                     device: str = "cpu") -> "OrderedDict[str, torch.Tensor]":
    """Weights for a {name: shape} table.  Conv/Linear weights ~ U(-a, a) with a = gain/sqrt(fan_in)
    (variance preserving), biases ~ U(-1, 1)/sqrt(fan_in) of their layer, norm scales 1 + 0.2u,
    norm biases 0.1u, BatchNorm running_mean 0.2u, running_var 1 + 0.5u, embeddings U(-1, 1).
    device='cuda' generates the identical values with the cs_synth_fill kernel."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    on_dev = str(device).startswith("cuda")
    for name, shape in shapes.items():
        pr = _entry_params(name, shape, shapes, gain)
        if pr is None:
            sd[name] = torch.tensor(1, dtype=torch.long)
        elif on_dev:
            sd[name] = tensor_device(name, tuple(shape), pr[0], pr[1], seed, device)
        else:
            sd[name] = tensor(name, tuple(shape), pr[0], pr[1], seed)
    return sd


def _is_bn(name: str, shapes) -> bool:
    return (name[:-4] + "running_mean") in shapes


def gaussian_like(name: str, shape: Tuple[int, ...], seed: int = 111, scale: float = 1.0) -> torch.Tensor:
    """Approximately N(0, scale^2) samples (sum of 12 hash uniforms; every float64 operation below is
    exact or a single correctly-rounded IEEE op, so the result is platform independent)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for k in range(12):
        acc += hash_uniform(f"{name}#{k}", n, seed)
    # each term is uniform[-1,1): variance 1/3, sum variance 4 -> divide by 2
    return torch.from_numpy((acc / 2.0 * float(scale)).astype(np.float32).reshape(shape))


def random_scene_graph(n_objects: int, seed: int = 111, n_obj_classes: int = 35, n_pred_classes: int = 16):
    """Synthetic SG-FRONT-like graph: O = n_objects + 2 nodes (last two: floor, _scene_),
    an `in` edge (predicate 0 here) from every node to `_scene_`, plus 2-6 typed edges per object.
    Returns objs (O,), triples (T,3) [s,p,o] int64, text_feats (O,512), rel_feats (T,512), z (O,64)."""
    O = n_objects + 2
    u = lambda tag, n: hash_uniform(f"graph:{tag}", n, seed)
    objs = ((u("objs", O) * 0.5 + 0.5) * (n_obj_classes - 2)).astype(np.int64) + 1
    objs[-2] = n_obj_classes - 1          # floor
    objs[-1] = 0                          # _scene_
    triples = []
    ne = ((u("nedges", n_objects) * 0.5 + 0.5) * 5).astype(np.int64) + 2
    tgt = u("targets", int(ne.sum()) + 1)
    prd = u("preds", int(ne.sum()) + 1)
    k = 0
    for i in range(n_objects):
        for _ in range(int(ne[i])):
            j = int((tgt[k] * 0.5 + 0.5) * (O - 1)) % (O - 1)
            if j == i:
                j = (j + 1) % (O - 1)
            p = int((prd[k] * 0.5 + 0.5) * (n_pred_classes - 1)) % (n_pred_classes - 1) + 1
            triples.append((i, p, j))
            k += 1
    for i in range(O - 1):
        triples.append((i, 0, O - 1))
    triples = np.asarray(triples, dtype=np.int64)
    T = triples.shape[0]

    def clip_like(tag, n):       # CLIP-like scale: ||f|| ~ 10 (no data-dependent normalisation:
        return gaussian_like(f"graph:{tag}", (n, 512), seed, scale=10.0 / float(np.sqrt(512.0)))  # keeps it exact)

    return dict(objs=torch.from_numpy(objs), triples=torch.from_numpy(triples),
                text_feats=clip_like("text", O), rel_feats=clip_like("rel", T),
                z=gaussian_like("graph:z", (O, 64), seed))

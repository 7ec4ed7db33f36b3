"""ORACLE -- test infrastructure only.  Channels-last CPU references (plain PyTorch fp32) for the
single-op C-ABI entries, used by tests/test_ops_gpu.py.  Each mirrors the ATen call the reference
makes (cited per function); nothing here is imported by the product path."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def conv_ndhwc(x, w, bias=None, stride=(1, 1, 1), up=(0, 0, 0), rowvec=None, res=None, act=None,
               scale=None, shift=None):
    """x [nb,d,h,w,c] ; w torch layout (cout,cin,k,k,k) or (cout,cin).  Nearest upsample by 2**up
    first (openai_model_3d.py:150-153, vqvae_modules.py:35-39), then conv pad k//2."""
    xn = x.permute(0, 4, 1, 2, 3)
    if any(up):
        xn = F.interpolate(xn, scale_factor=tuple(float(1 << u) for u in up), mode="nearest")
    if w.dim() == 2:
        w = w[:, :, None, None, None]
    k = w.shape[-1]
    y = F.conv3d(xn, w, bias, stride=stride, padding=k // 2)
    y = y.permute(0, 2, 3, 4, 1)
    if scale is not None:
        y = y * scale + shift
    if rowvec is not None:
        y = y + rowvec[:, None, None, None, :]
    if act == "relu":
        y = F.relu(y)
    elif act == "silu":
        y = F.silu(y)
    elif act == "gelu":
        y = F.gelu(y)
    if res is not None:
        y = y + res
    return y.contiguous()


def groupnorm_ndhwc(x, gamma, beta, groups, eps, act=None):
    xn = x.permute(0, 4, 1, 2, 3)
    y = F.group_norm(xn, groups, gamma, beta, eps)
    if act == "silu":
        y = F.silu(y)
    elif act == "gelu":
        y = F.gelu(y)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def attention(q, k, v, heads, scale):
    """attention.py:201-218 on [nb, n, heads*dh] tensors."""
    b, nq, c = q.shape
    dh = c // heads
    sp = lambda t: t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], dh)
    qh, kh, vh = sp(q), sp(k), sp(v)
    sim = torch.einsum("bid,bjd->bij", qh, kh) * scale
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, vh)
    return out.reshape(b, heads, nq, dh).permute(0, 2, 1, 3).reshape(b, nq, c)


def geglu(x):
    a, g = x.chunk(2, dim=-1)
    return a * F.gelu(g)


def ddim_update(x, eps, a_t, a_prev, sigma_t, sqrt_1m, scale, cfg):
    """samplers/ddim.py:206-243 with python-float coefficients turned into fp32 tensors."""
    if cfg:
        e_uc, e_c = eps.chunk(2)
        e = e_uc + scale * (e_c - e_uc)
    else:
        e = eps
    shp = (x.shape[0],) + (1,) * (x.dim() - 1)
    a_t = torch.full(shp, a_t)
    a_prev = torch.full(shp, a_prev)
    sigma_t = torch.full(shp, sigma_t)
    sqrt_1m = torch.full(shp, sqrt_1m)
    pred = (x - sqrt_1m * e) / a_t.sqrt()
    dirx = (1.0 - a_prev - sigma_t ** 2).sqrt() * e
    return a_prev.sqrt() * pred + dirx, pred


def vq(z_rows, emb):
    """quantizer.py:76-84 on flattened rows."""
    d = torch.sum(z_rows ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * torch.einsum(
        "bd,dn->bn", z_rows, emb.t())
    idx = torch.argmin(d, dim=1)
    return idx, emb[idx], d


def gcn_pool(new_t, edges, n_obj, H, off_o):
    """graph.py:161-199 (avg pooling)."""
    s_idx, o_idx = edges[:, 0].contiguous(), edges[:, 1].contiguous()
    new_s, new_o = new_t[:, :H], new_t[:, off_o:off_o + H]
    pooled = torch.zeros(n_obj, H)
    pooled = pooled.scatter_add(0, s_idx.view(-1, 1).expand_as(new_s), new_s)
    pooled = pooled.scatter_add(0, o_idx.view(-1, 1).expand_as(new_o), new_o)
    cnt = torch.zeros(n_obj)
    ones = torch.ones(edges.shape[0])
    cnt = cnt.scatter_add(0, s_idx, ones).scatter_add(0, o_idx, ones).clamp(min=1)
    return pooled / cnt.view(-1, 1)


def chamfer_nm(xyz1, xyz2):
    """extension/chamfer.cu:11-75 (NmDistanceKernel), one direction: for each xyz1[b, j] the squared distance to its
    nearest xyz2[b, k] and the first minimising k.  fp32 arithmetic in the reference's operation order
    ((dx*dx + dy*dy) + dz*dz, no fused multiply-add on this side).  PARITY UNPINNED against the CUDA build of the
    reference (it cannot run in this image); the definition is the textbook one and the reference's own
    extension/test.py checks it the same way (against a python loop)."""
    import numpy as np
    a = np.asarray(xyz1, dtype=np.float32)
    b = np.asarray(xyz2, dtype=np.float32)
    B, n, _ = a.shape
    dist = np.empty((B, n), dtype=np.float32)
    idx = np.empty((B, n), dtype=np.int32)
    for i in range(B):
        dx = b[i, None, :, 0] - a[i, :, None, 0]
        dy = b[i, None, :, 1] - a[i, :, None, 1]
        dz = b[i, None, :, 2] - a[i, :, None, 2]
        d = (dx * dx + dy * dy) + dz * dz
        idx[i] = np.argmin(d, axis=1).astype(np.int32)
        dist[i] = d[np.arange(n), idx[i]]
    return dist, idx

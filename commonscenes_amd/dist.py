"""Object sharding of the shape sampler across the node's GPUs (one process per GPU, RCCL over xGMI).

After conditioning every object's DDIM trajectory and decode depend only on its own (x_T, c_i, uc_i)
(SURVEY 8e), so the path shards by object with NO per-step collective:
  in : one broadcast of the packed [x_T | uc | c] buffer from the rank that ran the scene-graph GCN
       (49 KB + 10 KB per object);
  out: one all-gather of the decoded SDFs (1 MiB per object).
xGMI is point-to-point, so both are sized to be single large messages rather than per-object sends.
The reference has no counterpart (its DDP wrappers are dormant and no process group is ever created,
SURVEY 2.3); `backend="nccl"` is RCCL on ROCm, and the same code runs on `gloo` for the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def _force() -> bool:
    """TEST-ONLY switch (CS_DIST_FORCE_COLLECTIVES=1): issue the collectives even in a one-rank group.  Every function
    below returns early at world size 1, and every multi-rank test on a one-GPU box pairs device tensors with gloo (RCCL
    needs one GPU per rank) -- so without this the device-to-device `backend="nccl"` lines would first execute on the
    8-GPU node.  tests/test_rccl_single_rank_gpu.py drives them through a one-rank RCCL group."""
    import os
    return os.environ.get("CS_DIST_FORCE_COLLECTIVES", "") == "1" and dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (total % world) ranks get one extra object."""
    q, r = divmod(total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_conditioning(x_T: Tensor, uc: Tensor, c: Tensor) -> Tensor:
    """[x_T (flattened) | uc (B,D) | c (B,D)] as one fp32 vector -> one broadcast instead of three."""
    B = c.shape[0]
    return torch.cat([x_T.reshape(-1).float(), uc.reshape(B, -1).float().reshape(-1),
                      c.reshape(B, -1).float().reshape(-1)])


def unpack_conditioning(buf: Tensor, n_obj: int, latent_shape=(3, 16, 16, 16), ctx_dim: int = 1280,
                        cond_shape: Optional[Tuple[int, ...]] = None):
    """cond_shape: per-object shape of uc / c -- (1, 1280) for the crossattn family (default), (1, 16, 16, 16) for the
    concat family's condition volume; its element count is ctx_dim."""
    n_lat = 1
    for v in latent_shape:
        n_lat *= v
    cs = tuple(cond_shape) if cond_shape is not None else (1, ctx_dim)
    x_T = buf[:n_lat].reshape(1, *latent_shape)
    uc = buf[n_lat:n_lat + n_obj * ctx_dim].reshape(n_obj, *cs)
    c = buf[n_lat + n_obj * ctx_dim:n_lat + 2 * n_obj * ctx_dim].reshape(n_obj, *cs)
    return x_T, uc, c


def broadcast_conditioning(x_T: Optional[Tensor], uc: Optional[Tensor], c: Optional[Tensor], n_obj: int,
                           device, src: int = 0, latent_shape=(3, 16, 16, 16), ctx_dim: int = 1280,
                           cond_shape: Optional[Tuple[int, ...]] = None):
    """Every rank returns rank `src`'s (x_T, uc, c) for ALL objects; the other ranks' arguments are ignored (they may
    pass None).  One collective over one packed fp32 buffer (49 KB + 10 KB per object)."""
    rank, ws = world()
    n_lat = 1
    for v in latent_shape:
        n_lat *= v
    if rank == src:
        buf = pack_conditioning(x_T, uc, c).to(device)
        if buf.numel() != n_lat + 2 * n_obj * ctx_dim:
            raise ValueError("broadcast_conditioning: tensor sizes do not match (n_obj, latent_shape, ctx_dim)")
    else:
        buf = torch.empty(n_lat + 2 * n_obj * ctx_dim, dtype=torch.float32, device=device)
    if ws > 1 or _force():
        dist.broadcast(buf, src=src)
    return unpack_conditioning(buf, n_obj, latent_shape, ctx_dim, cond_shape)


def all_gather_objects(local: Tensor, total: int) -> Tensor:
    """Concatenate per-rank object slabs (contiguous shard_range order).  Shards may differ by one object:
    pad to the largest shard so a single fixed-size all-gather moves everything."""
    rank, ws = world()
    if (ws == 1 and not _force()) or total == 0:
        return local
    sizes = [shard_range(total, ws, r)[1] - shard_range(total, ws, r)[0] for r in range(ws)]
    mx = max(sizes)               # >= 1; a rank whose shard is empty (more ranks than objects) sends only padding
    pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if local.is_cuda and dist.get_backend() == "gloo":
        # gloo has no device all-gather; only the single-GPU test rigs pair gloo with device tensors (RCCL needs one
        # GPU per rank), so the slab makes a host round trip there.  RCCL ("nccl") gathers device to device.
        host = torch.empty((ws * mx, *local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(host, pad.cpu().contiguous())
        out = host.to(local.device)
    else:
        out = torch.empty((ws * mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad.contiguous())
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(ws)], dim=0)


def all_reduce_max(t: Tensor) -> Tensor:
    """Element-wise max over the ranks (identity without a process group); gloo reduces on the host."""
    rank, ws = world()
    if ws == 1 and not _force():
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def any_rank_failed(failed: bool, device) -> bool:
    """One 4-byte all-reduce that every rank reaches whether or not its shard raised: lets a failing rank tell its peers
    BEFORE the data collective, so nobody blocks in an all-gather that will never complete."""
    rank, ws = world()
    if ws == 1 and not _force():
        return bool(failed)
    return bool(all_reduce_max(torch.tensor([1.0 if failed else 0.0], device=device))[0] > 0)


def sharded_rel2shape(sample_fn: Callable[[Tensor, Tensor, Tensor], Tensor], x_T: Tensor, uc: Tensor, c: Tensor,
                      gather: bool = True) -> Tensor:
    """Run `sample_fn(x_T, uc_slice, c_slice) -> sdf_slice` on this rank's contiguous object shard and
    all-gather the result.  With identical per-rank mini-batching the gathered tensor equals the
    single-rank result bit for bit (objects are independent; tests/test_dist_cpu.py)."""
    rank, ws = world()
    total = c.shape[0]
    lo, hi = shard_range(total, ws, rank)
    if hi > lo:
        local = sample_fn(x_T, uc[lo:hi], c[lo:hi])
    else:
        # more ranks than objects: this rank has nothing to sample but must still join the collective; the slab's
        # trailing shape comes from a peer-independent source -- the sampler's declared output shape
        shp = getattr(sample_fn, "out_shape", (1, 64, 64, 64))
        local = torch.empty((0, *shp), dtype=torch.float32, device=c.device)
    return all_gather_objects(local, total) if gather else local

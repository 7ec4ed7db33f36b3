"""world_size-2 (and 3) gloo tests of the object-sharding path (commonscenes_amd/dist.py) on CPU.

The HIP kernels cannot run here, so the per-object sampler is replaced by a deterministic CPU stand-in with
the same independence structure (each output row depends only on its own (x_T, uc_i, c_i)); what is under
test is the distributed plumbing: shard ranges, the packed broadcast, the padded all-gather and the
"sharded == unsharded, bit for bit" contract of SURVEY 8e."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from commonscenes_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(x_T, uc, c):
    """stand-in for rel2shape: per-object, deterministic, nonlinear."""
    B = c.shape[0]
    base = x_T.reshape(1, -1)[:, :64]
    feat = torch.tanh(c.reshape(B, -1)[:, :64] * 0.1 + base) - 3.0 * torch.sin(uc.reshape(B, -1)[:, :64])
    return feat.reshape(B, 1, 4, 4, 4).contiguous()


_fake_sampler.out_shape = (1, 4, 4, 4)      # what an EMPTY shard contributes to the all-gather (dist.sharded_rel2shape)


def _inputs(total):
    g = torch.Generator().manual_seed(5)
    return (torch.randn(1, 3, 16, 16, 16, generator=g), torch.randn(total, 1, 1280, generator=g),
            torch.randn(total, 1, 1280, generator=g))


def _worker(rank, ws, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        x_T, uc, c = _inputs(total)
        if rank == 0:
            bx, buc, bc = D.broadcast_conditioning(x_T, uc, c, total, "cpu", src=0)
        else:
            bx, buc, bc = D.broadcast_conditioning(None, None, None, total, "cpu", src=0)
        assert torch.equal(bx, x_T) and torch.equal(buc, uc) and torch.equal(bc, c)
        out = D.sharded_rel2shape(_fake_sampler, bx, buc, bc)
        lo, hi = D.shard_range(total, ws, rank)
        # by value (numpy): a torch tensor would travel as a shared-memory handle that dies with this process
        q.put((rank, out.numpy().copy(), (lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (8, 256): BASELINE configs[3]'s partition -- 256 objects, eight ranks of 32; (8, 5): more ranks than objects, three EMPTY
# shards that must still join the broadcast and the padded all-gather (the 8-rank pre-flight)
@pytest.mark.parametrize("ws,total", [(2, 8), (2, 7), (3, 10), (8, 256), (8, 5)])
def test_sharded_equals_unsharded(ws, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, total, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(ws)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x_T, uc, c = _inputs(total)
    if ws == 8:
        sizes = sorted(hi - lo for _, _, (lo, hi) in got)
        assert sizes == ([32] * 8 if total == 256 else [0, 0, 0, 1, 1, 1, 1, 1])
    ref = _fake_sampler(x_T, uc, c)
    ranges = sorted(r for _, _, r in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for rank, out, _ in got:
        out = torch.from_numpy(out)
        assert out.shape == ref.shape
        assert torch.equal(out, ref), f"rank {rank}: gathered result differs from the single-rank result"


def test_shard_range_properties():
    for total in (1, 7, 32, 255, 256):
        for ws in (1, 2, 3, 8):
            spans = [D.shard_range(total, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)
    assert D.shard_range(256, 8, 3) == (96, 128)


def test_pack_unpack_roundtrip():
    x_T, uc, c = _inputs(5)
    buf = D.pack_conditioning(x_T, uc, c)
    assert buf.numel() == 3 * 16 ** 3 + 2 * 5 * 1280
    a, b, cc = D.unpack_conditioning(buf, 5)
    assert torch.equal(a, x_T) and torch.equal(b, uc) and torch.equal(cc, c)


def _fail_worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        # rank 1 "fails": every rank must learn it from ONE small collective it reaches either way
        told = D.any_rank_failed(rank == 1, "cpu")
        nobody = D.any_rank_failed(False, "cpu")
        mx = D.all_reduce_max(torch.tensor([float(rank), 1.0 - rank]))
        q.put((rank, told, nobody, mx.tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_failure_flag_reaches_every_rank():
    """a rank that raises inside its shard must not leave its peers blocked in the all-gather -- rel2shape
    all-reduces a failure flag first (dist.any_rank_failed); the same reduction carries the fp32 fall-back decision."""
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fail_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, told, nobody, mx in got:
        assert told is True and nobody is False and mx == [1.0, 1.0]
    assert D.any_rank_failed(True, "cpu") is True and D.any_rank_failed(False, "cpu") is False     # no process group

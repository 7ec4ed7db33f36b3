import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The channel-split ResBlocks (unet.py::_res_split / cs_unet.hip::res_block_split) are taken from 65536 rows up in the
# product (>= 8 objects); the test models are small, so the suite forces the split route everywhere -- goldens,
# trajectories, driver-vs-driver equality all run through it -- and test_channel_split_* compares it with the
# small-batch (unsplit) route under the product threshold.
os.environ.setdefault("CS_CFG_SPLIT_MIN_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _reset_ops_switch_shadows():
    """a test may shadow a CsDebug switch with a module global (ops.SPLITK = False, monkeypatch.setattr(ops, ...)); whatever
    it leaves behind is dropped here, so no override outlives its test (ops._sw no longer clears anything on read)."""
    yield
    mod = sys.modules.get("commonscenes_amd.ops")
    if mod is not None:
        mod.reset_switches()


def rel_l2(a, b):
    """rel-L2 of a against b in fp64.  CS_PARITY_LOG=<file>: every measured value is appended with the id of the test that
    took it (how profiles/r03_parity_per_op.txt -- the record the per-op gates were set from -- is produced)."""
    import torch
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    v = float((a - b).norm() / b.norm().clamp_min(1e-30))
    log = os.environ.get("CS_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{v:.3e}\t{os.environ.get('PYTEST_CURRENT_TEST', '?')}\n")
    return v


GOLDEN = ROOT / "tests" / "golden"


def vq_flip_report(lat_mine, lat_ref, idx_mine, idx_ref, codebook):
    """VQ code-flip accounting for an end-to-end fixture ("VQ indices: report flips, target 0").

    lat_*: (B,3,g,g,g) latents handed to decode_no_quant (mine / the reference's), idx_*: (B,g,g,g) argmin indices,
    codebook: (n_embed, 3).  Returns (flips_per_object, unexplained) where `unexplained` counts flipped voxels that are
    NOT provable fp32 near-ties: with the reference's own latent z_ref, the code I picked must be at most
        2 |z_mine - z_ref| |e_mine - e_ref|  +  8 eps32 (|z|^2 + |e|^2)
    farther than the code the reference picked -- the first term is how much a latent perturbation can move the
    difference of two squared distances, the second the rounding of quantizer.py:76-79's fp32 z^2 + e^2 - 2 z.e."""
    import torch
    zm = lat_mine.detach().double().cpu().permute(0, 2, 3, 4, 1).reshape(-1, 3)
    zr = lat_ref.detach().double().cpu().permute(0, 2, 3, 4, 1).reshape(-1, 3)
    im = idx_mine.detach().cpu().reshape(-1).long()
    ir = idx_ref.detach().cpu().reshape(-1).long()
    cb = codebook.detach().double().cpu()
    B = lat_mine.shape[0]
    per = zm.shape[0] // B
    bad = (im != ir).nonzero().flatten()
    flips = [int(((bad >= b * per) & (bad < (b + 1) * per)).sum()) for b in range(B)]
    if bad.numel() == 0:
        return flips, 0
    em, er = cb[im[bad]], cb[ir[bad]]
    z_r, z_m = zr[bad], zm[bad]
    gap = ((z_r - em) ** 2).sum(1) - ((z_r - er) ** 2).sum(1)
    eps32 = 2.0 ** -23
    bound = 2 * (z_m - z_r).norm(dim=1) * (em - er).norm(dim=1) + 8 * eps32 * (
        (z_r ** 2).sum(1) + torch.maximum((em ** 2).sum(1), (er ** 2).sum(1)))
    return flips, int((gap > bound).sum())

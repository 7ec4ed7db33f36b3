"""Worker of test_f16x3_gpu.py::test_folded_upsample_convs_on_the_four_tap_slab_path: the Upsample convs folded onto the
source grid (cs_conv_gemm_up2: 3x2x2 / 2x2x2 kernels per output parity class) at sizes where the library picks the 256-row
tiles, written to a file.  Run once as is (four-tap slab path) and once with CS_NO_SLAB4=1 (per-tap gather path): the
chunk order, hence every accumulation order, is the same, so the two files must be equal bit for bit.  A third run with
CS_NO_UP2_DIRECT=1 keeps the slab kernel but sends the classes through the scratch tensor + interleave pass instead of the
scattered-store epilogue: equal again."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from commonscenes_amd import lib as L, ops, synth  # noqa: E402

out = {}
CASES = [("unet_hw", (48, 16, 8, 8), 48, 224, (0, 1, 1)),      # H, W doubled (3x2x2 taps), 256x224 tile
         ("unet_hw_w4", (200, 16, 4, 4), 32, 224, (0, 1, 1)),  # W = 4: a 256-row tile spans samples
         ("dec_dhw", (12, 16, 16, 16), 32, 128, (1, 1, 1)),    # all three doubled (2x2x2 taps), 256x128 tile
         ("dec_dhw_256", (13, 16, 16, 16), 16, 256, (1, 1, 1)),
         ("ragged_strided", (77, 16, 5, 8), 32, 224, (0, 1, 1)),   # M = 49280 = 192.5 tiles; out = a slice of a wider buffer
         ("unet_dhw", (96, 8, 8, 8), 32, 224, (1, 1, 1)),      # dims = 4 UNet (concat conditioning): all three doubled, 256x224 tile
         ("dec_big", (33, 32, 32, 32), 128, 128, (1, 1, 1))]   # a 4.4 GiB output: the scattered store's windows are per tile
for name, shp, cin, cout, up in CASES:
    x = synth.tensor_device(f"s4:x:{name}", (*shp, cin), 1.0)
    x[1] = float("nan") if name == "unet_hw_w4" else x[1]      # a NaN sample must stay confined to itself
    w = synth.tensor_device(f"s4:w:{name}", (cout, cin, 3, 3, 3), (cin * 27) ** -0.5)
    b = synth.tensor_device(f"s4:b:{name}", (cout,), 0.1)
    pk = ops.pack_weight(w, b, math=L.MATH_F16X3, fold_up=up)
    prof = ops.GEMM_PROFILE = []
    kw = {}
    if name == "ragged_strided":
        buf = torch.full((shp[0], shp[1] << up[0], shp[2] << up[1], shp[3] << up[2], cout + 32), 7.0, device="cuda")
        kw["out"] = buf[..., 16:16 + cout]
    y = ops.conv_gemm(x, pk, up=up, act=L.ACT_SILU if name == "dec_dhw" else L.ACT_NONE, **kw)
    ops.GEMM_PROFILE = None
    torch.cuda.synchronize()
    if name == "dec_big":                                    # too large to keep: a strided sample + the two ends + a checksum
        flat = y.view(-1)
        out[name] = torch.cat([flat[::1000003], flat[:4096], flat[-4096:], y.double().sum().float().view(1)]).cpu()
        del y, flat, x
        torch.cuda.empty_cache()
        continue
    out[name] = y.cpu()
    if name == "ragged_strided":
        out[name + ":pad"] = torch.stack([buf[..., :16].min(), buf[..., :16].max(), buf[..., 16 + cout:].min(),
                                          buf[..., 16 + cout:].max()]).cpu()      # untouched columns: all 7.0
    out[name + ":tile"] = torch.tensor(prof[0]["tile"])
torch.save(out, sys.argv[1])

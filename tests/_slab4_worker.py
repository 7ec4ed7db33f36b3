"""Worker of test_f16x3_gpu.py::test_folded_upsample_convs_on_the_four_tap_slab_path: the Upsample convs folded onto the
source grid (cs_conv_gemm_up2: 3x2x2 / 2x2x2 kernels per output parity class) at sizes where the library picks the 256-row
tiles, written to a file.  Run once as is (four-tap slab path) and once with CS_NO_SLAB4=1 (per-tap gather path): the
chunk order, hence every accumulation order, is the same, so the two files must be equal bit for bit."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from commonscenes_amd import lib as L, ops, synth  # noqa: E402

out = {}
CASES = [("unet_hw", (48, 16, 8, 8), 48, 224, (0, 1, 1)),      # H, W doubled (3x2x2 taps), 256x224 tile
         ("unet_hw_w4", (200, 16, 4, 4), 32, 224, (0, 1, 1)),  # W = 4: a 256-row tile spans samples
         ("dec_dhw", (12, 16, 16, 16), 32, 128, (1, 1, 1)),    # all three doubled (2x2x2 taps), 256x128 tile
         ("dec_dhw_256", (13, 16, 16, 16), 16, 256, (1, 1, 1))]
for name, shp, cin, cout, up in CASES:
    x = synth.tensor_device(f"s4:x:{name}", (*shp, cin), 1.0)
    x[1] = float("nan") if name == "unet_hw_w4" else x[1]      # a NaN sample must stay confined to itself
    w = synth.tensor_device(f"s4:w:{name}", (cout, cin, 3, 3, 3), (cin * 27) ** -0.5)
    b = synth.tensor_device(f"s4:b:{name}", (cout,), 0.1)
    pk = ops.pack_weight(w, b, math=L.MATH_F16X3, fold_up=up)
    prof = ops.GEMM_PROFILE = []
    y = ops.conv_gemm(x, pk, up=up, act=L.ACT_SILU if name == "dec_dhw" else L.ACT_NONE)
    ops.GEMM_PROFILE = None
    torch.cuda.synchronize()
    out[name] = y.cpu()
    out[name + ":tile"] = torch.tensor(prof[0]["tile"])
torch.save(out, sys.argv[1])

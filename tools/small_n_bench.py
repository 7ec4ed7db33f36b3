"""Times the library's automatic choice (the 256x64 tile) against the 64x64 tile for convs with <= 4 output channels, on the two
shapes of the hot path: the UNet's 224 -> 3 output conv (batch 64, 16^3) and the VQ decoder's 64 -> 1 (per object,
64^3).  usage (GPU box): python tools/small_n_bench.py [objects]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for name, nb, s, cin, cout in (("unet out 224->3", 2 * nobj, 16, 224, 3), ("decoder conv_out 64->1", nobj // 4 or 1, 64, 64, 1)):
        x = synth.tensor_device("snb:x", (nb, s, s, s, cin), 1.0)
        w = synth.tensor_device("snb:w", (cout, cin, 3, 3, 3), (cin * 27) ** -0.5)
        b = synth.tensor_device("snb:b", (cout,), 1.0)
        pk = ops.pack_weight(w, b, math=L.MATH_F16X3)
        m = nb * s ** 3
        row = [f"{name}: M={m}"]
        ref = None
        for label, tile in (("auto", 0), ("tile3 64x64", 3), ("tile7 256x64", 7)):
            try:
                o = ops.conv_gemm(x, pk, tile=tile)
            except Exception as e:          # tile 7 wants cout == 64-column tiles only where it applies
                row.append(f"{label} n/a")
                continue
            ms = timeit(lambda: ops.conv_gemm(x, pk, tile=tile))
            if ref is None:
                ref = o
            err = float((o - ref).norm() / ref.norm())
            gbs = (x.numel() * 4 + m * cout * 4) / ms / 1e6
            row.append(f"{label} {ms * 1e3:.0f} us ({gbs:.0f} GB/s of input+output, rel diff {err:.1e})")
        print("  ".join(row))


if __name__ == "__main__":
    main()

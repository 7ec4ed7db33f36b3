"""GroupNorm at one object (CFG batch 2): cs_groupnorm's single launch against the statistics + apply launches.
usage (GPU box): python tools/gn_bench.py [nb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, synth


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    print(f"nb={nb}: rows x channels | cs_groupnorm (auto) us | stats + apply us")
    for rows, c in ((4096, 224), (4096, 448), (4096, 672), (1024, 448), (1024, 896), (1024, 1120), (1024, 1344),
                    (256, 672), (256, 1344), (256, 2016), (64, 672), (64, 1344)):
        x = synth.tensor_device("gnb:x", (nb, rows, c), 1.0)
        g = synth.tensor_device("gnb:g", (c,), 1.0)
        b = synth.tensor_device("gnb:b", (c,), 1.0)
        y = torch.empty_like(x)
        ws = torch.empty(lib.cs_groupnorm_ws_bytes(nb, 32) // 8, dtype=torch.float64, device="cuda")
        st = torch.empty(nb, 32, 2, device="cuda")
        a = (x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), nb, rows, c, c, c, 32, 1e-5, L.ACT_SILU, ws.data_ptr(), st.data_ptr(), s)
        t1 = timeit(lambda: lib.cs_groupnorm(*a))

        def three():
            lib.cs_groupnorm_stats(x.data_ptr(), nb, rows, c, c, 32, 1e-5, ws.data_ptr(), st.data_ptr(), s)
            lib.cs_groupnorm_apply(x.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), nb, rows, c, c, c, 32, L.ACT_SILU, s)
        t3 = timeit(three)
        print(f"  {rows:5d} x {c:5d} ({nb * rows * c * 4 / 2**20:5.1f} MB) | {t1:7.1f} | {t3:7.1f}")


if __name__ == "__main__":
    main()

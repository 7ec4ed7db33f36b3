#!/usr/bin/env python
"""One attention shape (the UNet's 16x8x8 level at batch 64: 1024 tokens, 8 heads x 56), F16X3, a few launches: the PMC target
of tools/attn_pmc.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from commonscenes_amd import lib as L, ops, synth

n, c, heads, nb = 1024, 448, 8, 64
qkv = synth.tensor_device("qkv", (nb, n, 3 * c), 1.0)
q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
for _ in range(int(os.environ.get("ATTN_ITERS", "3"))):
    ops.attention(q, k, v, heads, (c // heads) ** -0.5, math=L.MATH_F16X3)
torch.cuda.synchronize()

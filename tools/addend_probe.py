#!/usr/bin/env python3
"""Data gradient of a bottleneck's first 1 x 1 (256 -> 1024 at 16 x 33 x 33) with and without the shortcut's gradient
added in the epilogue, cold operands, for a kernel trace: what the addend costs per launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402

N, H, Cin, Cout = 16, 33, 256, 1024
M = N * H * H
dev, dt = "cuda", torch.bfloat16
sets = 12
data = []
for _ in range(sets):
    x = torch.randn((M, Cin), device=dev).to(dt)
    ad = torch.randn((M, Cout), device=dev).to(dt)
    bits = torch.randint(0, 256, (M, Cout // 8), device=dev, dtype=torch.uint8)
    out = torch.empty((M, Cout), dtype=dt, device=dev)
    data.append((x, ad, bits, out))
w = ops.pack_weight(torch.randn((Cout, Cin, 1, 1), device=dev) * 0.05, dt)
for rnd in range(4):
    for kind in range(3):
        for x, ad, bits, out in data:
            if kind == 0:
                ops.conv_fwd([(x, w, 1, 1)], N, H, H, Cout, out)
            elif kind == 1:
                ops.conv_fwd([(x, w, 1, 1)], N, H, H, Cout, out, addend=ad)
            else:
                ops.conv_fwd([(x, w, 1, 1)], N, H, H, Cout, out, addend=ad, addend_bits=bits)
torch.cuda.synchronize()

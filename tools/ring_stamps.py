#!/usr/bin/env python3
"""Where a loading wavefront of the forward ring tile spends a K step (experiment build, SEGK_DMA_ABL bit 256): cycles
waiting for its loads, at the barrier, in the step bookkeeping and issuing, from the first tile of the first
workgroup.  usage: SEGK_DMA_ABL=256 tools/ring_stamps.py   (add 2 / 1 for the skeleton variants)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402

for name, H, cin, cout, R in (("dec 3x3 256", 129, 256, 256, 3), ("feeder l3 3x3", 33, 256, 256, 3), ("aspp 3x3 d6", 33, 2048, 256, 3)):
    n, dil = 16, 6 if "aspp" in name else 1
    M = n * H * H
    x = torch.randn((M, cin), device="cuda").to(torch.bfloat16)
    w = ops.pack_weight(torch.randn((cout, cin, R, R), device="cuda") * 0.05, torch.bfloat16)
    out = torch.empty((M, cout), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        stats = ops.conv_fwd([(x, w, R, dil)], n, H, H, cout, out, want_stats=True)
    torch.cuda.synchronize()
    v = stats.reshape(-1)[:32].cpu().reshape(4, 8)
    for wv in range(4):
        wait, bar, adv, iss, total = [float(t) for t in v[wv, :5]]
        if total <= 0:
            continue
        steps = max(total - 2, 1)
        print("%-14s loader %d: %3d steps; per step: wait %6.0f  barrier %6.0f  bookkeeping %6.0f  issue %6.0f  = %6.0f cycles"
              % (name, wv, int(total), wait / total, bar / total, adv / steps, iss / steps,
                 (wait + bar) / total + (adv + iss) / steps))

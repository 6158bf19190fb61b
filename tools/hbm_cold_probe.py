#!/usr/bin/env python3
"""The loss / resampling kernels at the benchmark's shapes, each launched on ONE operand set over and over and then on
three sets in rotation (1.1 GB of logits: none of them is in the 256 MB memory-side cache when its turn comes), for a
kernel trace (tools/hbm_cold_probe.sh): how much of a stand-alone timing is the cache's."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402

dev = "cuda"
N, K, Hi, Wi, Ho, Wo = 16, 21, 129, 129, 513, 513
cpad = 24
torch.manual_seed(0)
sets = []
for _ in range(3):
    logits = torch.randn(N, K, Ho, Wo, device=dev)
    target = torch.randint(0, K, (N, Ho, Wo), device=dev).float()
    target[:, :40] = 255.0
    low = torch.randn(N * Hi * Wi, cpad, device=dev)
    sets.append((logits, target, low, torch.empty(N * Hi * Wi, cpad, device=dev), torch.empty(N, K, Ho, Wo, device=dev)))
coef = torch.full((1,), 1e-6, device=dev)
_, lse = ops.ce_fwd(sets[0][0], sets[0][1], None, 255)


def run(kind, s):
    logits, target, low, dlow, full = s
    if kind == 0:
        ops.upsample_fwd(low[:, :K], N, K, Hi, Wi, full, Ho, Wo, dst_layout=ops.NCHW)
    elif kind == 1:
        ops.ce_fwd(logits, target, None, 255)
    elif kind == 2:
        ops.ce_bwd(logits, target, None, lse, 255, coef)
    else:
        ops.upsample_bwd(logits, ops.NCHW, N, K, Ho, Wo, dlow, cpad, Hi, Wi)


for kind in range(4):
    for _ in range(9):
        run(kind, sets[0])
    for r in range(3):
        for s in sets:
            run(kind, s)
torch.cuda.synchronize()

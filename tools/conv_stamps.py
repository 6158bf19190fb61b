#!/usr/bin/env python3
"""Diagnostic: per-section cycle sums of the deep-pipelined convolution kernel (SEGK_CONV_ABL=9 build path).
Prints, per wavefront group, the average cycles per phase spent in each section."""
import os
import sys

os.environ["SEGK_CONV_ABL"] = "9"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402

n, H, cin, cout, R, dil = 16, 129, 256, 256, 3, 1
M = n * H * H
x = torch.randn((M, cin), device="cuda").to(torch.bfloat16)
w = torch.randn((cout, cin, R, R), device="cuda") * 0.05
wp = ops.pack_weight(w, torch.bfloat16)
out = torch.empty((M, cout), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    stats = ops.conv_fwd([(x, wp, R, dil)], n, H, H, cout, out, want_stats=True)
torch.cuda.synchronize()
t = stats[:, 0, :64].reshape(-1, 8, 8).cpu()      # [tile][wave][k]
names = ["load section", "vmcnt wait", "barrier 1", "lgkmcnt wait", "mfma section", "barrier 2"]
phases = float(t[0, 0, 6]) * 4
print("tiles %d, K-tiles per tile %d" % (t.shape[0], int(t[0, 0, 6])))
for g, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    tot = 0.0
    for k, nm in enumerate(names):
        v = float(t[1:-1, sl, k].mean()) / phases
        tot += v
        print("%s  %-14s %8.1f cycles/phase" % (g, nm, v))
    print("%s  %-14s %8.1f cycles/phase" % (g, "sum", tot))

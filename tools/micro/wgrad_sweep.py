"""Graph-timed sweep of the number of parts (and the tile) of weight gradients on the feeder / head shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from segk import ops
from unit_bench import graphed_us
n, dt, dev = 16, torch.bfloat16, "cuda"
for name, H, Cin, Cout, R, dil in (("l3 3x3 256", 33, 256, 256, 3, 1), ("l4 3x3 512", 33, 512, 512, 3, 2), ("l2 3x3 128", 65, 128, 128, 3, 1),
                                   ("aspp 3x3", 33, 2048, 256, 3, 6), ("aspp 1x1", 33, 2048, 256, 1, 1), ("proj", 33, 1280, 256, 1, 1),
                                   ("dec 320", 129, 320, 256, 3, 1), ("dec 256", 129, 256, 256, 3, 1)):
    M = n * H * H
    x = torch.randn((M, Cin), device=dev).to(dt); dy = torch.randn((M, Cout), device=dev).to(dt)
    tile, sp = ops.wgrad_plan(M, Cin, Cout, R, dt)
    for t in (128, 256):
        if t == 256 and not (Cin >= 256 and Cout % 256 == 0):
            continue
        res = []
        for s in sorted(set([max(1, sp // 2), max(1, (3 * sp) // 4), sp, sp * 3 // 2, sp * 2, sp * 3])):
            try:
                res.append("%d:%.1f" % (s, graphed_us(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, Cout, Cin, splits=s, tile=t))))
            except Exception as e:
                res.append("%d:err" % s)
        print("%-12s plan (%d, %d) | tile %d parts:us %s" % (name, tile, sp, t, "  ".join(res)), flush=True)

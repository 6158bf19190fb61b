#!/usr/bin/env python3
"""The batch-norm backward statistics passes on the feeder's shapes with cold operands, for a kernel trace
(tools/bn_stats_sweep.sh): plain (ReLU pattern recomputed) and with the stored ReLU bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402
from segk._lib import PM  # noqa: E402


def main():
    n, dt, dev = 16, torch.bfloat16, "cuda"
    for C, H in [(256, 33), (1024, 33), (512, 33), (2048, 33), (128, 65), (512, 65), (256, 129)]:
        M = n * H * H
        sets = max(2, int(1.2e9 // (M * C * 2 * 2)) + 1)
        ops_ = []
        for _ in range(sets):
            x = torch.randn((M, C), device=dev).to(dt)
            g = torch.randn((M, C), device=dev).to(dt)
            bits = torch.randint(0, 256, (M, C // 8), device=dev, dtype=torch.uint8)
            ops_.append((x, g, bits))
        x = ops_[0][0]
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        coef = ops.bn_coeffs(ops.bn_local_sums(x, PM), gamma, beta, 1e-5, 0.1, None, None)
        for x, g, bits in ops_:
            ops.bn_bwd_stat_partials(g, x, coef, PM, True)
        torch.cuda.synchronize()
        for r in range(6):
            for k in range(2):
                for x, g, bits in ops_:
                    if k == 0:
                        ops.bn_bwd_stat_partials(g, x, coef, PM, True)
                    else:
                        ops.bn_bwd_stat_partials(g, x, coef, PM, True, act=bits)
        torch.cuda.synchronize()
        print("shape", C, H, "sets", sets, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The batch-norm apply passes of a bottleneck on the feeder's shapes, a few launches each, for a kernel trace
(tools/bn_apply_sweep.sh): forward apply with residual and ReLU bits, backward apply with the bits, and the plain pair."""
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
        # enough distinct operand sets that none of them is still in the memory-side cache (256 MB) when its turn
        # comes again: in the training step these passes read what was written many launches ago
        sets = max(2, int(1.2e9 // (M * C * 2 * 5)) + 1)
        ops_ = []
        for _ in range(sets):
            x = torch.randn((M, C), device=dev).to(dt)
            g = torch.randn((M, C), device=dev).to(dt)
            res = torch.randn((M, C), device=dev).to(dt)
            ops_.append((x, g, res, torch.empty_like(x), torch.empty_like(x), ops.relu_bits_like(x)))
        x, g = ops_[0][0], ops_[0][1]
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        coef = ops.bn_coeffs(ops.bn_local_sums(x, PM), gamma, beta, 1e-5, 0.1, None, None)
        gs = ops.bn_bwd_local_sums(g, x, coef, PM, True)
        for x, g, res, y, dx, bits in ops_:
            ops.bn_apply(x, y, coef, PM, True, res=res, relu_bits=bits)
        torch.cuda.synchronize()
        for r in range(6):
            for k in range(4):
                for x, g, res, y, dx, bits in ops_:
                    if k == 0:
                        ops.bn_apply(x, y, coef, PM, True)
                    elif k == 1:
                        ops.bn_apply(x, y, coef, PM, True, res=res, relu_bits=bits)
                    elif k == 2:
                        ops.bn_bwd_apply(g, x, dx, coef, gs, PM, True)
                    else:
                        ops.bn_bwd_apply(g, x, dx, coef, gs, PM, True, act=bits)
        torch.cuda.synchronize()
        print("shape", C, H, "sets", sets, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmark of the pixel-major batch-norm kernels on the feeder's layer shapes (n = 16, bf16), with the
achieved HBM bandwidth per kernel and PyTorch's (MIOpen) batch norm on the same channels_last tensor beside it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from segk import ops  # noqa: E402
from segk._lib import PM  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    n, dt, dev = 16, torch.bfloat16, "cuda"
    shapes = [(64, 257), (64, 129), (256, 129), (128, 129), (128, 65), (512, 65), (256, 65), (256, 33), (1024, 33),
              (512, 33), (2048, 33)]
    print("%-12s %9s | %7s %7s %7s %7s (us) | %6s %6s %6s %6s (TB/s) | torch fwd+bwd us" %
          ("C x H", "MB", "stats", "apply", "bstats", "bapply", "stats", "apply", "bstats", "bapply"))
    for C, H in shapes:
        M = n * H * H
        x = torch.randn((M, C), device=dev).to(dt)
        g = torch.randn((M, C), device=dev).to(dt)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        gamma = torch.ones(C, device=dev)
        beta = torch.zeros(C, device=dev)
        sums = ops.bn_local_sums(x, PM)
        coef = ops.bn_coeffs(sums, gamma, beta, 1e-5, 0.1, None, None)
        gs = ops.bn_bwd_local_sums(g, x, coef, PM, True)
        t1 = timeit(lambda: ops.bn_local_sums(x, PM))
        t2 = timeit(lambda: ops.bn_apply(x, y, coef, PM, True))
        t3 = timeit(lambda: ops.bn_bwd_local_sums(g, x, coef, PM, True))
        t4 = timeit(lambda: ops.bn_bwd_apply(g, x, dx, coef, gs, PM, True))
        mb = M * C * 2 / 1e6
        bw = [mb * k / t for k, t in ((1, t1), (2, t2), (2, t3), (3, t4))]  # MB/us = TB/s
        x4 = x.reshape(n, H, H, C).permute(0, 3, 1, 2).requires_grad_(True)
        g4 = g.reshape(n, H, H, C).permute(0, 3, 1, 2)

        def torch_bn():
            x4.grad = None
            F.relu(F.batch_norm(x4, None, None, gamma, beta, True, 0.1, 1e-5)).backward(g4)
        tt = timeit(torch_bn, 10)
        print("%4d x %-5d %9.1f | %7.1f %7.1f %7.1f %7.1f      | %6.2f %6.2f %6.2f %6.2f        | %8.1f" %
              (C, H, mb, t1, t2, t3, t4, *bw, tt))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The feeder's 3x3 bottleneck convolutions (n = 16, bf16): the implicit-GEMM kernels (forward with batch-norm
statistics, data gradient, weight gradient) beside PyTorch-ROCm's convolution on the same channels_last tensors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from segk import ops  # noqa: E402


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
    return s.elapsed_time(e) / iters * 1e3


def main():
    torch.backends.cudnn.benchmark = True
    n, dt, dev = 16, torch.bfloat16, "cuda"
    for name, H, C, dil in (("layer1 3x3", 129, 64, 1), ("layer2 3x3", 65, 128, 1), ("layer3 3x3", 33, 256, 1),
                            ("layer4 3x3 d2", 33, 512, 2), ("layer4 3x3 d4", 33, 512, 4)):
        M = n * H * H
        x = torch.randn((M, C), device=dev).to(dt)
        dy = torch.randn((M, C), device=dev).to(dt)
        w = torch.randn((C, C, 3, 3), device=dev) * 0.05
        wp, wpt = ops.pack_weight(w, dt), ops.pack_weight(w, dt, transpose=True)
        out = torch.empty((M, C), dtype=dt, device=dev)
        wcl = w.to(dt).contiguous(memory_format=torch.channels_last)
        t_p = timeit(lambda: ops.pack_weight(wcl, dt, transpose=True))
        t_f = timeit(lambda: ops.conv_fwd([(x, wcl if C % 32 == 0 else wp, 3, dil)], n, H, H, C, out, want_stats=True))
        t_d = timeit(lambda: ops.conv_fwd([(dy, wpt, 3, dil)], n, H, H, C, out))
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, n, H, H, 3, dil, C, C))
        x4 = x.reshape(n, H, H, C).permute(0, 3, 1, 2).requires_grad_(True)
        w4 = w.to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        g4 = dy.reshape(n, H, H, C).permute(0, 3, 1, 2)
        t_tf = timeit(lambda: F.conv2d(x4, w4, padding=dil, dilation=dil))

        def both():
            x4.grad = None
            w4.grad = None
            F.conv2d(x4, w4, padding=dil, dilation=dil).backward(g4)
        t_tb = timeit(both) - t_tf
        t_st = timeit(lambda: ops.bn_local_sums(out, 0))
        print("%-14s M=%6d C=%4d | segk pack^T %5.1f fwd+stats %6.1f dgrad %6.1f wgrad %6.1f us | torch fwd %6.1f (+stats kernel %5.1f) "
              "bwd %6.1f us" % (name, M, C, t_p, t_f, t_d, t_w, t_tf, t_st, t_tb))


if __name__ == "__main__":
    main()

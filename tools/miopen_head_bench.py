#!/usr/bin/env python3
"""Calibration: PyTorch-ROCm (MIOpen) convolution on the head's shapes (n = 16, bf16, channels_last), forward and
backward, in nominal TFLOP/s - the figure the hand-written kernels of this repository are compared with."""
import torch
import torch.nn.functional as F


def timeit(fn, iters=10):
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
    for name, H, Cin, Cout, R, dil in (("aspp 3x3 d6", 33, 2048, 256, 3, 6), ("aspp 3x3 d18", 33, 2048, 256, 3, 18),
                                       ("aspp 1x1", 33, 2048, 256, 1, 1), ("dec 3x3 304", 129, 304, 256, 3, 1),
                                       ("dec 3x3 256", 129, 256, 256, 3, 1)):
        x = torch.randn((n, Cin, H, H), device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn((Cout, Cin, R, R), device=dev) * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        w.requires_grad_(True)
        pad = dil * (R // 2)
        y = F.conv2d(x, w, padding=pad, dilation=dil)
        g = torch.randn_like(y)
        tf = timeit(lambda: F.conv2d(x, w, padding=pad, dilation=dil))

        def both():
            x.grad = None
            w.grad = None
            F.conv2d(x, w, padding=pad, dilation=dil).backward(g)
        tb = timeit(both) - tf
        fl = 2.0 * n * H * H * Cout * Cin * R * R
        print("%-14s fwd %7.1f us %6.1f TF | dgrad+wgrad %7.1f us %6.1f TF" % (name, tf, fl / tf / 1e6, tb, 2 * fl / tb / 1e6))


if __name__ == "__main__":
    main()

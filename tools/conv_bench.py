#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernels on the head's real shapes (config A, n = 16 by default).
Prints nominal TFLOP/s per launch.  usage: tools/conv_bench.py [--n 16] [--dtype bf16|fp32] [--only substr]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import ops  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda"
    n = args.n
    # name, H, Cin(true), Cin(pad), Cout, R, dil
    fwd = [("aspp1 1x1", 33, 2048, 2048, 256, 1, 1), ("aspp2 3x3 d6", 33, 2048, 2048, 256, 3, 6),
           ("aspp3 3x3 d12", 33, 2048, 2048, 256, 3, 12), ("aspp4 3x3 d18", 33, 2048, 2048, 256, 3, 18),
           ("aspp proj 1x1", 33, 1280, 1280, 256, 1, 1), ("dec conv1 1x1", 129, 256, 256, 48, 1, 1),
           ("dec 3x3 304", 129, 304, 320, 256, 3, 1), ("dec 3x3 256", 129, 256, 256, 256, 3, 1),
           ("dec cls 1x1", 129, 256, 256, 21, 1, 1),
           ("dgrad dec 3x3 256", 129, 256, 256, 256, 3, 1), ("dgrad dec 3x3 304", 129, 256, 256, 304, 3, 1),
           ("dgrad proj", 33, 256, 256, 1280, 1, 1),
           # the feeder's stride-1 bottleneck 3x3s (ResNet layers 3 and 4 at output stride 16)
           ("feeder l3 3x3", 33, 256, 256, 256, 3, 1), ("feeder l4 3x3 d2", 33, 512, 512, 512, 3, 2),
           ("feeder l4 3x3 d4", 33, 512, 512, 512, 3, 4),
           # ... and their 1x1 neighbours: short K loops onto many channels, long ones onto few
           ("feeder l3 c3 1x1", 33, 256, 256, 1024, 1, 1), ("feeder l3 c1 1x1", 33, 1024, 1024, 256, 1, 1),
           ("feeder l4 c3 1x1", 33, 512, 512, 2048, 1, 1), ("feeder l4 c1 1x1", 33, 2048, 2048, 512, 1, 1),
           # same map and output width, growing K: the slope is the cost of a K step, the intercept the fixed part
           ("probe k 3x3 c64", 33, 64, 64, 256, 3, 1), ("probe k 3x3 c128", 33, 128, 128, 256, 3, 1),
           ("probe k 3x3 c512", 33, 512, 512, 256, 3, 1), ("probe k 3x3 c1024", 33, 1024, 1024, 256, 3, 1),
           # the decoder's map (several tiles per CU), growing K: slope = K step, intercept = the tiles' fixed cost
           ("probe d 3x3 c64", 129, 64, 64, 256, 3, 1), ("probe d 3x3 c128", 129, 128, 128, 256, 3, 1),
           ("probe d 3x3 c512", 129, 512, 512, 256, 3, 1)]
    for name, H, cin, cpad, cout, R, dil in fwd:
        if args.only and args.only not in name:
            continue
        M = n * H * H
        x = torch.randn((M, cpad), device=dev).to(dt)
        w = torch.randn((cout, cin, R, R), device=dev) * 0.05
        wp = ops.pack_weight(w, dt)
        out = torch.empty((M, ops.round_up(cout, 4)), dtype=dt, device=dev)
        ms = timeit(lambda: ops.conv_fwd([(x, wp, R, dil)], n, H, H, cout, out, want_stats=not os.environ.get("CONV_BENCH_NOSTATS")))
        fl = 2.0 * M * cout * cin * R * R
        print("fwd   %-20s M=%7d  %8.1f us  %7.1f TF nominal" % (name, M, ms * 1e3, fl / ms / 1e9))
    if not args.only or "asppdgrad" in args.only:
        M = n * 33 * 33
        segs = []
        for R, dil in ((1, 1), (3, 6), (3, 12), (3, 18)):
            dy = torch.randn((M, 256), device=dev).to(dt)
            w = torch.randn((256, 2048, R, R), device=dev) * 0.05
            segs.append((dy, ops.pack_weight(w, dt, transpose=True), R, dil))
        out = torch.empty((M, 2048), dtype=torch.float32, device=dev)
        ms = timeit(lambda: ops.conv_fwd(segs, n, 33, 33, 2048, out))
        fl = 2.0 * M * 2048 * 256 * 28
        print("dgrad %-20s M=%7d  %8.1f us  %7.1f TF nominal" % ("asppdgrad 4 segs", M, ms * 1e3, fl / ms / 1e9))
    if not args.only or "asppbatch" in args.only:
        # the three dilated ASPP branches in one launch (blockIdx.y = branch)
        M = n * 33 * 33
        x = torch.randn((M, 2048), device=dev).to(dt)
        w = torch.randn((3, 256, 2048, 3, 3), device=dev) * 0.05
        wp = torch.stack([ops.pack_weight(w[b], dt) for b in range(3)]).contiguous()
        out = torch.empty((M, 1280), dtype=dt, device=dev)
        ms = timeit(lambda: ops.conv_fwd([(x, wp, 3, 6)], n, 33, 33, 256, out[:, 256:], want_stats=True, batch=3,
                                         dil_step=6, out_batch=256))
        fl = 3 * 2.0 * M * 256 * 2048 * 9
        print("fwd   %-20s M=%7d  %8.1f us  %7.1f TF nominal" % ("asppbatch 3 x 3x3", M, ms * 1e3, fl / ms / 1e9))
    wg = [("aspp 3x3 d6", 33, 2048, 256, 3, 6), ("aspp 3x3 d18", 33, 2048, 256, 3, 18), ("aspp proj", 33, 1280, 256, 1, 1),
          ("dec 3x3 304", 129, 320, 256, 3, 1), ("dec 3x3 256", 129, 256, 256, 3, 1), ("dec conv1", 129, 256, 48, 1, 1)]
    for name, H, cin, cout, R, dil in wg:
        if args.only and args.only not in ("wgrad " + name):
            continue
        M = n * H * H
        x = torch.randn((M, cin), device=dev).to(dt)
        dy = torch.randn((M, cout), device=dev).to(dt)
        ms = timeit(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin))
        fl = 2.0 * M * cout * cin * R * R
        print("wgrad %-20s M=%7d  %8.1f us  %7.1f TF nominal (incl. slab reduce)" % (name, M, ms * 1e3, fl / ms / 1e9))


if __name__ == "__main__":
    main()

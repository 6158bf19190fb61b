#!/usr/bin/env python3
"""Weight-gradient kernels on the step's real shapes (batch 16): the older tiles' plan and the current plan against the ring tile
(tile = 384) over a sweep of part counts, with the ring's result checked against the older tile's.
usage: tools/wgrad_bench.py [--only substr] [--parts 1,2,4]"""
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
    ap.add_argument("--only", default="")
    ap.add_argument("--parts", default="")
    args = ap.parse_args()
    dt, dev, n = torch.bfloat16, "cuda", args.n
    # name, H, Cin (padded), Cout, R, dilation
    shapes = [("aspp 1x1", 33, 2048, 256, 1, 1), ("aspp 3x3 d6", 33, 2048, 256, 3, 6),
              ("aspp 3x3 d12", 33, 2048, 256, 3, 12), ("aspp 3x3 d18", 33, 2048, 256, 3, 18),
              ("aspp proj", 33, 1280, 256, 1, 1), ("dec 3x3 304", 129, 320, 256, 3, 1),
              ("dec 3x3 256", 129, 256, 256, 3, 1), ("feeder l3 3x3", 33, 256, 256, 3, 1),
              ("feeder l4 3x3 d2", 33, 512, 512, 3, 2), ("feeder l4 3x3 d4", 33, 512, 512, 3, 4)]
    for name, H, cin, cout, R, dil in shapes:
        if args.only and args.only not in name:
            continue
        M = n * H * H
        torch.manual_seed(1)
        x = torch.randn((M, cin), device=dev).to(dt)
        dy = torch.randn((M, cout), device=dev).to(dt)
        fl = 2.0 * M * cout * cin * R * R
        tile0, sp0 = ops.wgrad_plan(M, cin, cout, R, dt, ring=False)
        ref = ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin, splits=sp0, tile=tile0)
        ms = timeit(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin, splits=sp0, tile=tile0))
        print("%-18s older tile %3d   parts %3d  %8.1f us  %7.1f TF" % (name, tile0, sp0, ms * 1e3, fl / ms / 1e9), flush=True)
        tile1, sp1 = ops.wgrad_plan(M, cin, cout, R, dt)
        ms = timeit(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin))
        print("%-18s plan  tile %3d   parts %3d  %8.1f us  %7.1f TF" % (name, tile1, sp1, ms * 1e3, fl / ms / 1e9), flush=True)
        tiles = (cout // 256) * ((cin + 127) // 128) * R * R
        if args.parts:
            cand = [int(v) for v in args.parts.split(",")]
        else:
            cand = sorted({max(1, t // tiles) for t in (128, 256, 384, 512, 768, 1024)})
        for sp in cand:
            got = ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin, splits=sp, tile=384)
            err = float((got - ref).abs().max() / ref.abs().max())
            ms = timeit(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin, splits=sp, tile=384))
            print("%-18s ring  %4d wgs parts %3d  %8.1f us  %7.1f TF   rel err %.2e%s"
                  % (name, tiles * sp, sp, ms * 1e3, fl / ms / 1e9, err, "" if err < 2e-3 else "   <-- MISMATCH"), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the wavefronts of the ring weight-gradient tile spend a 64-pixel chunk (experiment build, SEGK_WGRAD_STAMP=1):
cycle stamps of the first workgroup - loaders: vote + bookkeeping, waiting for loads, barrier, issue; multipliers:
barrier waits and everything else.  usage: SEGK_WGRAD_STAMP=1 tools/wring_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import _lib, ops  # noqa: E402

lib = _lib.load()
for name, H, cin, cout, R, dil in (("dec 3x3 256", 129, 256, 256, 3, 1), ("feeder l3 3x3", 33, 256, 256, 3, 1),
                                   ("aspp 3x3 d6", 33, 2048, 256, 3, 6)):
    n = 16
    M = n * H * H
    x = torch.randn((M, cin), device="cuda").to(torch.bfloat16)
    dy = torch.randn((M, cout), device="cuda").to(torch.bfloat16)
    for _ in range(3):
        ops.conv_wgrad(x, dy, n, H, H, R, dil, cout, cin)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 96)()
    assert lib.segk_debug_wring_stamps(buf) == 0
    v = [list(buf[w * 8:w * 8 + 8]) for w in range(12)]
    for w in (0, 3):
        book, wait, bar, iss, chunks = v[w][:5]
        c = max(chunks, 1)
        print("%-14s loader %d:     %4d chunks; per chunk: vote + bookkeeping %5.0f  wait %5.0f  barrier %5.0f  issue %5.0f  = %5.0f cycles"
              % (name, w, chunks, book / c, wait / c, bar / c, iss / c, (book + wait + bar + iss) / c))
    for w in (4, 11):
        bar, rest, chunks = v[w][0], v[w][1], v[w][4]
        c = max(chunks, 1)
        print("%-14s multiplier %d: %4d chunks; per chunk: barrier %5.0f  reads + MFMAs %5.0f  = %5.0f cycles"
              % (name, w - 4, chunks, bar / c, rest / c, (bar + rest) / c))

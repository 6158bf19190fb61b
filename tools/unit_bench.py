#!/usr/bin/env python3
"""One convolution of the feeder (n = 16, bf16, channels_last) as the HIP kernels run it (forward with batch-norm
statistics, data gradient, weight gradient) beside PyTorch-ROCm's convolution forward + backward + a statistics
kernel.  Ten iterations are recorded into a HIP graph and replayed, so the figures are GPU time without the host's
per-call cost (which hides everything below ~60 us in an eagerly issued loop)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from segk import ops  # noqa: E402

ITERS = 10


def graphed_us(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(ITERS):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * ITERS) * 1e3


SHAPES = (("l1 1x1 256->64", 129, 256, 64, 1, 1), ("l1 1x1 64->256", 129, 64, 256, 1, 1),
          ("l2 1x1 512->128", 65, 512, 128, 1, 1), ("l2 1x1 128->512", 65, 128, 512, 1, 1),
          ("l3 1x1 1024->256", 33, 1024, 256, 1, 1), ("l3 1x1 256->1024", 33, 256, 1024, 1, 1),
          ("l4 1x1 2048->512", 33, 2048, 512, 1, 1), ("l4 1x1 512->2048", 33, 512, 2048, 1, 1),
          ("l2 3x3 128", 65, 128, 128, 3, 1), ("l3 3x3 256", 33, 256, 256, 3, 1), ("l4 3x3 512 d2", 33, 512, 512, 3, 2))


def main():
    torch.backends.cudnn.benchmark = True
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    n, dt, dev = 16, torch.bfloat16, "cuda"
    for name, H, Cin, Cout, R, dil in SHAPES:
        if only and only not in name:
            continue
        M = n * H * H
        x = torch.randn((M, Cin), device=dev).to(dt)
        dy = torch.randn((M, Cout), device=dev).to(dt)
        w = (torch.randn((Cout, Cin, R, R), device=dev) * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        wpt = ops.pack_weight(w, dt, transpose=True)
        out = torch.empty((M, Cout), dtype=dt, device=dev)
        dx = torch.empty((M, Cin), dtype=dt, device=dev)
        w_fwd = w if ops.weight_usable_in_place(w, R, Cout) else ops.pack_weight(w, dt)
        t_f = graphed_us(lambda: ops.conv_fwd([(x, w_fwd, R, dil)], n, H, H, Cout, out, want_stats=True))
        t_d = graphed_us(lambda: (ops.pack_weight(w, dt, transpose=True, out=wpt),
                                  ops.conv_fwd([(dy, wpt, R, dil)], n, H, H, Cin, dx)))
        t_w = graphed_us(lambda: ops.conv_wgrad(x, dy, n, H, H, R, dil, Cout, Cin, like=w))
        x4 = x.reshape(n, H, H, Cin).permute(0, 3, 1, 2).requires_grad_(True)
        w4 = w.clone().requires_grad_(True)
        g4 = dy.reshape(n, H, H, Cout).permute(0, 3, 1, 2)
        pad = dil * (R // 2)
        t_tf = graphed_us(lambda: F.conv2d(x4, w4, padding=pad, dilation=dil))

        def both():
            x4.grad = None
            w4.grad = None
            F.conv2d(x4, w4, padding=pad, dilation=dil).backward(g4)
        t_tb = graphed_us(both) - t_tf
        # the two halves of PyTorch-ROCm's backward on their own
        y4 = F.conv2d(x4, w4, padding=pad, dilation=dil).detach()

        def half(mask):
            return torch.ops.aten.convolution_backward(g4, x4.detach(), w4.detach(), None, [1, 1], [pad, pad], [dil, dil],
                                                       False, [0, 0], 1, mask)
        t_td = graphed_us(lambda: half([True, False, False]))
        t_tw = graphed_us(lambda: half([False, True, False]))
        t_st = graphed_us(lambda: ops.bn_local_sums(out, 0))
        print("%-18s M=%6d | segk fwd+stats %6.1f dgrad(+pack) %6.1f wgrad %6.1f = %6.1f us | torch fwd %6.1f + stats %5.1f + bwd %6.1f "
              "(dgrad %5.1f, wgrad %5.1f alone) = %6.1f us" % (name, M, t_f, t_d, t_w, t_f + t_d + t_w, t_tf, t_st, t_tb, t_td, t_tw,
                                                             t_tf + t_st + t_tb), flush=True)


if __name__ == "__main__":
    main()

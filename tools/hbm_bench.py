"""Times the HBM-bound kernels of the loss / resampling / evaluation path at the benchmark's shapes
(batch 16, 21 classes, 129x129 -> 513x513) and prints algorithmic GB/s against the 8 TB/s peak."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semantic-segmentation_amd"))
import torch
from segk import ops

dev = "cuda"
N, K, Hi, Wi, Ho, Wo = 16, 21, 129, 129, 513, 513
cpad = 24
torch.manual_seed(0)
logits = torch.randn(N, K, Ho, Wo, device=dev)
target = torch.randint(0, K, (N, Ho, Wo), device=dev).float()
target[:, :40] = 255.0
low = torch.randn(N * Hi * Wi, cpad, device=dev)
dlow = torch.empty(N * Hi * Wi, cpad, device=dev)
coef = torch.full((1,), 1e-6, device=dev)
cm = torch.zeros(K * K + 1, dtype=torch.int64, device=dev)


def timeit(name, fn, nbytes, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("%-18s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s" % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000), flush=True)


sums, lse = ops.ce_fwd(logits, target, None, 255)
pred = ops.argmax(logits)
B = lambda *ts: sum(t.numel() * t.element_size() for t in ts)
full = torch.empty(N, K, Ho, Wo, device=dev)
timeit("upsample_fwd nchw", lambda: ops.upsample_fwd(low[:, :K], N, K, Hi, Wi, full, Ho, Wo, dst_layout=ops.NCHW),
       N * K * (Hi * Wi + Ho * Wo) * 4)
timeit("ce_fwd", lambda: ops.ce_fwd(logits, target, None, 255), B(logits, target, lse))
timeit("ce_bwd", lambda: ops.ce_bwd(logits, target, None, lse, 255, coef), B(logits, target, lse, logits))
timeit("upsample_bwd nchw", lambda: ops.upsample_bwd(logits, ops.NCHW, N, K, Ho, Wo, dlow, cpad, Hi, Wi),
       N * (K * Ho * Wo * 4 + cpad * Hi * Wi * 4))
timeit("argmax", lambda: ops.argmax(logits), B(logits, pred))
timeit("upsample_argmax", lambda: ops.upsample_argmax(low[:, :K], N, Hi, Wi, Ho, Wo), N * Hi * Wi * K * 4 + B(pred))
timeit("confmat", lambda: ops.confmat_accumulate(target, pred, K, cm), B(target, pred, cm))
timeit("upsample_ce_fwd", lambda: ops.upsample_ce_fwd(low[:, :K], N, Hi, Wi, target, None, Ho, Wo, 255),
       N * Hi * Wi * K * 4 + B(target, lse))
timeit("upsample_ce_bwd", lambda: ops.upsample_ce_bwd(low[:, :K], N, Hi, Wi, target, None, lse, Ho, Wo, 255, coef, cpad),
       N * Hi * Wi * K * 4 + B(target, lse, dlow))

"""What does the library GEMM (hipBLASLt through torch.matmul) reach on the GEMM shapes of the head's convolutions?"""
import torch, time
dev = "cuda"
def bench(M, N, K, n=20):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): torch.matmul(a, b.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("M %7d N %5d K %6d  %8.1f us  %7.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
bench(8192, 8192, 8192)
bench(266256, 256, 2304)
bench(266256, 256, 2736)
bench(266256, 256, 256)
bench(17424, 256, 18432)
bench(17424, 256, 2048)
bench(266256, 48, 256)
